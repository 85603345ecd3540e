#!/bin/bash
# round 5 session 10: the bench line with the driver's flags + rocprofv3 kernel statistics of the same command
set -u
O=$PWD/gpurun_out/r5s10; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
ACMI_BENCH_INSITU_KEEP=$O/bench_insitu_kernel_stats.csv timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1_driver_flags.json 2> $O/bench_n1_driver_flags.err
cut -c1-600 $O/bench_n1_driver_flags.json
cd /tmp && rm -rf /tmp/prof_b && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_under_rocprof.json 2> /dev/null
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); cp $f $O/bench_kernel_stats.csv; head -12 $O/bench_kernel_stats.csv | cut -c1-150
