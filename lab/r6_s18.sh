#!/bin/bash
# round 6 session 18: final build -- whole GPU suite, smoke, bench-format lines (configs[2], [1], [3]), the bench command under rocprofv3
set -u
O=$PWD/gpurun_out/r6s18; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -5 | tee $O/full_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
ACMI_BENCH_INSITU_KEEP=$O/cfg2_insitu_kernel_stats.csv timeout 1500 python bench.py --steps 3 --warmup 1 > $O/line_cfg2.json 2> $O/line_cfg2.err; tail -1 $O/line_cfg2.err; cut -c1-220 $O/line_cfg2.json
ACMI_BENCH_INSITU_KEEP=$O/cfg1_insitu_kernel_stats.csv timeout 600 python bench.py --steps 3 --warmup 1 --model facebook/musicgen-small --batch 1 --duration 10 --greedy --no-cpu-baseline > $O/line_cfg1.json 2> $O/line_cfg1.err; cut -c1-220 $O/line_cfg1.json
ACMI_BENCH_INSITU_KEEP=$O/cfg3_insitu_kernel_stats.csv timeout 900 python bench.py --steps 2 --warmup 1 --model facebook/musicgen-large --batch 8 --duration 30 --no-cpu-baseline > $O/line_cfg3.json 2> $O/line_cfg3.err; cut -c1-220 $O/line_cfg3.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
find $O/ks -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \; ; python $GRAFT_REPO_ROOT/scripts/top_kernels.py $O/ks 12 | tee $O/bench_top_kernels.txt; rm -rf $O/ks
