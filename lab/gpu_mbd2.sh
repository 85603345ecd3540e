#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mbd2; mkdir -p $O
for s in 1 10; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$s -- python $R/scripts/mbd_bench.py --seconds $s --reps 3 > $O/prof_$s.log 2>&1)
  find /tmp/kt$s -name '*kernel_stats.csv' -exec cp {} $O/mbd_kernel_stats_${s}s.csv \;
  find /tmp/kt$s -name '*kernel_trace.csv' -exec cp {} $O/mbd_kernel_trace_${s}s.csv \;
done
head -14 $O/mbd_kernel_stats_1s.csv; head -14 $O/mbd_kernel_stats_10s.csv
