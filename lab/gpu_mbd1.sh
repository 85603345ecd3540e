#!/bin/bash
# MBD: failing test rerun, cost at the released geometry, kernel stats of one forward
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mbd1; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_mbd.py -q -x 2>&1 | tail -5 > $O/pytest.log
timeout 300 python scripts/mbd_bench.py --seconds 10 --cpu > $O/mbd_bench_10s.json 2> $O/mbd_bench_10s.err
timeout 300 python scripts/mbd_bench.py --seconds 30 --batch 2 > $O/mbd_bench_30s_b2.json 2>> $O/mbd_bench_10s.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o mbd -- python $R/scripts/mbd_bench.py --seconds 10 --reps 3 > $O/prof.log 2>&1
find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/mbd_kernel_stats.csv \;
rm -rf $O/prof
cat $O/pytest.log $O/mbd_bench_10s.json $O/mbd_bench_30s_b2.json; tail -3 $O/mbd_bench_10s.err; head -12 $O/mbd_kernel_stats.csv
