#!/bin/bash
# round 6 session 7: checkpoint -- the whole GPU suite with durations, smoke, the bench line
set -u
O=$PWD/gpurun_out/r6s7; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -x -m gpu --durations=25 2>&1 | tail -40 | tee $O/full_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
ACMI_BENCH_INSITU_KEEP=$O/bench_insitu_kernel_stats.csv timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; tail -2 $O/bench_n1.err; cut -c1-300 $O/bench_n1.json
