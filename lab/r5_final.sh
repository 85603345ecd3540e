#!/bin/bash
# round 5, final evidence on the last build: the whole GPU suite, smoke, the bench line (5 steps; the driver runs 20) with its in-situ kernel statistics kept
set -u
O=$PWD/gpurun_out/r5final; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/full_pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
ACMI_BENCH_INSITU_KEEP=$O/bench_insitu_kernel_stats.csv timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 2>$O/bench.err | tee $O/bench_n1.json | cut -c1-400
tail -3 $O/bench.err
