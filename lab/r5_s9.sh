#!/bin/bash
# round 5 session 9: the full GPU suite + smoke on the round's build
set -u
O=$PWD/gpurun_out/r5s9; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -12 | tee $O/full_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
