#!/bin/bash
# round 4: full GPU suite + smoke on the current build
set -u
O=$PWD/gpurun_out/full
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
: > $O/progress.log
timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -22 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
