#!/bin/bash
# round 3, GPU session 6: the whole -m gpu suite on the current build + smoke
set -u
OUT=gpurun_out/s6
mkdir -p $OUT
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1
echo "== pytest -m gpu (full)" | tee $OUT/progress.log
timeout 2400 python -m pytest tests -m gpu -q -rP --maxfail=40 > $OUT/pytest_full.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/progress.log
grep -E "^\[(parity|near-tie|single-term)\]|passed|failed|^FAILED|^ERROR" $OUT/pytest_full.log > $OUT/pytest_summary.log
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_summary.log | tail -20 | tee -a $OUT/progress.log
echo "== smoke" | tee -a $OUT/progress.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $OUT/progress.log
echo "== done" | tee -a $OUT/progress.log
