#!/bin/bash
# round 4, session 32: LSTM at H = 1024, one recurrence per XCD (lstm_xcd_kernel)
set -u
O=$PWD/gpurun_out/s32
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "bench first (small T): modes 2, 1, 0"
LSTM_MODES=2,1,0 timeout 120 python scripts/lstm_bench.py --T 50 --reps 2 2> $O/err_small | tee -a $O/progress.log
tail -3 $O/err_small
log "tests (default mode 1)"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "lstm" 2>&1 | tail -3 | tee -a $O/progress.log
log "tests (mode 2)"
ACMI_LSTM_XCD=2 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "lstm" 2>&1 | tail -3 | tee -a $O/progress.log
log "bench T = 1500: modes 1, 0, 2; B = 8 / 1"
timeout 200 python scripts/lstm_bench.py 2> $O/err_b8 | tee -a $O/progress.log
timeout 200 python scripts/lstm_bench.py --B 1 --T 500 2> $O/err_b1 | tee -a $O/progress.log
log "done"
