#!/bin/bash
# round 4, session 33: lstm_xcd_kernel, which loads see another CU's plain stores through the XCD's L2?
# ACMI_LSTM_XCD: 1 = buffer_inv sc1 + plain load, 3 = nt load, 4 = sc0 load, 2 = memory side, 0 = all-CU form
set -u
O=$PWD/gpurun_out/s33
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "bench T = 300: modes 1, 3, 2, 0"
LSTM_MODES=1,3,2,0 timeout 120 python scripts/lstm_bench.py --T 300 --reps 3 2> $O/err_small | tee -a $O/progress.log
log "bench T = 1500: modes 1, 2, 0"
LSTM_MODES=1,2,0 timeout 120 python scripts/lstm_bench.py --T 1500 --reps 3 2> $O/err_big | tee -a $O/progress.log
log "done"
