#!/bin/bash
# round 4, session 35: lstm_xcd_kernel for H = 512 / 768 / 1024: tests, step times, EnCodec-24k geometry with / without the two-layer wavefront
set -u
O=$PWD/gpurun_out/s35
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "lstm tests"
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "lstm" 2>&1 | tail -4 | tee -a $O/progress.log
log "step times: H = 1024 B = 8; H = 512 B = 1 / 16; H = 768 B = 8"
LSTM_MODES=1,0 timeout 120 python scripts/lstm_bench.py --reps 3 2> $O/err1 | tee -a $O/progress.log
LSTM_MODES=1,0 timeout 120 python scripts/lstm_bench.py --H 512 --B 1 --T 750 --reps 3 2> $O/err2 | tee -a $O/progress.log
LSTM_MODES=1,0 timeout 120 python scripts/lstm_bench.py --H 512 --B 16 --T 750 --reps 3 2> $O/err3 | tee -a $O/progress.log
LSTM_MODES=1,0 timeout 120 python scripts/lstm_bench.py --H 768 --B 8 --T 500 --reps 3 2> $O/err4 | tee -a $O/progress.log
log "codec bench, default"
timeout 300 python scripts/codec_bench.py 2> $O/codec_bench.err | cut -c1-600 | tee -a $O/progress.log
log "codec bench, ACMI_LSTM_WAVE=0 (per-layer launches: the XCD-local form at H = 512 too)"
ACMI_LSTM_WAVE=0 timeout 300 python scripts/codec_bench.py 2> $O/codec_bench0.err | cut -c1-600 | tee -a $O/progress.log
log "done"
