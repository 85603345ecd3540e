#!/bin/bash
# round 4, session 34: lstm_xcd_kernel with nt polling loads and the four gate activations side by side: tests, step time, codec
set -u
O=$PWD/gpurun_out/s34
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "bench T = 1500: modes 1, 2, 0; B = 8, 1, 16"
timeout 120 python scripts/lstm_bench.py --reps 3 2> $O/err_b8 | tee -a $O/progress.log
LSTM_MODES=1,0 timeout 120 python scripts/lstm_bench.py --B 1 --T 500 --reps 3 2> $O/err_b1 | tee -a $O/progress.log
LSTM_MODES=1,0 timeout 120 python scripts/lstm_bench.py --B 16 --T 500 --reps 3 2> $O/err_b16 | tee -a $O/progress.log
log "lstm / codec tests"
timeout 800 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_parity_configs.py -q -x -m gpu -k "encodec or lstm or codec or compression or stereo or graph or seanet or epic" 2>&1 | tail -4 | tee -a $O/progress.log
log "codec bench"
timeout 300 python scripts/codec_bench.py > $O/codec_bench.jsonl 2> $O/codec_bench.err; cut -c1-700 $O/codec_bench.jsonl | tee -a $O/progress.log
log "codec bench, all-CU form"
ACMI_LSTM_XCD=0 timeout 300 python scripts/codec_bench.py > $O/codec_bench_xcd0.jsonl 2> $O/codec_bench0.err; cut -c1-700 $O/codec_bench_xcd0.jsonl | tee -a $O/progress.log
log "done"
