#!/bin/bash
# round 4, session 14: 256 x 256 tiles + transposed epilogue of the prefill GEMM
set -u
O=$PWD/gpurun_out/s14
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "linear_big tests"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "linear_big" 2>&1 | tail -6 | tee -a $O/progress.log
log "prefill / lm tests, default tile choice"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "prefill or golden or window or melody or streaming" 2>&1 | tail -6 | tee -a $O/progress.log
log "the same with the 256 x 256 tile forced"
ACMI_BIG_TILE=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "prefill or golden or window or melody or streaming or linear_big" 2>&1 | tail -6 | tee -a $O/progress.log
log "GEMM alone: 128 tile / 256 tile"
ACMI_BIG_TILE=0 timeout 300 python scripts/big_gemm_bench.py > $O/big_gemm_t128.jsonl 2> $O/big_gemm_t128.err; cat $O/big_gemm_t128.jsonl | tee -a $O/progress.log
ACMI_BIG_TILE=1 timeout 300 python scripts/big_gemm_bench.py > $O/big_gemm_t256.jsonl 2> $O/big_gemm_t256.err; cat $O/big_gemm_t256.jsonl | tee -a $O/progress.log
log "prefill bench: auto / 128 forced"
timeout 600 python scripts/prefill_bench.py window melody > $O/prefill.jsonl 2> $O/prefill.err; cut -c1-300 $O/prefill.jsonl | tee -a $O/progress.log
ACMI_BIG_TILE=0 timeout 600 python scripts/prefill_bench.py window > $O/prefill_t128.jsonl 2> $O/prefill_t128.err; cut -c1-300 $O/prefill_t128.jsonl | tee -a $O/progress.log
log "done"
