#!/bin/bash
# round 4, session 23: the cleaned-up prefill GEMM (one K loop for both tiles): tests, GEMM alone, prefill, timeline
set -u
O=$PWD/gpurun_out/s23
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "tests: auto tile choice / 128 forced / 256 forced"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "linear_big or prefill or golden or window or melody or streaming" 2>&1 | tail -2 | tee -a $O/progress.log
ACMI_BIG_TILE=0 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "linear_big or prefill or golden or window or melody or streaming" 2>&1 | tail -2 | tee -a $O/progress.log
ACMI_BIG_TILE=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "linear_big or prefill or golden or window or melody or streaming" 2>&1 | tail -2 | tee -a $O/progress.log
log "GEMM alone: 128 / 256"
for t in 0 1; do ACMI_BIG_TILE=$t timeout 300 python scripts/big_gemm_bench.py > $O/big_gemm_t$t.jsonl 2> $O/big_gemm_t$t.err; cat $O/big_gemm_t$t.jsonl | tee -a $O/progress.log; done
log "prefill bench: auto / 128 forced"
timeout 600 python scripts/prefill_bench.py window melody > $O/prefill.jsonl 2> $O/prefill.err; cut -c1-300 $O/prefill.jsonl | tee -a $O/progress.log
ACMI_BIG_TILE=0 timeout 600 python scripts/prefill_bench.py window melody > $O/prefill_t128.jsonl 2> $O/prefill_t128.err; cut -c1-300 $O/prefill_t128.jsonl | tee -a $O/progress.log
log "timeline of the final loop"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace.so ACMI_BIG_TILE=1 timeout 300 python scripts/big_gemm_bench.py --trace --reps 3 > $O/big_gemm_trace.jsonl 2> $O/err; cat $O/big_gemm_trace.jsonl | tee -a $O/progress.log
log "done"
