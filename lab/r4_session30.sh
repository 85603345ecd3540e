#!/bin/bash
# round 4, session 30: prefill GEMM, the upper half of the waves multiplies first and requests afterwards (ACMI_BIG_PP)
set -u
O=$PWD/gpurun_out/s30
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "tests"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "linear_big or prefill or golden or window or melody" 2>&1 | tail -3 | tee -a $O/progress.log
log "GEMM alone: PP 1 / PP 0, twice"
for rep in 1 2; do
  timeout 300 python scripts/big_gemm_bench.py 2> $O/err_a | tee -a $O/progress.log
  ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_pp0.so timeout 300 python scripts/big_gemm_bench.py 2> $O/err_b | sed 's/^/PP0 /' | tee -a $O/progress.log
done
log "timeline PP 1"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace.so ACMI_BIG_TILE=1 timeout 300 python scripts/big_gemm_bench.py --trace --reps 3 2> $O/err_tr | tee -a $O/progress.log
log "prefill bench: PP 1 / PP 0"
timeout 600 python scripts/prefill_bench.py window melody 2> $O/prefill.err | cut -c1-300 | tee -a $O/progress.log
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_pp0.so timeout 600 python scripts/prefill_bench.py window 2> $O/prefill0.err | cut -c1-300 | tee -a $O/progress.log
log "done"
