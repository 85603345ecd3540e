#!/bin/bash
# round 4, session 28: prefill GEMM, 256 x 128 tiles with two workgroups per CU (ACMI_BIG_TILE=2) against the 256 x 256 tile
set -u
O=$PWD/gpurun_out/s28
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "tests: 256 x 128 forced"
ACMI_BIG_TILE=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "linear_big" 2>&1 | tail -3 | tee -a $O/progress.log
log "GEMM alone: 256 x 256 / 256 x 128, twice"
for rep in 1 2; do for t in 1 2; do ACMI_BIG_TILE=$t timeout 300 python scripts/big_gemm_bench.py 2> $O/err_t$t | tee -a $O/progress.log; done; done
log "timeline 256 x 128"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace.so ACMI_BIG_TILE=2 timeout 300 python scripts/big_gemm_bench.py --trace --reps 3 2> $O/err_tr | tee -a $O/progress.log
log "prefill bench: 256 x 256 / 256 x 128"
for t in 1 2; do ACMI_BIG_TILE=$t timeout 600 python scripts/prefill_bench.py window 2> $O/prefill_t$t.err | cut -c1-300 | tee -a $O/progress.log; done
log "done"
