#!/bin/bash
# round 4, session 24: prefill attention: two query blocks per wave, prefetch of the next key block
set -u
O=$PWD/gpurun_out/s24
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "attention tests (default: QB 2 + prefetch; then QB 2 without prefetch, QB 1 with)"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "attn_prefill or prefill or window or melody" 2>&1 | tail -2 | tee -a $O/progress.log
ACMI_PFA_PF=0 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attn_prefill" 2>&1 | tail -1 | tee -a $O/progress.log
ACMI_PFA_QB=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attn_prefill" 2>&1 | tail -1 | tee -a $O/progress.log
log "kernel alone: QB x PF"
for qb in 1 2; do for pf in 0 1; do ACMI_PFA_QB=$qb ACMI_PFA_PF=$pf timeout 200 python scripts/attn_prefill_bench.py 2>/dev/null | tee -a $O/progress.log; done; done
log "prefill bench (default)"
timeout 600 python scripts/prefill_bench.py window melody > $O/prefill.jsonl 2> $O/prefill.err; cut -c1-300 $O/prefill.jsonl | tee -a $O/progress.log
log "done"
