#!/bin/bash
# round 4, session 27: prefill attention, prefetch of the next key block on top of the lighter softmax
set -u
O=$PWD/gpurun_out/s27
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "attention + prefill tests"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "attn_prefill or prefill or window or melody or golden" 2>&1 | tail -2 | tee -a $O/progress.log
log "kernel alone: PF 0 / 1 (QB 2), twice"
for rep in 1 2; do for pf in 0 1; do ACMI_PFA_PF=$pf timeout 200 python scripts/attn_prefill_bench.py 2>/dev/null | tee -a $O/progress.log; done; done
log "prefill bench PF 1 / PF 0"
timeout 600 python scripts/prefill_bench.py window > $O/prefill.jsonl 2> $O/prefill.err; cut -c1-300 $O/prefill.jsonl | tee -a $O/progress.log
ACMI_PFA_PF=0 timeout 600 python scripts/prefill_bench.py window > $O/prefill_pf0.jsonl 2> $O/prefill_pf0.err; cut -c1-300 $O/prefill_pf0.jsonl | tee -a $O/progress.log
log "done"
