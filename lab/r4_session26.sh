#!/bin/bash
# round 4, session 26: prefill attention with the lighter softmax arithmetic
set -u
O=$PWD/gpurun_out/s26
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "attention + prefill tests (QB 2, then QB 1)"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "attn_prefill or prefill or window or melody or golden" 2>&1 | tail -2 | tee -a $O/progress.log
ACMI_PFA_QB=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attn_prefill" 2>&1 | tail -1 | tee -a $O/progress.log
log "kernel alone: QB 1 / 2"
for qb in 1 2; do ACMI_PFA_QB=$qb timeout 200 python scripts/attn_prefill_bench.py 2>/dev/null | tee -a $O/progress.log; done
log "counters, QB 2"
cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$tag -- python $R/scripts/attn_prefill_bench.py --reps 2 > /dev/null 2> $O/pmc_$tag.err
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  python $R/scripts/summarize_pmc.py $f | grep attn_prefill | tee -a $O/progress.log
done
cd $R
log "prefill bench"
timeout 600 python scripts/prefill_bench.py window melody > $O/prefill.jsonl 2> $O/prefill.err; cut -c1-300 $O/prefill.jsonl | tee -a $O/progress.log
ACMI_PFA_QB=1 timeout 600 python scripts/prefill_bench.py window > $O/prefill_qb1.jsonl 2> $O/prefill_qb1.err; cut -c1-300 $O/prefill_qb1.jsonl | tee -a $O/progress.log
log "done"
