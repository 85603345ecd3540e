#!/bin/bash
# round 4, session 25: counters of the prefill attention kernel alone
set -u
O=$PWD/gpurun_out/s25
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
cd /tmp
(rocprofv3 --list-avail > $O/list_avail.txt 2>&1 || rocprofv3 -L > $O/list_avail.txt 2>&1)
grep -o "SQ_[A-Z_0-9]*" $O/list_avail.txt | sort -u | tr '\n' ' ' | cut -c1-6000 | tee -a $O/progress.log
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  log "pmc: $set"
  ACMI_PFA_QB=1 ACMI_PFA_PF=0 timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$tag -- python $R/scripts/attn_prefill_bench.py --reps 2 > /dev/null 2> $O/pmc_$tag.err
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/scripts/summarize_pmc.py $f > $O/pmc_$tag.csv; grep -E "attn_prefill|Kernel|kernel" $O/pmc_$tag.csv | head -8 | tee -a $O/progress.log; else tail -3 $O/pmc_$tag.err | tee -a $O/progress.log; fi
done
log "done"
