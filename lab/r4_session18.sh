#!/bin/bash
# round 4, session 18: per-kernel times of the window prefill, 128 tile vs 256 tile (placement 3), same box
set -u
O=$PWD/gpurun_out/s18
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
for v in t128 t256s3 t256s1; do
  case $v in t128) export ACMI_BIG_TILE=0 ACMI_BIG_SCHED=0;; t256s3) export ACMI_BIG_TILE=1 ACMI_BIG_SCHED=3;; t256s1) export ACMI_BIG_TILE=1 ACMI_BIG_SCHED=1;; esac
  log "$v: prefill bench"
  timeout 600 python scripts/prefill_bench.py window > $O/prefill_$v.jsonl 2> $O/prefill_$v.err; cut -c1-300 $O/prefill_$v.jsonl | tee -a $O/progress.log
  log "$v: kernel stats"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -- python $R/scripts/prefill_bench.py window --reps 2 > /dev/null 2>&1)
  cp $(find /tmp/kt_$v -name "*kernel_stats.csv" | head -1) $O/prefill_kernel_stats_$v.csv
  python scripts/short_names.py $O/prefill_kernel_stats_$v.csv | grep -E "lin_big|attn_prefill|ln_tile" | head -8 | tee -a $O/progress.log
done
log "done"
