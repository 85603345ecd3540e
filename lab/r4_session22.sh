#!/bin/bash
# round 4, session 22: the K loop without its LDS reads (timing only): 7 = no weight-fragment reads, 8 = no reads at all
set -u
O=$PWD/gpurun_out/s22
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
for s in 7 8; do
log "timeline, variant $s"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace.so ACMI_BIG_TILE=1 ACMI_BIG_SCHED=$s timeout 300 python scripts/big_gemm_bench.py --trace --reps 3 > $O/big_gemm_trace_s${s}.jsonl 2> $O/err; cat $O/big_gemm_trace_s${s}.jsonl | tee -a $O/progress.log
done
log "done"
