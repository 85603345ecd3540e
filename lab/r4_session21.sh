#!/bin/bash
# round 4, session 21: the same timeline with all-zero operands (is the MFMA rate data dependent?)
set -u
O=$PWD/gpurun_out/s21
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
for s in 5 6; do
log "timeline, placement $s, zeros"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace.so ACMI_BIG_TILE=1 ACMI_BIG_SCHED=$s timeout 300 python scripts/big_gemm_bench.py --trace --zeros --reps 3 > $O/big_gemm_trace_s${s}_zeros.jsonl 2> $O/err; cat $O/big_gemm_trace_s${s}_zeros.jsonl | tee -a $O/progress.log
done
log "done"
