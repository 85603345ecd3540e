#!/bin/bash
# round 4, session 17: per-wave timeline of the 256 x 256 prefill GEMM (placements 1 and 3), zero operands
set -u
O=$PWD/gpurun_out/s17
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "timeline (trace build), placement 3 / 1"
for s in 3 1; do ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace.so ACMI_BIG_TILE=1 ACMI_BIG_SCHED=$s timeout 300 python scripts/big_gemm_bench.py --trace --reps 3 > $O/big_gemm_trace_s$s.jsonl 2> $O/big_gemm_trace_s$s.err; cat $O/big_gemm_trace_s$s.jsonl | tee -a $O/progress.log; done
log "zero operands, placement 3"
ACMI_BIG_TILE=1 ACMI_BIG_SCHED=3 timeout 300 python scripts/big_gemm_bench.py --zeros > $O/big_gemm_s3_zeros.jsonl 2> $O/big_gemm_s3_zeros.err; cat $O/big_gemm_s3_zeros.jsonl | tee -a $O/progress.log
ACMI_BIG_TILE=1 ACMI_BIG_SCHED=3 timeout 300 python scripts/big_gemm_bench.py > $O/big_gemm_s3.jsonl 2> $O/big_gemm_s3.err; cat $O/big_gemm_s3.jsonl | tee -a $O/progress.log
log "done"
