#!/bin/bash
# round 4, session 31: ACMI_BIG_PP 1 (priority only while multiplying) / 2 (no priority) / 0
set -u
O=$PWD/gpurun_out/s31
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "GEMM alone: PP 1 / PP 2 / PP 0"
timeout 300 python scripts/big_gemm_bench.py 2> $O/err_a | tee -a $O/progress.log
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_pp2.so timeout 300 python scripts/big_gemm_bench.py 2> $O/err_b | sed 's/^/PP2 /' | tee -a $O/progress.log
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_pp0.so timeout 300 python scripts/big_gemm_bench.py 2> $O/err_b | sed 's/^/PP0 /' | tee -a $O/progress.log
log "timeline PP 1"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace.so ACMI_BIG_TILE=1 timeout 300 python scripts/big_gemm_bench.py --trace --reps 3 2> $O/err_tr | grep -A1 qkv | tee -a $O/progress.log
log "timeline PP 2"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace_pp2.so ACMI_BIG_TILE=1 timeout 300 python scripts/big_gemm_bench.py --trace --reps 3 2> $O/err_tr | grep -A1 qkv | tee -a $O/progress.log
log "done"
