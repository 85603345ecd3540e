#!/bin/bash
# round 4, session 20: one request block per stage (M0 once, SGPR bases: ACMI_BIG_SCHED=5); the loop without requests (6, timing only)
set -u
O=$PWD/gpurun_out/s20
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "linear_big tests, 256 tile + placement 5 forced"
ACMI_BIG_TILE=1 ACMI_BIG_SCHED=5 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "linear_big" 2>&1 | tail -3 | tee -a $O/progress.log
log "GEMM alone, 256 tile, placement 3 / 5 / 6"
for s in 3 5 6; do ACMI_BIG_TILE=1 ACMI_BIG_SCHED=$s timeout 300 python scripts/big_gemm_bench.py > $O/big_gemm_s$s.jsonl 2> $O/big_gemm_s$s.err; cat $O/big_gemm_s$s.jsonl | tee -a $O/progress.log; done
log "timeline (trace build), placement 5 / 6"
for s in 5 6; do ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace.so ACMI_BIG_TILE=1 ACMI_BIG_SCHED=$s timeout 300 python scripts/big_gemm_bench.py --trace --reps 3 > $O/big_gemm_trace_s$s.jsonl 2> $O/big_gemm_trace_s$s.err; grep -E "trace" $O/big_gemm_trace_s$s.jsonl | tee -a $O/progress.log; done
log "done"
