#!/bin/bash
# round 4, session 15: where the 256 x 256 prefill GEMM spends its cycles; request placement variants
set -u
O=$PWD/gpurun_out/s15
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "linear_big tests, each request placement"
for s in 0 1 2; do ACMI_BIG_TILE=1 ACMI_BIG_SCHED=$s timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "linear_big" 2>&1 | tail -1 | tee -a $O/progress.log; done
log "GEMM alone, 256 tile, request placement 0 / 1 / 2"
for s in 0 1 2; do ACMI_BIG_TILE=1 ACMI_BIG_SCHED=$s timeout 300 python scripts/big_gemm_bench.py > $O/big_gemm_s$s.jsonl 2> $O/big_gemm_s$s.err; cat $O/big_gemm_s$s.jsonl | tee -a $O/progress.log; done
log "timeline (trace build), placement 0 / 1"
for s in 0 1; do ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace.so ACMI_BIG_TILE=1 ACMI_BIG_SCHED=$s timeout 300 python scripts/big_gemm_bench.py --trace --reps 3 > $O/big_gemm_trace_s$s.jsonl 2> $O/big_gemm_trace_s$s.err; cat $O/big_gemm_trace_s$s.jsonl | tee -a $O/progress.log; done
log "done"
