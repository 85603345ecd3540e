#!/bin/bash
# round 4, session 19: all requests from the older wave of each SIMD (ACMI_BIG_SCHED=4)
set -u
O=$PWD/gpurun_out/s19
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "linear_big + prefill tests, 256 tile + placement 4 forced"
ACMI_BIG_TILE=1 ACMI_BIG_SCHED=4 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "linear_big or prefill or golden or window or melody or streaming" 2>&1 | tail -3 | tee -a $O/progress.log
log "GEMM alone, 256 tile, placement 3 / 4"
for s in 3 4; do ACMI_BIG_TILE=1 ACMI_BIG_SCHED=$s timeout 300 python scripts/big_gemm_bench.py > $O/big_gemm_s$s.jsonl 2> $O/big_gemm_s$s.err; cat $O/big_gemm_s$s.jsonl | tee -a $O/progress.log; done
log "timeline (trace build), placement 4"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace.so ACMI_BIG_TILE=1 ACMI_BIG_SCHED=4 timeout 300 python scripts/big_gemm_bench.py --trace --reps 3 > $O/big_gemm_trace_s4.jsonl 2> $O/big_gemm_trace_s4.err; cat $O/big_gemm_trace_s4.jsonl | tee -a $O/progress.log
log "prefill bench, placement 3 / 4"
for s in 3 4; do ACMI_BIG_SCHED=$s timeout 600 python scripts/prefill_bench.py window > $O/prefill_s$s.jsonl 2> $O/prefill_s$s.err; cut -c1-300 $O/prefill_s$s.jsonl | tee -a $O/progress.log; done
log "done"
