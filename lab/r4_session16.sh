#!/bin/bash
# round 4, session 16: fragments read half a K step ahead (ACMI_BIG_SCHED=3), A&S GELU for bf16 results
set -u
O=$PWD/gpurun_out/s16
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "linear_big tests: 256 tile forced, placement 3 / 1; auto"
for s in 3 1; do ACMI_BIG_TILE=1 ACMI_BIG_SCHED=$s timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "linear_big" 2>&1 | tail -1 | tee -a $O/progress.log; done
ACMI_BIG_SCHED=3 timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "linear_big" 2>&1 | tail -1 | tee -a $O/progress.log
log "prefill / lm tests with the 256 tile + placement 3 forced"
ACMI_BIG_TILE=1 ACMI_BIG_SCHED=3 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "prefill or golden or window or melody or streaming" 2>&1 | tail -3 | tee -a $O/progress.log
log "GEMM alone, 256 tile, placement 1 / 3"
for s in 1 3; do ACMI_BIG_TILE=1 ACMI_BIG_SCHED=$s timeout 300 python scripts/big_gemm_bench.py > $O/big_gemm_s$s.jsonl 2> $O/big_gemm_s$s.err; cat $O/big_gemm_s$s.jsonl | tee -a $O/progress.log; done
log "timeline (trace build), placement 3"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace.so ACMI_BIG_TILE=1 ACMI_BIG_SCHED=3 timeout 300 python scripts/big_gemm_bench.py --trace --reps 3 > $O/big_gemm_trace_s3.jsonl 2> $O/big_gemm_trace_s3.err; cat $O/big_gemm_trace_s3.jsonl | tee -a $O/progress.log
log "prefill bench, placement 3"
ACMI_BIG_SCHED=3 timeout 600 python scripts/prefill_bench.py window melody > $O/prefill_s3.jsonl 2> $O/prefill_s3.err; cut -c1-300 $O/prefill_s3.jsonl | tee -a $O/progress.log
log "done"
