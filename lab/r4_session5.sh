#!/bin/bash
# round 4, session 5: attention kernel with preloaded arguments + speculative first chunk; request-order variants of the GEMMs
set -u
O=$PWD/gpurun_out/s5
R=$PWD
L=$R/audiocraft_amd/csrc
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "attention + LM tests (new attention kernel)"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "attn or attention or lm_ or golden or streaming or melody or rope" 2>&1 | tail -6 | tee -a $O/progress.log
log "attention microbench: new, round-3 library"
timeout 300 python scripts/attn_bench.py > $O/attn_bench_new.log 2>&1; tail -3 $O/attn_bench_new.log | cut -c1-400 | tee -a $O/progress.log
ACMI_LIB=$L/libacmi_r3.so timeout 300 python scripts/attn_bench.py > $O/attn_bench_r3.log 2>&1; tail -3 $O/attn_bench_r3.log | cut -c1-400 | tee -a $O/progress.log
log "GEMM chain by request order: 0 (weights first), 1, 2, 4"
timeout 300 python scripts/dbg_chain.py 2>&1 | tail -1 | tee -a $O/progress.log
for k in 1 2 4; do ACMI_LIB=$L/libacmi_ord$k.so timeout 300 python scripts/dbg_chain.py 2>&1 | tail -1 | tee -a $O/progress.log; done
log "bench: order 0, 2, 1"
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_ord0.json 2> $O/bench_ord0.err; cut -c1-160 $O/bench_ord0.json | tee -a $O/progress.log
ACMI_LIB=$L/libacmi_ord2.so timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_ord2.json 2> $O/bench_ord2.err; cut -c1-160 $O/bench_ord2.json | tee -a $O/progress.log
ACMI_LIB=$L/libacmi_ord1.so timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_ord1.json 2> $O/bench_ord1.err; cut -c1-160 $O/bench_ord1.json | tee -a $O/progress.log
log "timeline, order 2"
ACMI_LIB=$L/libacmi_trace_ord2.so timeout 300 python scripts/lin_timeline.py --out $O/lin_timeline_ord2.csv > $O/lin_timeline_ord2.log 2>&1
tail -7 $O/lin_timeline_ord2.log | tee -a $O/progress.log
log "done"
