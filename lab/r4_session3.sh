#!/bin/bash
# round 4, session 3: specialised epilogues + LayerNorm statistics from the fragments: parity suite, timeline, A/B
set -u
O=$PWD/gpurun_out/s3
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "kernel tests first"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "linear or folded or half or pair or statistics or single or split" 2>&1 | tail -8 | tee -a $O/progress.log
log "timeline"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_trace.so timeout 300 python scripts/lin_timeline.py --out $O/lin_timeline.csv --raw $O/lin_timeline.npz > $O/lin_timeline.log 2>&1
tail -7 $O/lin_timeline.log | tee -a $O/progress.log
log "chain: gram, partials"
timeout 300 python scripts/dbg_chain.py 2>&1 | tail -1 | tee -a $O/progress.log
ACMI_LN_GRAM=0 timeout 300 python scripts/dbg_chain.py 2>&1 | tail -1 | tee -a $O/progress.log
log "bench: gram, partials, r3 library"
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cut -c1-160 $O/bench.json | tee -a $O/progress.log
ACMI_LN_GRAM=0 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_partials.json 2> $O/bench_partials.err
cut -c1-160 $O/bench_partials.json | tee -a $O/progress.log
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_r3.so timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_r3.json 2> $O/bench_r3.err
cut -c1-160 $O/bench_r3.json | tee -a $O/progress.log
log "GPU parity suite"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee -a $O/progress.log
log "done"
