#!/bin/bash
# round 4, session 4: wave-count-specialised epilogues: kernel + model tests, timeline, chain, bench
set -u
O=$PWD/gpurun_out/s4
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "kernel + model tests"
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_parity_configs.py -m gpu -x -q --durations=12 2>&1 | tail -22 | tee -a $O/progress.log
log "timeline"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_trace.so timeout 300 python scripts/lin_timeline.py --out $O/lin_timeline.csv --raw $O/lin_timeline.npz > $O/lin_timeline.log 2>&1
tail -7 $O/lin_timeline.log | tee -a $O/progress.log
log "chain"
timeout 300 python scripts/dbg_chain.py 2>&1 | tail -1 | tee -a $O/progress.log
log "bench"
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cut -c1-160 $O/bench.json | tee -a $O/progress.log
log "done"
