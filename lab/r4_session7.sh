#!/bin/bash
# round 4, session 7: wave-count-generic epilogues, left-padded streams (two_step_cfg + unequal prepend), MBD released shape
set -u
O=$PWD/gpurun_out/s7
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "MBD released shape test"
timeout 600 python -m pytest tests/test_gpu_mbd.py -m gpu -x -q -k "released_shape" 2>&1 | tail -40 | tee -a $O/progress.log
log "kernel + LM tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "not small_architecture" 2>&1 | tail -12 | tee -a $O/progress.log
log "timelines: medium B=8, small B=1"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_trace.so timeout 300 python scripts/lin_timeline.py --out $O/lin_timeline.csv > $O/lin_timeline.log 2>&1
tail -7 $O/lin_timeline.log | tee -a $O/progress.log
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_trace.so timeout 300 python scripts/lin_timeline.py --model facebook/musicgen-small --batch 1 --frames 300 --out $O/lin_timeline_small_b1.csv > $O/lin_timeline_small_b1.log 2>&1
tail -7 $O/lin_timeline_small_b1.log | tee -a $O/progress.log
log "chain + benches"
timeout 300 python scripts/dbg_chain.py 2>&1 | tail -1 | tee -a $O/progress.log
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-160 $O/bench.json | tee -a $O/progress.log
timeout 600 python bench.py --model facebook/musicgen-small --batch 1 --duration 10 --greedy --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_small_b1.json 2> $O/bench_small_b1.err
cut -c1-200 $O/bench_small_b1.json | tee -a $O/progress.log
log "done"
