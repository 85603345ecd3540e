#!/bin/bash
# round 4, session 1: kernarg latency lab, in-kernel timeline of the GEMM launches (round-3 kernels vs preloaded-argument
# kernels), same-box A/B of the two libraries, GPU parity suite on the new library
set -u
O=$PWD/gpurun_out/s1
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "kernarg lab"
timeout 120 lab/kernarg_lab 2>&1 | tee $O/kernarg_lab.log | tee -a $O/progress.log
log "timeline, round-3 kernels"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_trace_r3.so timeout 300 python scripts/lin_timeline.py --out $O/lin_timeline_r3.csv --raw $O/lin_timeline_r3.npz > $O/lin_timeline_r3.log 2>&1
tail -8 $O/lin_timeline_r3.log | tee -a $O/progress.log
log "timeline, preloaded arguments"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_trace.so timeout 300 python scripts/lin_timeline.py --out $O/lin_timeline_new.csv --raw $O/lin_timeline_new.npz > $O/lin_timeline_new.log 2>&1
tail -8 $O/lin_timeline_new.log | tee -a $O/progress.log
log "GEMM chain us / launch (bench.measure_lin_kernel): r3, new"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_r3.so timeout 300 python scripts/dbg_chain.py 2>&1 | tail -1 | tee -a $O/progress.log
timeout 300 python scripts/dbg_chain.py 2>&1 | tail -1 | tee -a $O/progress.log
log "bench A/B (3 steps): r3, new, r3, new"
for i in 1 2; do
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_r3.so timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_r3_$i.json 2> $O/bench_r3_$i.err
cut -c1-160 $O/bench_r3_$i.json | tee -a $O/progress.log
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_new_$i.json 2> $O/bench_new_$i.err
cut -c1-160 $O/bench_new_$i.json | tee -a $O/progress.log
done
log "GPU parity suite on the new library"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee -a $O/progress.log
log "done"
