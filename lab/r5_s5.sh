#!/bin/bash
# round 5 session 5: engine v3, larger chunks, plain vs sc1 consumer loads, skew diagnostics; small / large geometry
set -u
O=$PWD/gpurun_out/r5s5; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 python scripts/engine_lab.py --model medium --layers 48 --reps 60 --check-reps 10 --modes 2,0 --waves 8 --chunks 8,12,24 --epi 0 --sleep 4,8 --trace $O/tl 2>&1 | grep -v "^wave,\|^control,\|^compute0,\|^all,\|^skew\|f64 rest\|vs the launch" | tee $O/engine_lab_medium.log
timeout 200 python scripts/engine_lab.py --model small --layers 48 --reps 60 --check-reps 10 --modes 2 --waves 8,4 --chunks 8 --sleep 4 2>&1 | grep -v "f64 rest" | tee $O/engine_lab_small.log
timeout 200 python scripts/engine_lab.py --model large --layers 48 --reps 60 --check-reps 10 --modes 2 --waves 4 --chunks 8,16 --sleep 4 2>&1 | grep -v "f64 rest" | tee $O/engine_lab_large.log
