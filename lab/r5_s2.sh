#!/bin/bash
# round 5 session 2: first run of the persistent FFN engine (parity, determinism, us per layer, in-kernel timeline)
set -u
O=$PWD/gpurun_out/r5s2; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 420 python scripts/engine_lab.py --model medium --layers 48 --reps 100 --check-reps 20 --modes 0,1,2 --waves 4,8 --trace $O/engine_timeline 2>&1 | tee $O/engine_lab_medium.log
