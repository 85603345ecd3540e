#!/bin/bash
# round 5 session 4: engine v3 (replicated flags, paced polls)
set -u
O=$PWD/gpurun_out/r5s4; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python scripts/engine_lab.py --model medium --layers 48 --reps 60 --check-reps 10 --modes 2 --waves 8,4 --chunks 2,4,8 --epi 0 --sleep 0,1,4 --trace $O/tl 2>&1 | grep -v "^wave,\|^control,\|^compute0,\|^all,\|f64 rest\|vs the launch" | tee $O/engine_lab_medium.log
