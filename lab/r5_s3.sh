#!/bin/bash
# round 5 session 3: engine v2 (metered weight stream, op1's epilogue on the compute waves): chunk sweep + timelines
set -u
O=$PWD/gpurun_out/r5s3; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 500 python scripts/engine_lab.py --model medium --layers 48 --reps 60 --check-reps 10 --modes 2 --waves 4,8 --chunks 1,2,4,8,16 --epi 0 --trace $O/tl 2>&1 | grep -v "^wave,\|^control,\|^compute0,\|^all," | tee $O/engine_lab_medium.log
timeout 200 python scripts/engine_lab.py --model medium --layers 48 --reps 60 --check-reps 10 --modes 0 --waves 8 --chunks 2,4 --epi 0,2 2>&1 | tee $O/engine_lab_medium_b.log
