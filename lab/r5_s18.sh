#!/bin/bash
# round 5 session 18: two independent decode chains on two streams of ONE process against one chain of twice the batch
set -u
O=$PWD/gpurun_out/r5s18; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 python scripts/two_streams_lab.py 8 10 2>&1 | tail -12 | tee $O/two_streams_lab.log
