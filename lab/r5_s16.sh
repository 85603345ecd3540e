#!/bin/bash
# round 5 session 16: (a) any-order launch primitive (no AQL barrier bit?), (b) two concurrent generate chains on one device
# (2 processes x B = 4 against 1 x B = 8; 2 x 8 against 1 x 16): does a second dependency chain fill the HBM-idle launch edges?
set -u
O=$PWD/gpurun_out/r5s16; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ACMI_BENCH_INSITU=0 ACMI_BENCH_PMC=0
timeout 120 lab/anyorder_lab 2>&1 | tee $O/anyorder_lab.log
F="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 300 python bench.py $F --batch 8 2>$O/b8.err | tee $O/one_chain_b8.json
timeout 300 python bench.py $F --batch 4 2>$O/b4.err | tee $O/one_chain_b4.json
ACMI_ALLOW_SHARED_DEVICE=1 ACMI_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 $F --batch 4 2>$O/2x4.err | tee $O/two_chains_b4.json
ACMI_ALLOW_SHARED_DEVICE=1 ACMI_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 $F --batch 8 2>$O/2x8.err | tee $O/two_chains_b8.json
tail -3 $O/*.err
