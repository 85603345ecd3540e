#!/bin/bash
# round 6 session 1: QKV -> self-attention as one launch with a per-(row, head) hand-off (toy lab) + same-box bench baseline
set -u
O=$PWD/gpurun_out/r6s1; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for t in 750 200 1400; do timeout 120 lab/qkv_attn_lab $t 20 48 2>&1 | tee $O/qkv_attn_lab_t$t.log; done
timeout 400 python bench.py --steps 3 --warmup 1 2>$O/bench.err | tail -1 | tee $O/bench_n1_s3.json | cut -c1-600
