#!/bin/bash
# round 6 session 2: the hand-off lab with K of the second round staged in LDS, the A role delayed behind the weight stream, 8-wave attention
set -u
O=$PWD/gpurun_out/r6s2; mkdir -p $O
for t in 750 200 1400; do timeout 120 lab/qkv_attn_lab $t 20 48 2>&1 | tee $O/qkv_attn_lab_t$t.log; done
