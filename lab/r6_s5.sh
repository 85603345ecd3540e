#!/bin/bash
# round 6 session 5: prefetch workgroups inside the QKV launch (no flag): the attention's first positions warmed in the consumer XCD's L2
set -u
O=$PWD/gpurun_out/r6s5; mkdir -p $O
for t in 750 1400 200; do timeout 120 lab/qkv_attn_lab $t 20 48 2>&1 | tee $O/qkv_attn_lab_t$t.log; done
