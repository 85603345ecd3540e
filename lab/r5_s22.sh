#!/bin/bash
# round 5 session 22: the score-folded cross-attention at FEW conditioned rows (tables smaller than the matrices they replace)
set -u
O=$PWD/gpurun_out/r5s22; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for mode in 1 0; do
  echo "== ACMI_CROSS_FOLD=$mode"
  ACMI_CROSS_FOLD=$mode timeout 400 python scripts/config_sweep.py fold 2>&1 | grep -v amdgpu.ids
done | tee $O/fold_small_batches.txt
