#!/bin/bash
# round 5 session 11: concurrent MALL prefetcher beside the decode GEMM chain (lab)
set -u
O=$PWD/gpurun_out/r5s11; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 python scripts/mall_prefetch_lab.py --wgs 8,16,32,64,128 2>&1 | grep -v amdgpu.ids | tee $O/mall_prefetch_lab.log
timeout 200 python scripts/mall_prefetch_lab.py --wgs 32,64 --frac 0.5 2>&1 | grep -v amdgpu.ids | tee $O/mall_prefetch_lab_half.log
