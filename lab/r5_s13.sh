#!/bin/bash
# round 5 session 13: persistent time-paced MALL prefetcher beside the GEMM chain (lab)
set -u
O=$PWD/gpurun_out/r5s13; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 500 python scripts/mall_prefetch_lab.py --wgs '' --paced 64:4:100,64:4:50,64:4:25,128:4:100,128:4:50,32:4:50,128:12:100,64:12:50 2>&1 | grep -v amdgpu.ids | tee $O/mall_paced_prefetch_lab.log
