#!/bin/bash
# round 5 session 23: the opt-in form of the score-folded cross-attention (decision on the host side only) + neighbours
set -u
O=$PWD/gpurun_out/r5s23; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -q -x -m gpu -k "score_folded or cross_fold or golden or midsize or philox" 2>&1 | tail -8 | tee $O/fold_optin_pytest_gpu.txt
