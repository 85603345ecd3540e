#!/bin/bash
# round 6 session 23: conv_pw_kernel with packed fma (pairs of time steps)
set -u
O=$PWD/gpurun_out/r6s23; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "conv1d" 2>&1 | tail -3 | tee $O/conv_pytest.txt
timeout 400 python scripts/codec_line.py 32k 8 30 --no-cpu 2>/dev/null | tee $O/codec32k.json | cut -c1-160
