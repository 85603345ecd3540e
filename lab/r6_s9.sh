#!/bin/bash
set -u
O=$PWD/gpurun_out/r6s9; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -q -x -m gpu -k "fused_qkv" 2>&1 | tail -15 | tee $O/fused_pytest.txt
