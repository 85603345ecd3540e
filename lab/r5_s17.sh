#!/bin/bash
# round 5 session 17: post-norm layers (norm_first=False) on the device: reference golden + oracle at d = 256, both prefill forms
set -u
O=$PWD/gpurun_out/r5s17; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_zz_options.py -q -x -m gpu -k "post_norm or midsize or layer_norm_rows" 2>&1 | tail -30 | tee $O/post_norm_pytest.txt
