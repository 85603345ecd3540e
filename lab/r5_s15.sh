#!/bin/bash
# round 5 session 15: qk_layer_norm through the one-forward prefill
set -u
O=$PWD/gpurun_out/r5s15; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_zz_options.py tests/test_gpu_models.py -q -x -m gpu -k "options or qk or prefill or golden" 2>&1 | tail -8 | tee $O/qk_ln_prefill_pytest.txt
