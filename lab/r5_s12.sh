#!/bin/bash
# round 5 session 12: the two option combinations that used to raise (rotary + left-padded streams; sum / interpolate after prepend)
set -u
O=$PWD/gpurun_out/r5s12; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_zz_options.py -q -x -m gpu -k "two_step or rotary or rope or fuser or options or prepend or melody or streaming or predictions" 2>&1 | tail -8 | tee $O/options_pytest.txt
