#!/bin/bash
# round 5 session 1: the pending device test (non-delay codebook patterns) + a same-box headline number
set -u
O=$PWD/gpurun_out/r5s1; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_zz_options.py -q -x -m gpu 2>&1 | tail -5 | tee $O/options_pytest.txt
timeout 400 python bench.py --steps 3 --warmup 1 2>&1 | tail -2 | tee $O/bench_n1_s3.json
