#!/bin/bash
# round 5 session 19: stability of the two-thread generate test (10 runs) + the LM model tests on the 0.1.9 build
set -u
O=$PWD/gpurun_out/r5s19; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python -m pytest tests/test_gpu_models.py -q -x -m gpu -k "two_models_generate_concurrently" 2>&1 | tail -1
done | tee $O/two_thread_test_x10.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_zz_options.py -q -x -m gpu 2>&1 | tail -5 | tee $O/models_options_pytest.txt
