#!/bin/bash
# round 5 session 6: the new GPU tests (Philox replay, engine op, configs[3] flags through two processes) + the bench line with
# the in-situ kernel statistics
set -u
O=$PWD/gpurun_out/r5s6; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_distributed.py -q -x -m gpu -k "philox or ffn_engine or configs3 or two_processes" 2>&1 | tail -15 | tee $O/new_tests_pytest.txt
ACMI_BENCH_INSITU_KEEP=$O/bench_insitu_kernel_stats.csv timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
cut -c1-2500 $O/bench_n1.json; tail -3 $O/bench_n1.err
