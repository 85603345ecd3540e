#!/bin/bash
# round 5 session 20: score-folded cross-attention: kernel test, A/B test, LM goldens + geometry parity; then a same-box A/B of the generate
set -u
O=$PWD/gpurun_out/r5s20; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ACMI_BENCH_INSITU=0 ACMI_BENCH_PMC=0
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity_configs.py -q -x -m gpu -k "score_folded or golden or midsize or medium or philox" 2>&1 | tail -15 | tee $O/models.txt
F="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
timeout 300 python bench.py $F 2>$O/fold.err | tee $O/bench_fold.json
ACMI_CROSS_FOLD=0 timeout 300 python bench.py $F 2>$O/sep.err | tee $O/bench_separate.json
tail -2 $O/fold.err
