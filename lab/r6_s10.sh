#!/bin/bash
# round 6 session 10: the whole GPU suite with the fused QKV + self-attention launch ON; A/B of the other model sizes
set -u
O=$PWD/gpurun_out/r6s10; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ACMI_QKV_ATTN=1 timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -6 | tee $O/full_pytest_gpu_fused_on.txt
for mode in 0 1; do
  ACMI_QKV_ATTN=$mode timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --model facebook/musicgen-large 2>/dev/null | cut -c1-230 | tee $O/bench_large_fused_$mode.json
  ACMI_QKV_ATTN=$mode timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --model facebook/musicgen-small --batch 1 --duration 10 --greedy 2>/dev/null | cut -c1-230 | tee $O/bench_small_fused_$mode.json
  ACMI_QKV_ATTN=$mode timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --batch 4 2>/dev/null | cut -c1-230 | tee $O/bench_medium_b4_fused_$mode.json
done
