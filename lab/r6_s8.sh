#!/bin/bash
# round 6 session 8: the fused QKV + self-attention launch in the product: parity, then same-box A/B of the bench workload
set -u
O=$PWD/gpurun_out/r6s8; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_models.py -q -x -m gpu -k "fused_qkv" 2>&1 | tail -15 | tee $O/fused_pytest.txt
for mode in 0 1; do
  ACMI_QKV_ATTN=$mode timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-330 | tee $O/bench_fused_$mode.json
done
ACMI_QKV_ATTN=1 ACMI_QKV_STAGE_K=0 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-330 | tee $O/bench_fused_1_nostage.json
