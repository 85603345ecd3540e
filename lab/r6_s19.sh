#!/bin/bash
# round 6 session 19: the fused launch with two 16-row blocks (<= 32 rows): bit identity, melody (configs[4]) and medium B = 16 A/B
set -u
O=$PWD/gpurun_out/r6s19; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -q -x -m gpu -k "fused_qkv" 2>&1 | tail -4 | tee $O/fused_pytest.txt
for mode in 0 1; do
  echo "ACMI_QKV_ATTN=$mode melody 16 x 30 s" | tee -a $O/ab.txt
  ACMI_QKV_ATTN=$mode timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --model facebook/musicgen-melody --batch 16 2>/dev/null | cut -c1-200 | tee -a $O/ab.txt
  echo "ACMI_QKV_ATTN=$mode medium 16 x 30 s" | tee -a $O/ab.txt
  ACMI_QKV_ATTN=$mode timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --batch 16 2>/dev/null | cut -c1-200 | tee -a $O/ab.txt
done
