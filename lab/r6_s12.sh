#!/bin/bash
# round 6 session 12: one poller per workgroup (wave 0 + LDS hand-over) vs every wave polling
set -u
O=$PWD/gpurun_out/r6s12; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_models.py -q -x -m gpu -k "fused_qkv and (True-8 or False-5)" 2>&1 | tail -3 | tee $O/fused_pytest.txt
for po in 1 0 1 0; do
  echo "ACMI_QKV_POLL_ONE=$po" | tee -a $O/bench_ab.txt
  ACMI_QKV_POLL_ONE=$po timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200 | tee -a $O/bench_ab.txt
done
