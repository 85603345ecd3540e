#!/bin/bash
# round 6 session 15: A role with the length first (no speculative over-read): parity, A/B, the bench line with live PMC traffic of the fused launch
set -u
O=$PWD/gpurun_out/r6s15; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_models.py -q -x -m gpu -k "fused_qkv" 2>&1 | tail -3 | tee $O/fused_pytest.txt
ACMI_BENCH_INSITU_KEEP=$O/cfg2_insitu_kernel_stats.csv timeout 1200 python bench.py --steps 3 --warmup 1 > $O/line_cfg2.json 2> $O/line_cfg2.err; tail -1 $O/line_cfg2.err; cut -c1-220 $O/line_cfg2.json
ACMI_QKV_ATTN=0 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200 | tee $O/bench_sep.json
