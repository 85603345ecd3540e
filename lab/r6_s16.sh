#!/bin/bash
# round 6 session 16: one poller per workgroup + length-first A role: bit identity, the bench line with the fused launch's live PMC traffic at the bench's own contexts
set -u
O=$PWD/gpurun_out/r6s16; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_models.py -q -x -m gpu -k "fused_qkv" 2>&1 | tail -3 | tee $O/fused_pytest.txt
ACMI_BENCH_INSITU_KEEP=$O/cfg2_insitu_kernel_stats.csv timeout 1500 python bench.py --steps 3 --warmup 1 > $O/line_cfg2.json 2> $O/line_cfg2.err; tail -1 $O/line_cfg2.err; cut -c1-220 $O/line_cfg2.json
