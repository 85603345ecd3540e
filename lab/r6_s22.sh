#!/bin/bash
# round 6 session 22: final build (fused QKV + self-attention launch, conv_pw_kernel): whole GPU suite, smoke, codec lines, the bench line
set -u
O=$PWD/gpurun_out/r6s22; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -5 | tee $O/full_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 600 python scripts/codec_line.py 24k 1 10 > $O/line_cfg0.json 2> $O/line_cfg0.err; cut -c1-200 $O/line_cfg0.json
timeout 600 python scripts/codec_line.py 32k 8 30 > $O/line_codec32k.json 2> $O/line_codec32k.err; cut -c1-200 $O/line_codec32k.json
ACMI_BENCH_INSITU_KEEP=$O/cfg2_insitu_kernel_stats.csv timeout 1500 python bench.py --steps 3 --warmup 1 > $O/line_cfg2.json 2> $O/line_cfg2.err; tail -1 $O/line_cfg2.err; cut -c1-220 $O/line_cfg2.json
