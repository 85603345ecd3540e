#!/bin/bash
# round 6 session 13: bench-format lines on the build with the fused QKV + self-attention launch (configs[1], [2], [3])
set -u
O=$PWD/gpurun_out/r6s13; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ACMI_BENCH_INSITU_KEEP=$O/cfg2_insitu_kernel_stats.csv timeout 1200 python bench.py --steps 3 --warmup 1 > $O/line_cfg2.json 2> $O/line_cfg2.err; tail -2 $O/line_cfg2.err; cut -c1-300 $O/line_cfg2.json
ACMI_BENCH_INSITU_KEEP=$O/cfg1_insitu_kernel_stats.csv timeout 600 python bench.py --steps 3 --warmup 1 --model facebook/musicgen-small --batch 1 --duration 10 --greedy --no-cpu-baseline > $O/line_cfg1.json 2> $O/line_cfg1.err; tail -2 $O/line_cfg1.err; cut -c1-300 $O/line_cfg1.json
ACMI_BENCH_INSITU_KEEP=$O/cfg3_insitu_kernel_stats.csv timeout 900 python bench.py --steps 2 --warmup 1 --model facebook/musicgen-large --batch 8 --duration 30 --no-cpu-baseline > $O/line_cfg3.json 2> $O/line_cfg3.err; tail -2 $O/line_cfg3.err; cut -c1-300 $O/line_cfg3.json
