#!/bin/bash
# round 6 session 4: sampled-decision accounting at the medium bf16 geometry; bench-format lines for every BASELINE config
set -u
O=$PWD/gpurun_out/r6s4; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_configs.py -q -x -m gpu -k "sampled_decisions" -s 2>&1 | grep -v "^$" | tail -8 | tee $O/sampled_parity_pytest.txt
ACMI_BENCH_INSITU_KEEP=$O/cfg2_insitu_kernel_stats.csv timeout 900 python bench.py --steps 3 --warmup 1 > $O/line_cfg2.json 2> $O/line_cfg2.err; tail -2 $O/line_cfg2.err; cut -c1-400 $O/line_cfg2.json
ACMI_BENCH_INSITU_KEEP=$O/cfg1_insitu_kernel_stats.csv timeout 600 python bench.py --steps 3 --warmup 1 --model facebook/musicgen-small --batch 1 --duration 10 --greedy --no-cpu-baseline > $O/line_cfg1.json 2> $O/line_cfg1.err; tail -2 $O/line_cfg1.err; cut -c1-300 $O/line_cfg1.json
ACMI_BENCH_INSITU_KEEP=$O/cfg3_insitu_kernel_stats.csv timeout 900 python bench.py --steps 2 --warmup 1 --model facebook/musicgen-large --batch 8 --duration 30 --no-cpu-baseline > $O/line_cfg3.json 2> $O/line_cfg3.err; tail -2 $O/line_cfg3.err; cut -c1-300 $O/line_cfg3.json
ACMI_BENCH_INSITU_KEEP=$O/cfg4_insitu_kernel_stats.csv timeout 900 python bench.py --steps 2 --warmup 1 --model facebook/musicgen-melody --batch 16 --duration 30 --no-cpu-baseline > $O/line_cfg4.json 2> $O/line_cfg4.err; tail -2 $O/line_cfg4.err; cut -c1-300 $O/line_cfg4.json
timeout 600 python scripts/codec_line.py 24k 1 10 > $O/line_cfg0.json 2> $O/line_cfg0.err; tail -2 $O/line_cfg0.err; cut -c1-300 $O/line_cfg0.json
timeout 600 python scripts/codec_line.py 32k 8 30 > $O/line_codec32k.json 2> $O/line_codec32k.err; tail -2 $O/line_codec32k.err; cut -c1-300 $O/line_codec32k.json
