#!/bin/bash
# round 3, GPU session 4: attention kernel A/B (round-2 kernel vs single set vs two sets), conv kernel rewrite
set -u
OUT=gpurun_out/s4
mkdir -p $OUT
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1
echo "== attention microbench" | tee $OUT/progress.log
ACMI_LIB=$PWD/lab/libacmi_oldattn.so python scripts/attn_bench.py > $OUT/attn_r02kernel.log 2>&1; tail -1 $OUT/attn_r02kernel.log | tee -a $OUT/progress.log
ACMI_ATTN_DB=0 python scripts/attn_bench.py > $OUT/attn_db0.log 2>&1; tail -1 $OUT/attn_db0.log | tee -a $OUT/progress.log
ACMI_ATTN_DB=1 python scripts/attn_bench.py > $OUT/attn_db1.log 2>&1; tail -1 $OUT/attn_db1.log | tee -a $OUT/progress.log
echo "== conv / codec tests" | tee -a $OUT/progress.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_parity_configs.py -q -rP --maxfail=30 -k "conv or encodec or lstm or codec" > $OUT/pytest_codec.log 2>&1
echo "codec tests rc=$?" | tee -a $OUT/progress.log; tail -2 $OUT/pytest_codec.log | tee -a $OUT/progress.log
grep -hE "^FAILED|^ERROR" $OUT/pytest_codec.log | head -20 | tee -a $OUT/progress.log
echo "== codec bench" | tee -a $OUT/progress.log
timeout 600 python scripts/codec_bench.py > $OUT/codec_bench.jsonl 2> $OUT/codec_bench.err
python -c "
import json
for l in open('$OUT/codec_bench.jsonl'):
    d=json.loads(l); print(d['config'], 'enc', d['encode']['ms'], d['encode']['f32_mfma_frac'], 'dec', d['decode']['ms'], d['decode']['f32_mfma_frac'])" | tee -a $OUT/progress.log
for n in 1 2 4; do
ACMI_CONV_NTQ=$n timeout 600 python scripts/codec_bench.py > $OUT/codec_bench_ntq$n.jsonl 2> /dev/null
python -c "
import json
for l in open('$OUT/codec_bench_ntq$n.jsonl'):
    d=json.loads(l); print('NTQ=$n', d['config'][:28], 'enc', d['encode']['ms'], 'dec', d['decode']['ms'])" | tee -a $OUT/progress.log
done
echo "== done" | tee -a $OUT/progress.log
