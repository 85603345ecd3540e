#!/bin/bash
# round 3, GPU session 5: attention back on one register set, conv variants (plain / parity inner loop, few-output kernel),
# prefill cross-attention through the MFMA attention kernel
set -u
OUT=gpurun_out/s5
mkdir -p $OUT
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1
echo "== attention microbench" | tee $OUT/progress.log
ACMI_LIB=$PWD/lab/libacmi_oldattn.so python scripts/attn_bench.py > $OUT/attn_r02kernel.log 2>&1; tail -1 $OUT/attn_r02kernel.log | tee -a $OUT/progress.log
python scripts/attn_bench.py > $OUT/attn_new.log 2>&1; tail -1 $OUT/attn_new.log | tee -a $OUT/progress.log
echo "== tests" | tee -a $OUT/progress.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_parity_configs.py -q -rP --maxfail=30 -k "conv or encodec or lstm or codec or attn or attention or prefill or golden" > $OUT/pytest.log 2>&1
echo "tests rc=$?" | tee -a $OUT/progress.log; tail -2 $OUT/pytest.log | tee -a $OUT/progress.log
grep -hE "^FAILED|^ERROR" $OUT/pytest.log | head -20 | tee -a $OUT/progress.log
ACMI_CONV_PARITY=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q --maxfail=10 -k "conv" > $OUT/pytest_parity_conv.log 2>&1
echo "parity-conv tests rc=$?" | tee -a $OUT/progress.log; tail -1 $OUT/pytest_parity_conv.log | tee -a $OUT/progress.log
echo "== codec bench variants" | tee -a $OUT/progress.log
for v in default PARITY1 FEWOUT0 NTQ1; do
  case $v in
    default) E="" ;;
    PARITY1) E="ACMI_CONV_PARITY=1" ;;
    FEWOUT0) E="ACMI_CONV_FEWOUT=0" ;;
    NTQ1) E="ACMI_CONV_NTQ=1" ;;
  esac
  env $E timeout 600 python scripts/codec_bench.py > $OUT/codec_bench_$v.jsonl 2> /dev/null
  python -c "
import json
for l in open('$OUT/codec_bench_$v.jsonl'):
    d=json.loads(l); print('$v', d['config'][:30], 'enc', d['encode']['ms'], d['encode']['f32_mfma_frac'], 'dec', d['decode']['ms'], d['decode']['f32_mfma_frac'])" | tee -a $OUT/progress.log
done
echo "== prefill cost" | tee -a $OUT/progress.log
timeout 600 python scripts/prefill_bench.py window melody --reps 3 > $OUT/prefill_big.jsonl 2> $OUT/prefill_big.err
cat $OUT/prefill_big.jsonl | tee -a $OUT/progress.log
echo "== bench default" | tee -a $OUT/progress.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "
import json
d=json.load(open('$OUT/bench_default.json')); r=d.get('roofline',{})
print('default RTF', d['value'], 'ms', d['ms_per_step'], 'gemm us', r.get('avg_launch_us'), 'frac', r.get('frac'))" | tee -a $OUT/progress.log
echo "== done" | tee -a $OUT/progress.log
