#!/bin/bash
# round 3, GPU session 3: deeper prefetch in the prefill GEMM, exact-count double buffering in the decode attention (A/B),
# LDS-DMA weight stream (A/B), 16-byte LSTM publishes
set -u
OUT=gpurun_out/s3
mkdir -p $OUT
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1
echo "== targeted tests" | tee $OUT/progress.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -rP --maxfail=30 -k "attn or attention or linear_big or prefill or lstm" > $OUT/pytest_kernels.log 2>&1
echo "kernels rc=$?" | tee -a $OUT/progress.log; tail -2 $OUT/pytest_kernels.log | tee -a $OUT/progress.log
timeout 1500 python -m pytest tests/test_gpu_models.py -q -rP --maxfail=30 -k "prefill or golden or encodec" > $OUT/pytest_models.log 2>&1
echo "models rc=$?" | tee -a $OUT/progress.log; tail -2 $OUT/pytest_models.log | tee -a $OUT/progress.log
echo "== the same GEMM tests with the LDS-DMA weight stream (ACMI_LIN_DMA=1)" | tee -a $OUT/progress.log
ACMI_LIN_DMA=1 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q --maxfail=30 -k "linear or folded or half_tile or pair or statistics or single_term or midsize or lm_text" > $OUT/pytest_dma.log 2>&1
echo "dma rc=$?" | tee -a $OUT/progress.log; tail -2 $OUT/pytest_dma.log | tee -a $OUT/progress.log
grep -hE "^\[(parity|near-tie|single-term)\]|^FAILED|^ERROR" $OUT/pytest_*.log > $OUT/pytest_summary.log
grep -hE "^FAILED|^ERROR|^\[parity\] prefill" $OUT/pytest_summary.log | head -40 | tee -a $OUT/progress.log
echo "== prefill cost" | tee -a $OUT/progress.log
timeout 600 python scripts/prefill_bench.py window melody --reps 3 > $OUT/prefill_big.jsonl 2> $OUT/prefill_big.err
cat $OUT/prefill_big.jsonl | tee -a $OUT/progress.log
for v in default ATTN_DB0 LIN_DMA1; do
  echo "== bench $v" | tee -a $OUT/progress.log
  case $v in
    default) E="" ;;
    ATTN_DB0) E="ACMI_ATTN_DB=0" ;;
    LIN_DMA1) E="ACMI_LIN_DMA=1" ;;
  esac
  env $E timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python -c "
import json,sys
d=json.load(open('$OUT/bench_$v.json')); r=d.get('roofline',{})
print('$v', 'RTF', d['value'], 'ms', d['ms_per_step'], 'gemm us', r.get('avg_launch_us'), 'frac', r.get('frac'))" | tee -a $OUT/progress.log
done
echo "== codec bench" | tee -a $OUT/progress.log
timeout 600 python scripts/codec_bench.py > $OUT/codec_bench.jsonl 2> $OUT/codec_bench.err
python -c "
import json
for l in open('$OUT/codec_bench.jsonl'):
    d=json.loads(l); print({k:d[k] for k in d if k in ('config','batch','seconds','encode_ms','decode_ms','audio_s_per_s')})" | tee -a $OUT/progress.log
echo "== rocprof kernel stats: default, one generate" | tee -a $OUT/progress.log
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
python scripts/short_names.py $OUT/bench_kernel_stats.csv | head -14 | tee -a $OUT/progress.log
echo "== rocprof kernel stats: prefill window" | tee -a $OUT/progress.log
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/scripts/prefill_bench.py window --reps 2 > $R/$OUT/prefill_under_rocprof.json 2> $R/$OUT/rocprof2.err)
cp $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) $OUT/prefill_kernel_stats.csv
python scripts/short_names.py $OUT/prefill_kernel_stats.csv | head -14 | tee -a $OUT/progress.log
echo "== done" | tee -a $OUT/progress.log
