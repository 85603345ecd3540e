#!/bin/bash
# round 3, GPU session 2: prefill as one forward (kernels + model), double-buffered decode attention, bench A/B
set -u
OUT=gpurun_out/s2
mkdir -p $OUT
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1
echo "== targeted tests" | tee $OUT/progress.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -rP --maxfail=30 -k "attn or attention or linear_big or prefill or single_term or query" > $OUT/pytest_kernels.log 2>&1
echo "kernels rc=$?" | tee -a $OUT/progress.log; tail -3 $OUT/pytest_kernels.log | tee -a $OUT/progress.log
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_musicgen_api.py -q -rP --maxfail=30 -k "not small_architecture" > $OUT/pytest_models.log 2>&1
echo "models rc=$?" | tee -a $OUT/progress.log; tail -3 $OUT/pytest_models.log | tee -a $OUT/progress.log
timeout 1500 python -m pytest tests/test_gpu_parity_configs.py -q -rP --maxfail=30 -k "medium_bf16 or melody or large_bf16_late" > $OUT/pytest_parity.log 2>&1
echo "parity rc=$?" | tee -a $OUT/progress.log; tail -3 $OUT/pytest_parity.log | tee -a $OUT/progress.log
grep -hE "^\[(parity|near-tie|single-term)\]|^FAILED|^ERROR" $OUT/pytest_*.log > $OUT/pytest_summary.log
grep -hE "^FAILED|^ERROR|^\[parity\] prefill" $OUT/pytest_summary.log | head -40 | tee -a $OUT/progress.log
echo "== prefill cost: one forward vs chunk path" | tee -a $OUT/progress.log
timeout 600 python scripts/prefill_bench.py window melody --reps 3 > $OUT/prefill_big.jsonl 2> $OUT/prefill_big.err
cat $OUT/prefill_big.jsonl | tee -a $OUT/progress.log
echo "== bench default" | tee -a $OUT/progress.log
timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err
cat $OUT/bench_default.json | tee -a $OUT/progress.log
echo "== bench hi/lo (ACMI_LN_LO=1)" | tee -a $OUT/progress.log
ACMI_LN_LO=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_hilo.json 2> $OUT/bench_hilo.err
cat $OUT/bench_hilo.json | tee -a $OUT/progress.log
echo "== rocprof kernel stats of one default generate" | tee -a $OUT/progress.log
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
python scripts/short_names.py $OUT/bench_kernel_stats.csv | head -30 | tee -a $OUT/progress.log
echo "== done" | tee -a $OUT/progress.log
