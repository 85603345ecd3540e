#!/bin/bash
# round 3, GPU session 1: full -m gpu suite on the new default (single-term + shift), today's prefill cost, bench A/B
set -u
OUT=gpurun_out/s1
mkdir -p $OUT
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1
echo "== pytest -m gpu" | tee $OUT/progress.log
timeout 1500 python -m pytest tests -m gpu -q -rP --maxfail=40 > $OUT/pytest_full.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/progress.log
grep -E "^\[(parity|near-tie|single-term)\]|passed|failed|^FAILED|^ERROR" $OUT/pytest_full.log > $OUT/pytest_summary.log
tail -5 $OUT/pytest_summary.log | tee -a $OUT/progress.log
echo "== prefill cost (8-positions-per-call path)" | tee -a $OUT/progress.log
timeout 600 python scripts/prefill_bench.py window melody --reps 3 > $OUT/prefill_old.jsonl 2> $OUT/prefill_old.err
cat $OUT/prefill_old.jsonl | tee -a $OUT/progress.log
echo "== bench default" | tee -a $OUT/progress.log
timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err
cat $OUT/bench_default.json | tee -a $OUT/progress.log
echo "== bench hi/lo (ACMI_LN_LO=1)" | tee -a $OUT/progress.log
ACMI_LN_LO=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_hilo.json 2> $OUT/bench_hilo.err
cat $OUT/bench_hilo.json | tee -a $OUT/progress.log
echo "== bench single-term unshifted (ACMI_LN_SHIFT=0)" | tee -a $OUT/progress.log
ACMI_LN_SHIFT=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_noshift.json 2> $OUT/bench_noshift.err
cat $OUT/bench_noshift.json | tee -a $OUT/progress.log
echo "== small B=1 bench line" | tee -a $OUT/progress.log
timeout 600 python bench.py --model facebook/musicgen-small --batch 1 --duration 10 --greedy --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_small.json 2> $OUT/bench_small.err
cat $OUT/bench_small.json | tee -a $OUT/progress.log
echo "== done" | tee -a $OUT/progress.log
