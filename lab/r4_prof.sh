#!/bin/bash
# round 4: artefacts of the current build (run on the GPU box from the repo root); outputs under gpurun_out/prof
set -u
O=$PWD/gpurun_out/prof
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "QKV / linear kernel tests + LM goldens"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "linear or folded or lm_ or golden or midsize" 2>&1 | tail -4 | tee -a $O/progress.log
log "bench, default flags (with the CPU leg)"
timeout 900 python bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err
cut -c1-200 $O/bench_n1_default.json | tee -a $O/progress.log
log "rocprofv3 kernel stats of one generate"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python scripts/short_names.py $O/bench_kernel_stats.csv | head -14 | tee -a $O/progress.log
log "PMC passes over the GEMM chain"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/scripts/dbg_chain.py > /dev/null 2>&1)
  python scripts/summarize_pmc.py $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) > $O/lin_chain_pmc_$c.csv
done
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/pmc_mfma -- python $R/scripts/dbg_chain.py > /dev/null 2>&1)
python scripts/summarize_pmc.py $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) > $O/lin_chain_pmc_mfma.csv
cat $O/lin_chain_pmc_FETCH_SIZE.csv | head -12 | tee -a $O/progress.log
log "timeline"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_trace.so timeout 300 python scripts/lin_timeline.py --out $O/lin_timeline.csv > $O/lin_timeline.log 2>&1
tail -7 $O/lin_timeline.log | tee -a $O/progress.log
log "configs[1] bench line (small, B = 1, greedy)"
timeout 600 python bench.py --model facebook/musicgen-small --batch 1 --duration 10 --greedy --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_small_b1.json 2> $O/bench_small_b1.err
cut -c1-200 $O/bench_small_b1.json | tee -a $O/progress.log
log "config sweep"
timeout 900 python scripts/config_sweep.py > $O/config_sweep.log 2>&1
tail -12 $O/config_sweep.log | tee -a $O/progress.log
log "done"
