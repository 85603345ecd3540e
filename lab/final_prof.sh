# final-build artefacts of the round (run on the GPU box from the repo root); outputs under gpurun_out/final
set -x
mkdir -p gpurun_out/final
O=$GRAFT_REPO_ROOT/gpurun_out/final
python bench.py > $O/bench_n1_default.json 2> $O/bench_n1_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python $GRAFT_REPO_ROOT/scripts/dbg_chain.py > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/scripts/summarize_pmc.py $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) > $O/lin_chain_pmc_$c.csv
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/pmc_mfma -- python $GRAFT_REPO_ROOT/scripts/dbg_chain.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/summarize_pmc.py $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) > $O/lin_chain_pmc_mfma.csv
cd $GRAFT_REPO_ROOT
python scripts/codec_bench.py > $O/codec_bench.jsonl 2> $O/codec_bench.err
python scripts/config_sweep.py > $O/config_sweep.log 2>&1
python scripts/microbench.py --gen 400 > $O/microbench.log 2>&1
