#!/bin/bash
# round 6 session 6: the restructured sampler (tests, in-situ A/B against the round-5 numbers), MFMA / wave counters of the GEMM chain and the attention
set -u
O=$PWD/gpurun_out/r6s6; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "sample or embed or lm_step or pos_table" 2>&1 | tail -3 | tee $O/sampler_kernel_pytest.txt
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity_configs.py tests/test_gpu_zz_options.py -q -x -m gpu 2>&1 | tail -3 | tee $O/sampler_models_pytest.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $GRAFT_REPO_ROOT/scripts/short_generate.py facebook/musicgen-medium 8 8 > $O/short_generate.log 2>&1
python $GRAFT_REPO_ROOT/scripts/top_kernels.py $O/ks 14 | tee $O/short_generate_top_kernels.txt
find $O/ks -name "*kernel_stats.csv" -exec cp {} $O/short_generate_kernel_stats.csv \; ; rm -rf $O/ks
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- python $GRAFT_REPO_ROOT/scripts/dbg_chain.py > $O/pmc_chain.log 2>&1
find $O/pmc1 -name "*counter_collection.csv" -exec python $GRAFT_REPO_ROOT/scripts/summarize_pmc.py {} \; > $O/lin_chain_pmc_mfma.csv; rm -rf $O/pmc1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc2 -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py --contexts 751 > $O/pmc_attn.log 2>&1
find $O/pmc2 -name "*counter_collection.csv" -exec python $GRAFT_REPO_ROOT/scripts/summarize_pmc.py {} \; > $O/attn_pmc_waves_t751.csv; rm -rf $O/pmc2
cat $O/lin_chain_pmc_mfma.csv | head -40
