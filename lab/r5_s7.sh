#!/bin/bash
# round 5 session 7: cross_q_kernel (the decode step's cross-attention as its own kernel): tests + in-situ A/B
set -u
O=$PWD/gpurun_out/r5s7; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_zz_options.py -q -x -m gpu -k "attn or pair or hook or lm_ or cross or two_step or double or golden" 2>&1 | tail -6 | tee $O/attn_tests_pytest.txt
for v in 1 0; do
  cd /tmp; rm -rf /tmp/prof_$v; ACMI_CROSSQ=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $O/../../scripts/short_generate.py facebook/musicgen-medium 8 8 > /dev/null 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_crossq$v.csv
  echo "ACMI_CROSSQ=$v"; grep -E "attn_decode_kernel|cross_q_kernel|lin_pair|1, 0, 1, 8, false" $f | awk -F'","' '{print substr($1,1,70), $2, $4}'
done 2>&1 | tee $O/crossq_ab.txt
