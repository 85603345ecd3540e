#!/bin/bash
# round 6 session 11: 16-byte hand-off stores: bit identity (one case per layout), bench A/B, in-situ kernel breakdown of the fused form
set -u
O=$PWD/gpurun_out/r6s11; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_models.py -q -x -m gpu -k "fused_qkv and (True-8 or False-5)" 2>&1 | tail -5 | tee $O/fused_pytest.txt
for mode in 1 0 1; do
  ACMI_QKV_ATTN=$mode timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-230 | tee -a $O/bench_ab.txt
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $GRAFT_REPO_ROOT/scripts/short_generate.py facebook/musicgen-medium 8 30 > $O/short_generate.log 2>&1
python $GRAFT_REPO_ROOT/scripts/top_kernels.py $O/ks 10 | tee $O/fused_top_kernels_30s.txt
rm -rf $O/ks
