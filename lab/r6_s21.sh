#!/bin/bash
# round 6 session 21: the one-launch resnet block (acmi_seanet_resblock): kernel test, codec suites, codec line A/B
set -u
O=$PWD/gpurun_out/r6s21; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "resblock or conv1d" 2>&1 | tail -5 | tee $O/resblock_pytest.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity_configs.py -q -x -m gpu -k "encodec or codec or stereo or 16" 2>&1 | tail -3 | tee $O/codec_pytest.txt
for rb in 0 1; do
  echo "ACMI_RESBLOCK=$rb" | tee -a $O/codec_ab.txt
  ACMI_RESBLOCK=$rb timeout 400 python scripts/codec_line.py 32k 8 30 --no-cpu 2>/dev/null | tee $O/codec32k_rb$rb.json | cut -c1-160 | tee -a $O/codec_ab.txt
done
