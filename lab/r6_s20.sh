#!/bin/bash
# round 6 session 20: the pointwise resnet-block convolution (conv_pw_kernel): kernel tests, codec suites, codec line A/B
set -u
O=$PWD/gpurun_out/r6s20; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "conv1d" 2>&1 | tail -3 | tee $O/conv_pytest.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity_configs.py -q -x -m gpu -k "encodec or codec or stereo or 16" 2>&1 | tail -3 | tee $O/codec_pytest.txt
for pw in 0 1; do
  echo "ACMI_CONV_PW=$pw" | tee -a $O/codec_ab.txt
  ACMI_CONV_PW=$pw timeout 400 python scripts/codec_line.py 32k 8 30 --no-cpu 2>/dev/null | tee $O/codec32k_pw$pw.json | cut -c1-160 | tee -a $O/codec_ab.txt
  ACMI_CONV_PW=$pw timeout 400 python scripts/codec_line.py 24k 1 10 --no-cpu 2>/dev/null | tee $O/codec24k_pw$pw.json | cut -c1-160 | tee -a $O/codec_ab.txt
done
