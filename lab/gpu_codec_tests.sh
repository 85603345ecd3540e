#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ctests; mkdir -p $O
cd $R
timeout 800 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_mbd.py tests/test_gpu_parity_configs.py tests/test_gpu_musicgen_api.py tests/test_gpu_chroma.py -q -x -k "encodec or lstm or codec or compression or rvq or stereo or graph or mbd or unet or conv or audiogen or api or generate" 2>&1 | tail -6 > $O/pytest.log
cat $O/pytest.log
