#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/graph1; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_mbd.py tests/test_gpu_musicgen_api.py -q -x -k "encodec or lstm or codec or compression or rvq or stereo or graph" 2>&1 | tail -12 > $O/pytest.log
cat $O/pytest.log
timeout 200 python scripts/codec_bench.py 2>$O/cb.err | cut -c1-330 > $O/codec_bench.txt; cat $O/codec_bench.txt; tail -3 $O/cb.err
