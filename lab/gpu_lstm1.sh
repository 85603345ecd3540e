#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lstm1; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "lstm" 2>&1 | tail -12 > $O/pytest.log
cat $O/pytest.log



timeout 200 python scripts/codec_bench.py 2>/dev/null | cut -c1-330 > $O/codec_bench.txt; cat $O/codec_bench.txt
