#!/bin/bash
set -u
O=$PWD/gpurun_out/s38
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
: > $O/progress.log
echo "== lstm / codec tests after the fill-kernel change" | tee -a $O/progress.log
timeout 800 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_parity_configs.py tests/test_gpu_musicgen_api.py -q -x -m gpu -k "encodec or lstm or codec or compression or stereo or graph or seanet or epic or audiogen" 2>&1 | tail -4 | tee -a $O/progress.log
echo "== smoke" | tee -a $O/progress.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a $O/progress.log
