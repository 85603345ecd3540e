#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/splitk; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mbd.py -q -x -k "conv or mbd or unet" 2>&1 | tail -3 > $O/pytest.log; cat $O/pytest.log
for s in 1 3 5 10; do timeout 200 python scripts/mbd_bench.py --seconds $s --reps 5 2>/dev/null | tee $O/mbd_${s}s.json | cut -c1-200; done
timeout 200 python scripts/codec_bench.py 2>/dev/null | tee $O/codec_bench.jsonl | cut -c1-330
