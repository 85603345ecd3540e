#!/bin/bash
# conv staging through registers: parity tests, codec + MBD cost
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/conv1; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mbd.py tests/test_gpu_models.py -q -x -k "conv or seanet or mbd or unet or codec or encodec or lstm or reverse or split or group or compression or multiband" 2>&1 | tail -8 > $O/pytest.log
timeout 300 python scripts/codec_bench.py > $O/codec_bench.jsonl 2> $O/codec_bench.err
timeout 300 python scripts/codec_layers.py decode > $O/codec_layers_decode.log 2>&1
timeout 300 python scripts/mbd_bench.py --seconds 10 > $O/mbd_bench_10s.json 2> $O/mbd.err
timeout 300 python scripts/mbd_bench.py --seconds 30 --batch 2 > $O/mbd_bench_30s_b2.json 2>> $O/mbd.err
cat $O/pytest.log; cut -c1-700 $O/codec_bench.jsonl; cat $O/mbd_bench_10s.json $O/mbd_bench_30s_b2.json; cut -c1-160 $O/codec_layers_decode.log
