#!/bin/bash
set -u
O=$PWD/gpurun_out/s37
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
: > $O/progress.log
echo "== graph replay of the LSTM stack, per-layer launches (XCD-local form), arming kernel" | tee -a $O/progress.log
ACMI_LSTM_WAVE=0 timeout 120 python lab/dbg_lstm_graph.py 2>&1 | grep -v amdgpu.ids | tee -a $O/progress.log
echo "== codec bench, default" | tee -a $O/progress.log
timeout 300 python scripts/codec_bench.py 2> $O/codec_bench.err | cut -c1-600 | tee -a $O/progress.log
echo "== codec bench, ACMI_LSTM_WAVE=0" | tee -a $O/progress.log
ACMI_LSTM_WAVE=0 timeout 300 python scripts/codec_bench.py 2> $O/codec_bench0.err | cut -c1-600 | tee -a $O/progress.log
tail -3 $O/codec_bench0.err
