#!/bin/bash
set -u
O=$PWD/gpurun_out/s36
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
: > $O/progress.log
echo "== graph replay of the LSTM stack, per-layer launches (XCD-local form)" | tee -a $O/progress.log
ACMI_LSTM_WAVE=0 timeout 120 python lab/dbg_lstm_graph.py 2>&1 | grep -v amdgpu.ids | tee -a $O/progress.log
echo "== same, memory-side variant" | tee -a $O/progress.log
ACMI_LSTM_XCD=2 ACMI_LSTM_WAVE=0 timeout 120 python lab/dbg_lstm_graph.py 2>&1 | grep -v amdgpu.ids | tee -a $O/progress.log
