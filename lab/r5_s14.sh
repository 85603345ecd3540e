#!/bin/bash
# round 5 session 14: the other BASELINE.json configurations + batch scaling on the round's build (no regressions)
set -u
O=$PWD/gpurun_out/r5s14; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python scripts/config_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/config_sweep.log
