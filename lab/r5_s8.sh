#!/bin/bash
# round 5 session 8: self-attention: 2 waves x 16 positions per lane against 4 waves x 8 (same bytes in flight, half the waves)
set -u
O=$PWD/gpurun_out/r5s8; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
L=$PWD/audiocraft_amd/csrc/libacmi_ni16.so
( echo "== default (NI 8, 4 waves)"; timeout 200 python scripts/attn_bench.py | tail -10
  echo "== NI 8, 2 waves"; ACMI_ATTN_NW=2 timeout 200 python scripts/attn_bench.py | tail -10
  echo "== NI 16, 2 waves"; ACMI_LIB=$L ACMI_ATTN_NW=2 timeout 200 python scripts/attn_bench.py | tail -10
  echo "== NI 16, 4 waves"; ACMI_LIB=$L timeout 200 python scripts/attn_bench.py | tail -10 ) 2>&1 | grep -v amdgpu.ids | tee $O/attn_ni_sweep.log
