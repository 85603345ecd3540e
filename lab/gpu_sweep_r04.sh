#!/bin/bash
# round 4, last session: re-check of the wave-count knobs on the final kernels (same box, one call)
mkdir -p gpurun_out
{
echo "== GEMM chain (scripts/dbg_chain.py): us per launch"
for v in "" "ACMI_LIN_FPW=6" "ACMI_LIN_FPW=24" "ACMI_LIN_NW=4"; do
  echo "-- ${v:-default}"
  env $v timeout 60 python scripts/dbg_chain.py 2>&1 | grep "us/launch"
done
echo "== self-attention decode (scripts/attn_bench.py)"
for v in "" "ACMI_ATTN_NW=2"; do
  echo "-- ${v:-default}"
  env $v timeout 60 python scripts/attn_bench.py 2>&1 | tail -3
done
} > gpurun_out/r04_wave_knob_sweep.log 2>&1
cat gpurun_out/r04_wave_knob_sweep.log
