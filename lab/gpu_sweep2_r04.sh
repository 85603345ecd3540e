#!/bin/bash
# round 4, last session: the structural A/B switches of the decode step re-measured on the final kernels: MusicGen-medium,
# 8 prompts x 10 s (503 positions), one box, one call; ms / position is the comparable figure
mkdir -p gpurun_out
{
for v in "" "ACMI_LIN_WIDE=0" "ACMI_FFN2_HALF=0" "ACMI_CROSS_FUSED=0" "ACMI_LN_GRAM=0" ""; do
  echo "-- ${v:-default}"
  env $v timeout 60 python -c "import scripts.config_sweep as c; c.run('facebook/musicgen-medium', 8, 10, reps=2)" 2>&1 | grep RTF
done
} > gpurun_out/r04_structural_switch_sweep.log 2>&1
cat gpurun_out/r04_structural_switch_sweep.log
