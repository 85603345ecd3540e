#!/bin/bash
# round 5 session 21: where the score-folded step spends its time: kernel stats of a short generate in both modes + the cost of the tables
set -u
O=$PWD/gpurun_out/r5s21; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd /tmp
for mode in 1 0; do
  rm -rf /tmp/prof_$mode
  ACMI_CROSS_FOLD=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python $GRAFT_REPO_ROOT/scripts/short_generate.py facebook/musicgen-medium 8 8 > /dev/null 2>&1
  echo "== ACMI_CROSS_FOLD=$mode"; python $GRAFT_REPO_ROOT/scripts/top_kernels.py /tmp/prof_$mode 14
done 2>&1 | tee $O/kernel_stats_both_modes.txt
cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | tee $O/table_cost.txt
import time, torch
from audiocraft_amd.models.musicgen import MusicGen
m = MusicGen.get_random_init('facebook/musicgen-medium', 'cuda', torch.bfloat16, text_len=16, seed=0)
m.set_generation_params(use_sampling=True, top_k=250, duration=2.0)
d = [f"synthetic prompt {i}" for i in range(8)]
m.generate(d)
lm = m.lm
run = lm._run
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    lm._build_cross_fold(run, 8, 16)
torch.cuda.synchronize()
print(f"_build_cross_fold: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per generate")
PY
