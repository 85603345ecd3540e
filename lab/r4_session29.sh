#!/bin/bash
# round 4, session 29: 256 x 128 tiles, second workgroup of a CU started late (ACMI_BIG_STAGGER cycles; 0 = off)
set -u
O=$PWD/gpurun_out/s29
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "tests: 256 x 128 forced (default stagger)"
ACMI_BIG_TILE=2 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "linear_big" 2>&1 | tail -3 | tee -a $O/progress.log
log "GEMM alone: tile 1 / tile 2 stagger 0 / default / 24000 / 70000"
ACMI_BIG_TILE=1 timeout 300 python scripts/big_gemm_bench.py 2> $O/err_t1 | tee -a $O/progress.log
for sg in 0 -1 24000 70000; do
  echo "-- stagger $sg" | tee -a $O/progress.log
  if [ $sg = -1 ]; then ACMI_BIG_TILE=2 timeout 300 python scripts/big_gemm_bench.py 2> $O/err_t2 | tee -a $O/progress.log
  else ACMI_BIG_STAGGER=$sg ACMI_BIG_TILE=2 timeout 300 python scripts/big_gemm_bench.py 2> $O/err_t2 | tee -a $O/progress.log; fi
done
log "timeline 256 x 128, default stagger"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_bigtrace.so ACMI_BIG_TILE=2 timeout 300 python scripts/big_gemm_bench.py --trace --reps 3 2> $O/err_tr | tee -a $O/progress.log
log "done"
