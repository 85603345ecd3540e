#!/bin/bash
# round 4, session 9: prefill GEMM with LDS-DMA staging vs register staging
set -u
O=$PWD/gpurun_out/s9
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "prefill tests (DMA default)"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "big or prefill or golden or midsize or streaming" 2>&1 | tail -5 | tee -a $O/progress.log
log "prefill bench: DMA, register staging"
timeout 600 python scripts/prefill_bench.py window melody > $O/prefill_dma.jsonl 2> $O/prefill_dma.err; cat $O/prefill_dma.jsonl | cut -c1-300 | tee -a $O/progress.log
ACMI_BIG_DMA=0 timeout 600 python scripts/prefill_bench.py window melody > $O/prefill_reg.jsonl 2> $O/prefill_reg.err; cat $O/prefill_reg.jsonl | cut -c1-300 | tee -a $O/progress.log
log "kernel stats of the window prefill (DMA)"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python $R/scripts/prefill_bench.py window --reps 2 > /dev/null 2>&1)
cp $(find /tmp/kp -name "*kernel_stats.csv" | head -1) $O/prefill_kernel_stats.csv
python scripts/short_names.py $O/prefill_kernel_stats.csv 2>/dev/null | head -12 | tee -a $O/progress.log
log "parity configs with a prefill"
timeout 900 python -m pytest tests/test_gpu_parity_configs.py -m gpu -x -q -k "late_context or melody" 2>&1 | tail -4 | tee -a $O/progress.log
log "done"
