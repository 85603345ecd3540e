#!/bin/bash
# round 4, session 12: prefill GEMM: column groups per XCD for every N
set -u
O=$PWD/gpurun_out/s12
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "prefill tests (ring default)"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "big or prefill or golden or midsize or streaming" 2>&1 | tail -5 | tee -a $O/progress.log
for m in 2 1 0; do
log "prefill bench ACMI_BIG_DMA=$m"
ACMI_BIG_DMA=$m timeout 600 python scripts/prefill_bench.py window melody > $O/prefill_dma$m.jsonl 2> $O/prefill_dma$m.err; cat $O/prefill_dma$m.jsonl | cut -c1-300 | tee -a $O/progress.log
done
log "kernel stats of the window prefill (ring)"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python $R/scripts/prefill_bench.py window --reps 2 > /dev/null 2>&1)
cp $(find /tmp/kp -name "*kernel_stats.csv" | head -1) $O/prefill_kernel_stats.csv
python scripts/short_names.py $O/prefill_kernel_stats.csv 2>/dev/null | head -9 | tee -a $O/progress.log
log "done"
