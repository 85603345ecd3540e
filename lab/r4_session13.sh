#!/bin/bash
# round 4, session 13: GroupNorm folded into the convolution's input pack (MBD), ln_tile 4 rows per workgroup
set -u
O=$PWD/gpurun_out/s13
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "MBD tests + ln_tile + prefill tests"
timeout 900 python -m pytest tests/test_gpu_mbd.py tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "mbd or unet or diffusion or multiband or ln_tile or big or prefill or golden or conv" 2>&1 | tail -6 | tee -a $O/progress.log
log "MBD bench 10 s, 1 s, 2 x 30 s"
timeout 300 python scripts/mbd_bench.py --seconds 10 > $O/mbd_bench_10s.json 2> /dev/null; cut -c1-600 $O/mbd_bench_10s.json | tee -a $O/progress.log
timeout 300 python scripts/mbd_bench.py --seconds 1 > $O/mbd_bench_1s.json 2> /dev/null; cut -c1-300 $O/mbd_bench_1s.json | tee -a $O/progress.log
timeout 300 python scripts/mbd_bench.py --seconds 30 --batch 2 > $O/mbd_bench_30s_b2.json 2> /dev/null; cut -c1-300 $O/mbd_bench_30s_b2.json | tee -a $O/progress.log
log "MBD kernel stats (10 s forward)"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/km -- python $R/scripts/mbd_bench.py --seconds 10 --reps 3 > /dev/null 2>&1)
cp $(find /tmp/km -name "*kernel_stats.csv" | head -1) $O/mbd_kernel_stats.csv
python scripts/short_names.py $O/mbd_kernel_stats.csv 2>/dev/null | head -8 | tee -a $O/progress.log
log "prefill bench"
timeout 600 python scripts/prefill_bench.py window melody > $O/prefill.jsonl 2> $O/prefill.err; cat $O/prefill.jsonl | cut -c1-300 | tee -a $O/progress.log
log "done"
