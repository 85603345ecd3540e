#!/bin/bash
# round 4, session 6: MALL probe (GEMM launches with pre-touched weights), released-shape MBD test, small-model timeline
set -u
O=$PWD/gpurun_out/s6
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
for t in 0 1; do
log "MALL probe touch=$t"
(cd /tmp && ACMI_PROBE_TOUCH=$t rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp$t -- python $R/scripts/mall_probe.py > $O/mall_probe_$t.log 2>&1)
tail -1 $O/mall_probe_$t.log | tee -a $O/progress.log
cp $(find /tmp/mp$t -name "*kernel_stats.csv" | head -1) $O/mall_probe_kernel_stats_$t.csv
python scripts/short_names.py $O/mall_probe_kernel_stats_$t.csv 2>/dev/null | grep "lin_\|reduce" | head -8 | tee -a $O/progress.log
done
log "MBD released shape test"
timeout 600 python -m pytest tests/test_gpu_mbd.py -m gpu -x -q -k "released_shape or tokens_to_wav or roundtrip" 2>&1 | tail -4 | tee -a $O/progress.log
log "timeline small model, B = 1"
ACMI_LIB=$R/audiocraft_amd/csrc/libacmi_trace.so timeout 300 python scripts/lin_timeline.py --model facebook/musicgen-small --batch 1 --frames 300 --out $O/lin_timeline_small_b1.csv > $O/lin_timeline_small_b1.log 2>&1
tail -7 $O/lin_timeline_small_b1.log | tee -a $O/progress.log
log "done"
