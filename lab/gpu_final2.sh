#!/bin/bash
# round 3, last session: the artefacts that changed after lab/gpu_final.sh ran (conv split-K, LSTM): bench line + kernel stats,
# codec + MBD cost with same-box A/B, smoke
set -u
O=$PWD/gpurun_out/final2
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "bench default"
python bench.py --steps 4 --warmup 2 > $O/bench_n1_default.json 2> $O/bench_n1_default.err
cut -c1-300 $O/bench_n1_default.json | tee -a $O/progress.log
log "rocprof kernel stats (one generate)"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python scripts/short_names.py $O/bench_kernel_stats.csv | head -12 | tee -a $O/progress.log
log "codec bench (default / ACMI_CONV_KSPLIT=1) + per-layer breakdown"
python scripts/codec_bench.py > $O/codec_bench.jsonl 2> /dev/null
ACMI_CONV_KSPLIT=1 python scripts/codec_bench.py > $O/codec_bench_KSPLIT1.jsonl 2> /dev/null
python - <<PY | tee -a $O/progress.log
import json
for f in ('codec_bench.jsonl', 'codec_bench_KSPLIT1.jsonl'):
    for l in open('$O/' + f):
        d = json.loads(l); print(f, d['config'][:36], 'enc', d['encode']['ms'], 'dec', d['decode']['ms'])
PY
python scripts/codec_layers.py decode > $O/codec_layers_decode.log 2>&1
python scripts/codec_layers.py encode > $O/codec_layers_encode.log 2>&1
tail -1 $O/codec_layers_decode.log | tee -a $O/progress.log
log "MultiBandDiffusion"
python scripts/mbd_bench.py --seconds 10 > $O/mbd_bench_10s.json 2> /dev/null; cut -c1-330 $O/mbd_bench_10s.json | tee -a $O/progress.log
for sec in 1 3; do python scripts/mbd_bench.py --seconds $sec > $O/mbd_bench_${sec}s.json 2> /dev/null; ACMI_CONV_KSPLIT=1 python scripts/mbd_bench.py --seconds $sec > $O/mbd_bench_${sec}s_KSPLIT1.json 2> /dev/null; done
cut -c1-220 $O/mbd_bench_1s.json $O/mbd_bench_1s_KSPLIT1.json $O/mbd_bench_3s.json $O/mbd_bench_3s_KSPLIT1.json | tee -a $O/progress.log
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python $R/scripts/mbd_bench.py --seconds 1 --reps 3 > /dev/null 2>&1)
cp $(find /tmp/kt3 -name "*kernel_stats.csv" | head -1) $O/mbd_1s_kernel_stats.csv
python scripts/short_names.py $O/mbd_1s_kernel_stats.csv | head -8 | tee -a $O/progress.log
log "smoke"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/progress.log
log "remaining GPU tests (the files not run since the split-K change)"
timeout 500 python -m pytest tests/test_gpu_parity_configs.py tests/test_gpu_musicgen_api.py tests/test_gpu_chroma.py tests/test_gpu_distributed.py -q -x 2>&1 | tail -4 | tee -a $O/progress.log
log "done"
