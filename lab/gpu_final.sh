#!/bin/bash
# round 3: final-build artefacts (run on the GPU box from the repo root); outputs under gpurun_out/final
set -u
O=$PWD/gpurun_out/final
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "bench default (driver's flags scaled down)"
python bench.py --steps 5 --warmup 2 > $O/bench_n1_default.json 2> $O/bench_n1_default.err
cat $O/bench_n1_default.json | tee -a $O/progress.log
log "bench small B=1 (configs[1])"
python bench.py --model facebook/musicgen-small --batch 1 --duration 10 --greedy --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_small_b1.json 2> /dev/null
python -c "import json; d=json.load(open('$O/bench_small_b1.json')); print('small B=1 RTF', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])" | tee -a $O/progress.log
log "rocprof kernel stats (one generate)"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python scripts/short_names.py $O/bench_kernel_stats.csv | head -16 | tee -a $O/progress.log
log "PMC passes over the GEMM chain (FETCH_SIZE, WRITE_SIZE, MFMA)"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/scripts/dbg_chain.py > /dev/null 2>&1)
  python scripts/summarize_pmc.py $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) > $O/lin_chain_pmc_$c.csv
done
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/pmc_mfma -- python $R/scripts/dbg_chain.py > /dev/null 2>&1)
python scripts/summarize_pmc.py $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) > $O/lin_chain_pmc_mfma.csv
head -12 $O/lin_chain_pmc_FETCH_SIZE.csv | tee -a $O/progress.log
log "PMC passes over the attention kernel"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d /tmp/pmca_$c -- python $R/scripts/dbg_attn.py > /dev/null 2>&1)
  python scripts/summarize_pmc.py $(find /tmp/pmca_$c -name "*counter_collection.csv" | head -1) > $O/attn_pmc_$c.csv
done
log "prefill: cost, kernel stats, MFMA-busy PMC"
python scripts/prefill_bench.py window melody --reps 3 > $O/prefill_cost.jsonl 2> /dev/null
cat $O/prefill_cost.jsonl | tee -a $O/progress.log
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/scripts/prefill_bench.py window --reps 2 > /dev/null 2>&1)
cp $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) $O/prefill_kernel_stats.csv
python scripts/short_names.py $O/prefill_kernel_stats.csv | head -10 | tee -a $O/progress.log
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_pf -- python $R/scripts/prefill_bench.py window --reps 1 > /dev/null 2>&1)
python scripts/summarize_pmc.py $(find /tmp/pmc_pf -name "*counter_collection.csv" | head -1) | grep -E "kernel,|lin_big|attn_prefill|ln_tile" > $O/prefill_pmc_mfma.csv
cat $O/prefill_pmc_mfma.csv | tee -a $O/progress.log
log "attention microbench"
python scripts/attn_bench.py > $O/attn_microbench.log 2>&1; tail -1 $O/attn_microbench.log | tee -a $O/progress.log
log "codec bench + per-layer breakdown"
python scripts/codec_bench.py > $O/codec_bench.jsonl 2> $O/codec_bench.err
ACMI_CONV_KSPLIT=1 python scripts/codec_bench.py > $O/codec_bench_KSPLIT1.jsonl 2> /dev/null
python scripts/codec_layers.py decode > $O/codec_layers_decode.log 2>&1
python scripts/codec_layers.py encode > $O/codec_layers_encode.log 2>&1
tail -1 $O/codec_layers_decode.log | tee -a $O/progress.log
log "conv kernels: MFMA-busy / traffic PMC over one EnCodec-32k decode (8 x 30 s)"
cat > /tmp/dec.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from audiocraft_amd.models import builders
m = builders.get_compression_model(builders.ENCODEC_32KHZ, 'cuda')
codes = torch.randint(0, 2048, (8, 4, 1500), device='cuda')
m.decode(codes)
m.decode(codes)
torch.cuda.synchronize()
PY
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/pmc_conv -- python /tmp/dec.py $R > /dev/null 2>&1)
python scripts/summarize_pmc.py $(find /tmp/pmc_conv -name "*counter_collection.csv" | head -1) | grep -E "kernel,|conv_|lstm" > $O/conv_pmc_mfma.csv
cat $O/conv_pmc_mfma.csv | tee -a $O/progress.log
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d /tmp/pmcc_$c -- python /tmp/dec.py $R > /dev/null 2>&1)
  python scripts/summarize_pmc.py $(find /tmp/pmcc_$c -name "*counter_collection.csv" | head -1) | grep -E "kernel,|conv_|lstm" > $O/conv_pmc_$c.csv
done
log "MultiBandDiffusion: cost + kernel stats"
python scripts/mbd_bench.py --seconds 10 > $O/mbd_bench_10s.json 2> /dev/null; cat $O/mbd_bench_10s.json | tee -a $O/progress.log
python scripts/mbd_bench.py --seconds 30 --batch 2 > $O/mbd_bench_30s_b2.json 2> /dev/null; cat $O/mbd_bench_30s_b2.json | tee -a $O/progress.log
for sec in 1 3; do python scripts/mbd_bench.py --seconds $sec > $O/mbd_bench_${sec}s.json 2> /dev/null; ACMI_CONV_KSPLIT=1 python scripts/mbd_bench.py --seconds $sec > $O/mbd_bench_${sec}s_KSPLIT1.json 2> /dev/null; done
cat $O/mbd_bench_1s.json $O/mbd_bench_1s_KSPLIT1.json | tee -a $O/progress.log
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python $R/scripts/mbd_bench.py --seconds 10 --reps 3 > /dev/null 2>&1)
cp $(find /tmp/kt3 -name "*kernel_stats.csv" | head -1) $O/mbd_kernel_stats.csv
python scripts/short_names.py $O/mbd_kernel_stats.csv | head -10 | tee -a $O/progress.log
log "smoke"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $O/progress.log
log "windowed 60 s generation: one-forward prefill vs chunk path"
python scripts/window_bench.py > $O/window_60s.json 2> /dev/null; cat $O/window_60s.json | tee -a $O/progress.log
ACMI_PREFILL=chunk python scripts/window_bench.py > $O/window_60s_chunk.json 2> /dev/null; cat $O/window_60s_chunk.json | tee -a $O/progress.log
log "config sweep"
python scripts/config_sweep.py > $O/config_sweep.log 2>&1
cat $O/config_sweep.log | grep -v amdgpu | tee -a $O/progress.log
log "done"
