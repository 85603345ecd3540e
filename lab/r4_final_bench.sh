#!/bin/bash
# round 4, final build: the bench line with the driver's flags (live PMC traffic), rocprofv3 kernel stats of one generate
set -u
O=$PWD/gpurun_out/final
R=$PWD
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
log() { echo "== $*" | tee -a $O/progress.log; }
: > $O/progress.log
log "bench, driver flags"
S0=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1_driver_flags.json 2> $O/bench_n1_driver_flags.err
cut -c1-1500 $O/bench_n1_driver_flags.json | tee -a $O/progress.log
echo "wall $(( $(date +%s) - S0 )) s" | tee -a $O/progress.log; tail -5 $O/bench_n1_driver_flags.err | cut -c1-300 | tee -a $O/progress.log
log "done"
