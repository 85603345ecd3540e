#!/bin/bash
set -u
O=$PWD/gpurun_out/r6s14; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd /tmp
timeout 200 python $GRAFT_REPO_ROOT/scripts/fused_chain.py facebook/musicgen-medium 8 3 4 2>&1 | tail -3
for c in FETCH_SIZE WRITE_SIZE; do
timeout 280 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $GRAFT_REPO_ROOT/scripts/fused_chain.py facebook/musicgen-medium 8 3 4 > $O/pmc_$c.log 2>&1; echo "rc=$?"
tail -4 $O/pmc_$c.log
find $O/pmc_$c -name "*counter_collection.csv" -exec python $GRAFT_REPO_ROOT/scripts/summarize_pmc.py {} \; | grep -i "qkv_attn\|lin_tiled\|counter" | head; rm -rf $O/pmc_$c
done
