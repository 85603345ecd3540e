#!/bin/bash
set -u
O=$PWD/gpurun_out/r6s17; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd /tmp
for dur in 10 30; do
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_$dur -- python $GRAFT_REPO_ROOT/scripts/fused_chain.py facebook/musicgen-medium 8 $dur 4 > $O/pmc_$dur.log 2>&1; echo "dur $dur rc=$?"
grep -v "^W2026\|^E2026" $O/pmc_$dur.log | tail -6
find $O/pmc_$dur -name "*counter_collection.csv" -exec python $GRAFT_REPO_ROOT/scripts/summarize_pmc.py {} \; | grep -i "qkv_attn" | head -3; rm -rf $O/pmc_$dur
done
