#!/bin/bash
# round 5 session 26: HBM traffic of the self-attention launch at the bench's mean context (PMC, separate passes)
set -u
O=$PWD/gpurun_out/r5s26; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/scripts/attn_bench.py --contexts 751 > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  echo "== $c"; python $R/scripts/summarize_pmc.py $f | grep -i "attn\|kernel,counter"
done 2>&1 | tee $O/attn_pmc_t751.txt
