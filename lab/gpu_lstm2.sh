#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lstm2; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mbd.py -q -x -k "lstm or bilstm" 2>&1 | tail -5 > $O/pytest.log
cat $O/pytest.log
for v in ""; do
  echo "== $v" | tee -a $O/cb.txt
  env $v timeout 200 python scripts/codec_bench.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'][:40], 'enc', d['encode']['ms'], 'dec', d['decode']['ms'])" | tee -a $O/cb.txt
done
