#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/full1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest.log
cat > /tmp/dec.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from audiocraft_amd.models import builders
m = builders.get_compression_model(builders.ENCODEC_32KHZ, 'cuda')
wav = 0.1 * torch.randn(8, 1, 30 * 32000, device='cuda')
codes, _ = m.encode(wav)
for _ in range(3):
    m.decode(codes)
torch.cuda.synchronize()
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python /tmp/dec.py $R > $O/prof.log 2>&1)
find /tmp/kt -name '*kernel_stats.csv' -exec cp {} $O/codec_kernel_stats.csv \;
cat $O/pytest.log; cut -d, -f1-4 $O/codec_kernel_stats.csv | cut -c1-150 | head -14
