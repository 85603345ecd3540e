#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof24; mkdir -p $O
cat > /tmp/p24.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from audiocraft_amd.models import builders
torch.manual_seed(0)
m = builders.get_compression_model(builders.ENCODEC_24KHZ, 'cuda')
wav = 0.1 * torch.randn(1, 1, 240000, device='cuda')
for _ in range(6):
    codes, _ = m.encode(wav)
    m.decode(codes)
torch.cuda.synchronize()
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python /tmp/p24.py $R > $O/prof.log 2>&1)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/codec24_kernel_stats.csv
python $R/scripts/short_names.py $O/codec24_kernel_stats.csv | head -24
