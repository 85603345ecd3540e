#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mbd3; mkdir -p $O
cd $R
timeout 200 python scripts/mbd_bench.py --seconds 1 --reps 5 > $O/mbd_1s.json 2>/dev/null; cat $O/mbd_1s.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/scripts/mbd_bench.py --seconds 1 --reps 3 > /dev/null 2>&1)
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/mbd_1s_kernel_stats.csv
python scripts/short_names.py $O/mbd_1s_kernel_stats.csv | head -12
