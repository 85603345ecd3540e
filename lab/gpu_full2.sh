#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/full2; mkdir -p $O
cd $R
timeout 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest.log
cat $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
