#!/bin/bash
# round 6 session 3: the device lock (two host threads + a codec thread, 20 processes), the thread-local capture repro, model suites
set -u
O=$PWD/gpurun_out/r6s3; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ulimit -c 0
for i in $(seq 1 20); do
  timeout 300 python -m pytest tests/test_gpu_models.py -q -x -m gpu -k two_host_threads 2>&1 | tail -1
done | tee $O/two_thread_lock_x20.txt
for m in torch acmi; do
  for i in $(seq 1 12); do
    timeout 120 python -X faulthandler lab/two_thread_capture_repro.py $m 20 > $O/repro_${m}_$i.log 2>&1; echo "$m run $i rc=$?"
  done
done | tee $O/two_thread_capture_repro.txt
grep -l "Fatal\|Segmentation\|Aborted\|core" $O/repro_*.log | head -3 | while read f; do echo "== $f"; head -60 $f; done | tee $O/repro_first_crashes.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_distributed.py -q -x -m gpu 2>&1 | tail -4 | tee $O/models_distributed_pytest.txt
