set -x
mkdir -p gpurun_out/final2
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/final2/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final2/smoke.txt 2>&1
python bench.py > gpurun_out/final2/bench_n1_default.json 2> gpurun_out/final2/bench.err
