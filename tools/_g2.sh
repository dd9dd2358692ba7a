mkdir -p gpurun_out
tools/mb/mb_feat > gpurun_out/r05_a_mb_feat.txt 2>&1; tail -3 gpurun_out/r05_a_mb_feat.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_a_pytest.txt; cat gpurun_out/r05_a_pytest.txt
timeout 600 python bench.py > gpurun_out/r05_a_bench.json 2> gpurun_out/r05_a_bench.err; cat gpurun_out/r05_a_bench.json
