#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_round6_gpu.py tests/test_stream_gpu.py tests/test_round5_gpu.py -x -q -s > gpurun_out/r6_pytest2.log 2>&1; tail -25 gpurun_out/r6_pytest2.log
timeout 600 python bench.py > gpurun_out/r6_bench1.json 2> gpurun_out/r6_bench1.err; tail -3 gpurun_out/r6_bench1.err; cat gpurun_out/r6_bench1.json
