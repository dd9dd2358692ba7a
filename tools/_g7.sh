mkdir -p gpurun_out; rm -f gpurun_out/r05_parity.json
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r05_b_pytest.txt; cat gpurun_out/r05_b_pytest.txt
