mkdir -p gpurun_out; rm -f gpurun_out/r05_parity.json
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r05_d_pytest.txt; tail -5 gpurun_out/r05_d_pytest.txt
PS_LIB=prosim_amd/libprosim_hip_exp.so timeout 900 python -m pytest tests/test_round4_gpu.py -m gpu -q -k "row_impls_agree" 2>&1 | tail -3
