mkdir -p gpurun_out; rm -f gpurun_out/r05_parity.json
python tools/gpu_traj_digest.py 2>&1 | grep -v amdgpu | tee gpurun_out/r05_e_digest.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r05_e_pytest.txt; tail -4 gpurun_out/r05_e_pytest.txt
timeout 900 python bench.py > gpurun_out/r05_e_bench.json 2> gpurun_out/r05_e_bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_e_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','graph_nodes_per_rollout')}); print(d['streaming']['agent_steps_per_s_by_depth']); print(d['single_scene']); print(d['latency_mode'])
PY
