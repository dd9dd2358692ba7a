mkdir -p gpurun_out; rm -f gpurun_out/r05_parity.json
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r05_c_pytest.txt; tail -5 gpurun_out/r05_c_pytest.txt
timeout 900 python bench.py > gpurun_out/r05_c_bench.json 2> gpurun_out/r05_c_bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_c_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','graph_nodes_per_rollout')}); print(d['streaming']['agent_steps_per_s_by_depth']); print(d['single_scene']); print(d['latency_mode'])
PY
