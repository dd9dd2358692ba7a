#!/bin/bash
# round 6: the legalised product library under every load of the hunt, then the whole -m gpu suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== stage stress, product library, impl-1 rollouts as load (the hardest load of the hunt)"
PS_LOAD_IMPL=1 PS_STAGES=encode,generate,policy timeout 900 python tools/gpu_stage_stress.py 600
echo "== stage stress, product library, default rollouts as load"
PS_STAGES=encode,generate,policy timeout 900 python tools/gpu_stage_stress.py 600
echo "== stage stress, probes on k_chain16 (chain_impl 2), impl-1 load"
PS_IMPL=2 PS_LOAD_IMPL=1 PS_STAGES=encode,generate,policy timeout 900 python tools/gpu_stage_stress.py 600
echo "== search stress: six engines of three kinds, 900 rollouts each"
timeout 900 python tools/gpu_search_stress.py 6 900
echo "== search stress, chain_impl 2"
PS_IMPL=2 timeout 900 python tools/gpu_search_stress.py 6 900
} > gpurun_out/r6_verify1.log 2>&1
grep -v amdgpu.ids gpurun_out/r6_verify1.log | tail -40
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_pytest1.log 2>&1; tail -15 gpurun_out/r6_pytest1.log
