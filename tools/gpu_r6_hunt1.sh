#!/bin/bash
# round 6, hunt 1: which launch of the encoder loses its bits under load (experiments library)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PS_LIB=$PWD/prosim_amd/libprosim_hip_exp.so
{
echo "== stage stress, default, exp lib"; timeout 300 python tools/gpu_stage_stress.py 300
echo "== bisect under rollout load, impl 0"; timeout 600 python tools/gpu_stage_bisect.py 150 1,2,3,4,5,20
echo "== bisect under rollout load (impl-1 load), impl 0"; PS_LOAD_IMPL=1 timeout 600 python tools/gpu_stage_bisect.py 150 2,3,4,5
echo "== bisect impl 2 under rollout load"; PS_IMPL=2 timeout 600 python tools/gpu_stage_bisect.py 150 3,4,5
echo "== poison, impl 0"; PS_POISON=1 timeout 600 python tools/gpu_stage_bisect.py 60 1,2,3,4,5,20
echo "== poison, impl 1"; PS_POISON=1 PS_IMPL=1 timeout 600 python tools/gpu_stage_bisect.py 60 4,5,20
echo "== poison, impl 2"; PS_POISON=1 PS_IMPL=2 timeout 600 python tools/gpu_stage_bisect.py 60 4,5,20
} > gpurun_out/r6_hunt1.log 2>&1
tail -60 gpurun_out/r6_hunt1.log
