#!/bin/bash
# node-phase work of round 6: stage clocks of the policy launch (PROF build) + digests and launch time of the plain build of the same source
# usage: tools/gpu_r6_node.sh <tag>   (libraries prosim_amd/libprosim_abl_PROF.so and libprosim_abl_<tag>.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== stage clocks"; PS_LIB=$PWD/prosim_amd/libprosim_abl_PROF.so PS_CHAIN_PROF=1 PS_ROWS=16 timeout 300 python tools/gpu_c16_prof.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== digests $1"; PS_LIB=$PWD/prosim_amd/libprosim_abl_$1.so timeout 300 python tools/gpu_traj_digest.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r6_node_$1.log 2>&1
cat gpurun_out/r6_node_$1.log
