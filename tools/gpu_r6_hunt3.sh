#!/bin/bash
# round 6, hunt 3: geo_record as a standalone probe beside real rollouts; the whole encoder on the scalar-form library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for v in g0 g1; do
  echo "== geo probe, library $v, impl-1 rollouts as load"; PS_LIB=$PWD/prosim_amd/libprosim_hip_$v.so timeout 300 python tools/gpu_geo_probe.py 60 200
done
echo "== geo probe, library g0, no load"; PS_NOLOAD=1 PS_LIB=$PWD/prosim_amd/libprosim_hip_g0.so timeout 300 python tools/gpu_geo_probe.py 60 200
echo "== geo probe, library g0, 512 workgroups (two waves per SIMD of its own), no load"; PS_PROBE_WG=1024 PS_NOLOAD=1 PS_LIB=$PWD/prosim_amd/libprosim_hip_g0.so timeout 300 python tools/gpu_geo_probe.py 30 200
echo "== whole encoder + generator + first policy step, scalar-form library g1, impl-1 load"
PS_LIB=$PWD/prosim_amd/libprosim_hip_g1.so PS_LOAD_IMPL=1 PS_STAGES=encode,generate,policy timeout 900 python tools/gpu_stage_stress.py 300
echo "== the same, library g0"
PS_LIB=$PWD/prosim_amd/libprosim_hip_g0.so PS_LOAD_IMPL=1 PS_STAGES=encode,generate,policy timeout 900 python tools/gpu_stage_stress.py 300
} > gpurun_out/r6_hunt3.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r6_hunt3.log | tail -60
