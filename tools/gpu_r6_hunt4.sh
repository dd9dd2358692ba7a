#!/bin/bash
# round 6, hunt 4: which operand of the packed instructions -- scalar-pair divisors (5: in vector registers), scalar-pair constants (7), both (6)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for v in g0 g5 g6 g7; do
  echo "== geo probe, library $v, impl-1 rollouts as load"; PS_LIB=$PWD/prosim_amd/libprosim_hip_$v.so timeout 300 python tools/gpu_geo_probe.py 60 200
done
} > gpurun_out/r6_hunt4.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r6_hunt4.log | tail -60
