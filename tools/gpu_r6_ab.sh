#!/bin/bash
# A/B of experiment libraries: digests (tools/gpu_traj_digest.py prints the policy launch time beside them)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for f in "$@"; do
  echo "== $f"; PS_LIB=$PWD/prosim_amd/libprosim_abl_$f.so timeout 300 python tools/gpu_traj_digest.py 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r6_ab.log 2>&1
cat gpurun_out/r6_ab.log
