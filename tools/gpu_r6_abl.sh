#!/bin/bash
# round 6: ablation timing of the policy launch (k_chain16, 16 rows per workgroup, alone on the GPU): what disappears with each part of a tile
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for f in "$@"; do
  echo "== $f"; PS_LIB=$PWD/prosim_amd/libprosim_abl_$f.so PS_ROWS=16 timeout 200 python tools/gpu_c16_prof.py 2>&1 | grep "policy launch"
done
} > gpurun_out/r6_abl.log 2>&1
cat gpurun_out/r6_abl.log
