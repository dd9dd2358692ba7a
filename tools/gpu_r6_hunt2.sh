#!/bin/bash
# round 6, hunt 2: k_edge_geo's statistics under load -- source variants of geo_record, and the trans-op hazard microbenchmark
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== microbenchmark"; timeout 300 tools/mb/mb_transhaz 20000
for v in exp g1 g2 g4; do
  echo "== stage 2 under impl-1 rollout load, library $v"
  PS_LIB=$PWD/prosim_amd/libprosim_hip_$v.so PS_LOAD_IMPL=1 timeout 600 python tools/gpu_stage_bisect.py 200 2
done
} > gpurun_out/r6_hunt2.log 2>&1
tail -80 gpurun_out/r6_hunt2.log
