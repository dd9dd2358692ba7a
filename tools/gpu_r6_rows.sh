#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PS_LIB=$PWD/prosim_amd/libprosim_hip_exp.so
{
for v in "0 0" "8 0" "4 0" "0 8" "8 8" "0 0" "8 0"; do
  set -- $v
  echo "== PS_C16_ROWS_SMALL=$1 PS_C16_ROWS_S2S=$2"
  for i in 1 2; do PS_C16_ROWS_SMALL=$1 PS_C16_ROWS_S2S=$2 PS_STEPS=96 timeout 200 python tools/gpu_headline_loop.py 2>&1 | grep "agent-steps"; done
done
} > gpurun_out/r6_rows.log 2>&1
cat gpurun_out/r6_rows.log
