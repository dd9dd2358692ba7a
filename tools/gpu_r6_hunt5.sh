#!/bin/bash
# round 6, hunt 5: packed-fp32 instructions with scalar-pair sources beside eight kinds of load
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{ timeout 900 tools/mb/mb_pksgpr 20000; } > gpurun_out/r6_hunt5.log 2>&1
cat gpurun_out/r6_hunt5.log | tail -120
