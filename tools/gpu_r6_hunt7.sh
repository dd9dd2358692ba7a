#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{ timeout 900 tools/mb/mb_pksgpr3 20000; } > gpurun_out/r6_hunt7.log 2>&1
cat gpurun_out/r6_hunt7.log | tail -120
