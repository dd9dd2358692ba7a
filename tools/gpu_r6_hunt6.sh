#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{ timeout 900 tools/mb/mb_pksgpr2 20000; } > gpurun_out/r6_hunt6.log 2>&1
cat gpurun_out/r6_hunt6.log | tail -120
