#!/bin/bash
# Kernel-trace profile of the batch rollout + chain ablations; text summaries land in gpurun_out/.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
B=${PS_BATCH:-8}
for f in 0 8 16 24 1; do PS_BATCH=$B PS_CHAIN_FLAGS=$f python tools/gpu_ablate.py 2>&1 | grep flags; done | tee gpurun_out/ablate_b$B.txt
rm -rf /tmp/prof_x && PS_BATCH=$B rocprofv3 --kernel-trace -d /tmp/prof_x -o x -- python tools/gpu_ablate.py > /tmp/prof_x.log 2>&1
python tools/prof_summary.py $(find /tmp/prof_x -name '*.db' | head -1) > gpurun_out/trace_b$B.txt 2>&1
head -24 gpurun_out/trace_b$B.txt
