#!/bin/bash
# round 6: where do the waves of the policy launch sit?  rocprofv3 PC sampling (beta) over tools/gpu_c16_prof.py (PS_ROWS=16: the throughput-mode launch, alone)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out/pcs
METHOD=${1:-stochastic}; UNIT=${2:-cycles}; INT=${3:-1048576}
PS_ROWS=16 timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INT --kernel-trace \
   --output-format csv json -d gpurun_out/pcs -- python tools/gpu_c16_prof.py > gpurun_out/pcs/run.log 2>&1
echo "rc $?"; tail -5 gpurun_out/pcs/run.log; find gpurun_out/pcs -type f | head -20; du -sh gpurun_out/pcs
for f in $(find gpurun_out/pcs -name "*pc_sampling*csv" | head -2); do echo "== $f"; head -5 $f; wc -l $f; done
