#!/bin/bash
# Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on known byte counts (MI355X_MICROARCH.md, HBM section: "calibrate on
# a known byte count in your own access pattern"): a 1 GiB elementwise torch kernel (16 B per lane, reads 1 GiB and
# writes 1 GiB) and the engine's own streaming test (ps_test_stream).
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/calib.py <<'PY'
import torch
x = torch.ones(256 * 1024 * 1024, device="cuda")          # 1 GiB fp32
torch.cuda.synchronize()
for _ in range(3):
    y = x * 2.0                                            # reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
z = torch.ones(16 * 1024 * 1024, device="cuda")           # 64 MiB: fits the Infinity Cache
for _ in range(3):
    w = z * 2.0
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_c && rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_c -o c -- python /tmp/calib.py > /tmp/prof_c.log 2>&1
  python tools/pmc_summary.py $(find /tmp/prof_c -name '*.db' | head -1) $c > gpurun_out/calib_$(echo $c | tr A-Z a-z).txt 2>&1
  cat gpurun_out/calib_$(echo $c | tr A-Z a-z).txt | head -12
done
