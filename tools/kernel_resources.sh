#!/bin/bash
# Register / scratch / occupancy of every kernel in the library as hipcc sees them (no GPU needed):
#   tools/kernel_resources.sh [filter]      e.g. tools/kernel_resources.sh pointnet_rt
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -Iprosim_amd/csrc \
  -Rpass-analysis=kernel-resource-usage prosim_amd/csrc/prosim_hip.hip -o /tmp/ps_res.so 2>&1 | grep "remark:" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | \
  awk '/^Function Name:/ {name=$3} /^VGPRs:/ {v=$2} /^AGPRs:/ {a=$2} /^ScratchSize/ {s=$NF} /^Occupancy/ {o=$NF} /^LDS Size/ {print name, "vgpr", v, "agpr", a, "scratch", s, "occ", o}' | \
  c++filt | sed 's/(ps::PointNetW.*//; s/(float\*.*//; s/(int, .*//' | grep -i "${1:-.}"
