#!/bin/bash
cd "$(dirname "$0")"
for pat in 0 1 2; do
 for mb in 4 512; do
  for wgs in 1 256 512; do
   for nl in 8 16 32; do
     ./mb_gather $pat $mb $wgs $nl 64 1
   done
  done
 done
done
# transposed LDS read recipe (next round's single rel-PE image)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 mb_trread.hip -o mb_trread 2>/dev/null && ./mb_trread
