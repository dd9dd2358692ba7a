#!/bin/bash
# Ablations of the k_chain16 policy launch (timing only; results are wrong by construction):
# PS_C16_ABL bits: 1 every layer reads layer 0's k|v (L2-resident), 2 every k / v row of a tile from ONE source (L1 hits),
# 4 no Fourier rows.  usage: tools/gpu_c16_abl.sh <tag> [rows]
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r03_x}; ROWS=${2:-16}
OUT=gpurun_out/${TAG}_c16_abl.txt; mkdir -p gpurun_out; : > $OUT
for abl in 0 1 2 3 4 6 7; do
  echo "== PS_C16_ABL=$abl rows=$ROWS" >> $OUT
  PS_C16_ABL=$abl PS_ROWS=$ROWS python tools/gpu_c16_prof.py 2>&1 | grep "policy launch" >> $OUT
done
echo "== phase clocks, ABL=0 / 2 / 6" >> $OUT
for abl in 0 2 6; do PS_CHAIN_PROF=1 PS_C16_ABL=$abl PS_ROWS=$ROWS python tools/gpu_c16_prof.py 2>&1 | tail -2 >> $OUT; done
cat $OUT
