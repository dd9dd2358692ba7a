#!/bin/bash
# Hardware counters of the policy launch (k_attn_chain<.., true>), a few per pass (PMC passes only carry
# --kernel-trace).  Summary -> gpurun_out/<tag>_pmc_chain.txt
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r01_x}
OUT=gpurun_out/${TAG}_pmc_chain.txt
mkdir -p gpurun_out; : > $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL TA_BUSY_avr"; do
  rm -rf /tmp/prof_p && PS_BATCH=8 rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof_p -o p -- python tools/gpu_ablate.py > /tmp/prof_p.log 2>&1
  python - "$(find /tmp/prof_p -name '*.db' | head -1)" >> $OUT <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_attn_chain%true>%' "
                  "and grid_size = 131072 group by counter_name").fetchall()
for n, c, a in rows: print(f"{n:34s} launches {c:4d}  avg per launch {a:16.1f}")
PY
done
cat $OUT
