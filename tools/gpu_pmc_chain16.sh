#!/bin/bash
# Hardware counters of the k_chain16 policy launch, a few per pass (PMC passes only carry --kernel-trace).
# usage: tools/gpu_pmc_chain16.sh <tag> [rows]      summary -> gpurun_out/<tag>_pmc_chain16.txt
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r02_x}; ROWS=${2:-12}
OUT=gpurun_out/${TAG}_pmc_chain16.txt
mkdir -p gpurun_out; echo "# k_chain16 policy launch, rows per workgroup = $ROWS, 8 x cfg2 scenes (1024 rows); averages per launch" > $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_IFETCH" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/prof_p && PS_ROWS=$ROWS rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof_p -o p -- python tools/gpu_c16_prof.py > /tmp/prof_p.log 2>&1
  python - "$(find /tmp/prof_p -name '*.db' | head -1)" >> $OUT <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cc = [t for t in tabs if t.startswith("counters_collection")]
rows = db.execute(f"select counter_name, count(*), avg(value) from {cc[0]} where kernel_name like '%k_chain16<8, true,%' group by counter_name").fetchall()
for n, c, a in rows: print(f"{n:34s} launches {c:4d}  avg per launch {a:18.1f}")
PY
done
cat $OUT
