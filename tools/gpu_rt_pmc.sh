#!/bin/bash
# Hardware counters of one row-tile kernel configuration (separate --pmc passes).  usage: tools/gpu_rt_pmc.sh <tag> <kernel-like> [env...]
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=$1; KL=$2; shift 2
OUT=gpurun_out/${TAG}_pmc.txt
mkdir -p gpurun_out; echo "# $KL $*; averages per launch" > $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_IFETCH" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/prof_p && env "$@" rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof_p -o p -- python ${PS_PMC_SCRIPT:-tools/gpu_rt_prof.py} > /tmp/prof_p.log 2>&1
  python - "$(find /tmp/prof_p -name '*.db' | head -1)" "$KL" >> $OUT <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cc = [t for t in tabs if t.startswith("counters_collection")]
rows = db.execute(f"select counter_name, count(*), avg(value) from {cc[0]} where kernel_name like '%{sys.argv[2]}%' group by counter_name").fetchall()
for n, c, a in rows: print(f"{n:34s} launches {c:4d}  avg per launch {a:18.1f}")
PY
done
cat $OUT
