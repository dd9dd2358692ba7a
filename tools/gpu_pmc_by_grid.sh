#!/bin/bash
# Hardware counters of every launch shape of one kernel over a throughput-mode rollout of the bench batch (separate --pmc passes, a
# few counters each; PMC passes only carry --kernel-trace), averaged per (kernel, grid size): the generator / a2a / s2s launches of
# k_chain16<8, false, true> differ only by their grid, and so do the condition layers' k_attn_chain launches.
# usage: tools/gpu_pmc_by_grid.sh <tag> <kernel-like> [env...]      summary -> gpurun_out/<tag>_pmc.txt
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=$1; KL=$2; shift 2
OUT=gpurun_out/${TAG}_pmc.txt
mkdir -p gpurun_out; echo "# kernels like '$KL' during one throughput-mode rollout of 8 x cfg2 scenes ($*); per launch shape: launches, average per launch" > $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/prof_p && env PS_ROWS=16 "$@" rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof_p -o p -- python tools/gpu_c16_prof.py > /tmp/prof_p.log 2>&1
  python - "$(find /tmp/prof_p -name '*.db' | head -1)" "$KL" >> $OUT <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cc = [t for t in tabs if t.startswith("counters_collection")]
rows = db.execute(f"select kernel_name, grid_size, counter_name, count(*), avg(value) from {cc[0]} where kernel_name like '%{sys.argv[2]}%' group by kernel_name, grid_size, counter_name order by kernel_name, grid_size").fetchall()
for k, g, n, c, a in rows: print(f"{re.sub(r'[(].*', '', k)[:48]:48s} grid {g:8d} {n:30s} launches {c:4d}  avg {a:16.1f}")
PY
done
# durations of the same launch shapes (kernel trace of the same command)
rm -rf /tmp/prof_t && env PS_ROWS=16 "$@" rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python tools/gpu_c16_prof.py > /tmp/prof_t.log 2>&1
python tools/prof_by_grid.py "$(find /tmp/prof_t -name '*.db' | head -1)" "$KL" >> $OUT 2>&1
tail -40 $OUT
