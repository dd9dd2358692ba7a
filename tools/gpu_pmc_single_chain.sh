#!/bin/bash
# Hardware counters of the SINGLE-SCENE policy launch (k_attn_chain<1, 4, 3, false, true, ...>: 128 one-row workgroups), a few per pass
# (PMC passes only carry --kernel-trace), over the rollouts of tools/gpu_single_timeline.py.  Summary -> gpurun_out/<tag>_pmc_single_chain.txt
# usage: tools/gpu_pmc_single_chain.sh <tag> [env...]     (e.g. PS_IMPL=1: the operand-image build, ps_set_chain_impl(1))
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r05_x}; shift
OUT=gpurun_out/${TAG}_pmc_single_chain.txt
mkdir -p gpurun_out; echo "# policy launch of one 128-agent scene ($*): launches, average per launch" > $OUT
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES"; do
  rm -rf /tmp/prof_p && env "$@" rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof_p -o p -- python tools/gpu_single_timeline.py > /tmp/prof_p.log 2>&1
  python - "$(find /tmp/prof_p -name '*.db' | head -1)" >> $OUT <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cc = [t for t in tabs if t.startswith("counters_collection")]
rows = db.execute(f"select kernel_name, counter_name, count(*), avg(value) from {cc[0]} where kernel_name like '%k_attn_chain<1, 4, 3, false, true%' group by kernel_name, counter_name").fetchall()
for k, n, c, a in rows: print(f"{re.sub(r'[(].*', '', k)[:52]:52s} {n:30s} launches {c:4d}  avg {a:16.1f}")
PY
done
cat $OUT
