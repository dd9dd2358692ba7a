#!/bin/bash
# Round-end evidence: the default bench line, a rocprofv3 kernel trace of the same command (short), and two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE).  Text summaries land in gpurun_out/ (the DBs stay in /tmp).
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r01_x}
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 600 gpurun_out/${TAG}_bench.json; echo
# kernel averages and counters from UNPIPELINED steps (one rollout on the GPU at a time): with the default --inflight 3 the
# launches of up to three rollouts share the CUs and every duration is stretched by the concurrency
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --inflight 1 --chain-rows 4"
rm -rf /tmp/prof_t && rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- $CMD > /tmp/prof_t.log 2>&1
python tools/prof_summary.py $(find /tmp/prof_t -name '*.db' | head -1) > gpurun_out/${TAG}_kernel_trace.txt 2>&1
head -14 gpurun_out/${TAG}_kernel_trace.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_p && rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_p -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 --chain-rows 4 > /tmp/prof_p.log 2>&1
  python tools/pmc_summary.py $(find /tmp/prof_p -name '*.db' | head -1) $c > gpurun_out/${TAG}_pmc_$(echo $c | tr A-Z a-z).txt 2>&1
  head -6 gpurun_out/${TAG}_pmc_$(echo $c | tr A-Z a-z).txt
done
