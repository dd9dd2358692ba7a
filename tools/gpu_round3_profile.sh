#!/bin/bash
# Round-3 evidence: the default bench line, a rocprofv3 kernel trace of the same (pipelined) command with its CU-time table,
# the hardware counters of the k_chain16 policy launch (separate PMC passes) and the FETCH / WRITE sizes behind roofline.traffic.
# usage: tools/gpu_round3_profile.sh <tag>      (then on the build side: python tools/make_pmc_json.py <tag> 16 <git hash>)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r03_x}
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 300 gpurun_out/${TAG}_bench.json; echo
rm -rf /tmp/prof_t && rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/prof_t.log 2>&1
DB=$(find /tmp/prof_t -name '*.db' | head -1)
python tools/prof_summary.py $DB > gpurun_out/${TAG}_kernel_trace.txt 2>&1
python tools/prof_cu_time.py $DB > gpurun_out/${TAG}_cu_time.txt 2>&1
head -14 gpurun_out/${TAG}_kernel_trace.txt
bash tools/gpu_pmc_chain16.sh ${TAG} 16 > /dev/null 2>&1
cat gpurun_out/${TAG}_pmc_chain16.txt
