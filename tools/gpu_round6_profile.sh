#!/bin/bash
# Round-6 evidence on the final library: the default bench line, a rocprofv3 kernel trace of the same pipelined command, the CU-time table of the
# headline loop, the single-scene launch table, digests, the erratum microbenchmarks, and the PMC passes of the policy launch (own runs, --kernel-trace only).
# (the kernel trace runs bench.py at FOUR rollouts in flight, where roofline.avg_launch_ms is measured: deeper, a launch's duration includes its workgroups' wait for CUs)
# usage: tools/gpu_round6_profile.sh <tag> <git hash of the tree>
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r06_x}
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 300 gpurun_out/${TAG}_bench.json; echo
rm -rf /tmp/prof_t && rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --inflight 4 > /tmp/prof_t.log 2>&1
python tools/prof_summary.py $(find /tmp/prof_t -name '*.db' | head -1) > gpurun_out/${TAG}_kernel_trace.txt 2>&1
rm -rf /tmp/prof_h && rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o t -- python tools/gpu_headline_loop.py > /tmp/prof_h.log 2>&1
DBH=$(find /tmp/prof_h -name '*.db' | head -1)
python tools/prof_cu_time.py $DBH > gpurun_out/${TAG}_headline_cu_time.txt 2>&1
python tools/prof_by_grid.py $DBH > gpurun_out/${TAG}_headline_by_grid.txt 2>&1
head -14 gpurun_out/${TAG}_headline_cu_time.txt
rm -rf /tmp/prof_s && rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o t -- python tools/gpu_single_timeline.py > /tmp/prof_s.log 2>&1
python tools/prof_rollout_gaps.py $(find /tmp/prof_s -name '*.db' | head -1) 0 > gpurun_out/${TAG}_single_scene_launches.txt 2>&1
python tools/gpu_traj_digest.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_digest.txt
{ tools/mb/mb_mixsel; tools/mb/mb_pksgpr3 20000; } > gpurun_out/${TAG}_mb_erratum.txt 2>&1
tail -8 gpurun_out/${TAG}_mb_erratum.txt | cut -c1-200
bash tools/gpu_pmc_chain16.sh ${TAG} 16 > /dev/null 2>&1
python tools/make_pmc_json.py ${TAG} 16 ${2:-unknown}
