#!/bin/bash
# Round-4 evidence: the default bench line, a rocprofv3 kernel trace of the same (pipelined) command with its CU-time / by-grid
# tables, the hardware counters of the k_chain16 policy launch (separate PMC passes; FETCH / WRITE behind roofline.traffic), the
# counters of the row-tile map PointNet, token noise against the fp64 oracle, and the single-scene launch table.
# usage: tools/gpu_round4_profile.sh <tag>      (then on the build side: python tools/make_pmc_json.py <tag> 16 <git hash>)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r04_x}
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 300 gpurun_out/${TAG}_bench.json; echo
rm -rf /tmp/prof_t && rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/prof_t.log 2>&1
DB=$(find /tmp/prof_t -name '*.db' | head -1)
python tools/prof_summary.py $DB > gpurun_out/${TAG}_kernel_trace.txt 2>&1
rm -rf /tmp/prof_h && rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o t -- python tools/gpu_headline_loop.py > /tmp/prof_h.log 2>&1
DBH=$(find /tmp/prof_h -name '*.db' | head -1)
python tools/prof_cu_time.py $DBH > gpurun_out/${TAG}_headline_cu_time.txt 2>&1
python tools/prof_by_grid.py $DBH > gpurun_out/${TAG}_headline_by_grid.txt 2>&1
head -14 gpurun_out/${TAG}_headline_cu_time.txt
rm -rf /tmp/prof_s && rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o t -- python tools/gpu_single_timeline.py > /tmp/prof_s.log 2>&1
python tools/prof_rollout_gaps.py $(find /tmp/prof_s -name '*.db' | head -1) 0 > gpurun_out/${TAG}_single_scene_launches.txt 2>&1
bash tools/gpu_pmc_chain16.sh ${TAG} 16 > /dev/null 2>&1
cat gpurun_out/${TAG}_pmc_chain16.txt
tools/gpu_rt_pmc.sh ${TAG}_pointnet_rt5 "k_pointnet_rt<5" PS_RT_MT=5 PS_RT_WHICH=0 PS_RT_N=8192 PS_RT_P=19 > /dev/null 2>&1
(python tools/gpu_token_error.py 5; python tools/gpu_token_error.py 0) 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_token_error.txt
python tools/gpu_rowtile_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_rowtile_pointnet.txt
python tools/gpu_rowtile_rollout.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_rowtile_rollout.txt
python tools/gpu_traj_digest.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_digest.txt
