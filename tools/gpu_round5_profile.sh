#!/bin/bash
# Round-5 evidence: the default bench line, a rocprofv3 kernel trace of the same (pipelined) command with the CU-time / by-grid tables of
# the headline loop, the hardware counters of the k_chain16 policy launch (separate PMC passes; FETCH / WRITE behind roofline.traffic),
# of its other launch shapes (generator / a2a / s2s) and of the condition layers, the single-scene launch table, the streaming
# pipeline by depth, digests and the feature / butterfly bit checks.
# usage: tools/gpu_round5_profile.sh <tag>      (then on the build side: python tools/make_pmc_json.py <tag> 16 <git hash>)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r05_x}
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
tools/gpu_pmc_single_chain.sh ${TAG} > /dev/null 2>&1
bash tools/gpu_pmc_chain16.sh ${TAG} 16 > /dev/null 2>&1
cat gpurun_out/${TAG}_pmc_chain16.txt
tools/gpu_pmc_by_grid.sh ${TAG}_c16_other "k_chain16<8, false" > /dev/null 2>&1
tools/gpu_pmc_by_grid.sh ${TAG}_attn_chain444 "k_attn_chain<4, 4, 4" > /dev/null 2>&1
python tools/gpu_pipeline_depth.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_pipeline_depth.txt
python tools/gpu_traj_digest.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_digest.txt
tools/mb/mb_feat > gpurun_out/${TAG}_mb_feat.txt 2>&1
