mkdir -p gpurun_out
for r in 1 2 3; do
for si in 1 0; do echo -n "search_impl $si: "; PS_SEARCH_IMPL=$si PS_STEPS=200 timeout 200 python tools/gpu_headline_loop.py 2>&1 | tail -1; done
done > gpurun_out/headline_ab.txt 2>&1
cat gpurun_out/headline_ab.txt
