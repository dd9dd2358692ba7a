mkdir -p gpurun_out
timeout 400 python tools/gpu_search_ab.py > gpurun_out/search_ab.txt 2>&1; echo rc=$? >> gpurun_out/search_ab.txt
grep -v small gpurun_out/search_ab.txt
bash tools/_run2.sh
