#!/bin/bash
# 1920x1080 D=256 batch 8 on both triangulation routes by host cores (inside gpurun): the sides (~11 k support points) go through
# k_delaunay_sub / k_delaunay_top on the GPU route (round 6).
line() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, host cores busy", j["host_cpu"].get("cores_total"), ", check", j["check"].get("ok"))'; }
echo "720p host route: $(JN_GPU_DELAUNAY=0 line)"
echo "720p GPU route: $(JN_GPU_DELAUNAY=1 line)"
HD="--width 1920 --height 1080 --disp 256 --batch 8"
echo "1080p D=256 batch 8, host route, all cores: $(JN_GPU_DELAUNAY=0 line $HD)"
echo "1080p D=256 batch 8, GPU route, all cores: $(JN_GPU_DELAUNAY=1 line $HD)"
echo "1080p D=256 batch 8, host route, 2 cores: $(JN_GPU_DELAUNAY=0 taskset -c 0-1 bash -c "$(declare -f line); line $HD --host-threads 2 --no-pin")"
echo "1080p D=256 batch 8, GPU route, 2 cores: $(JN_GPU_DELAUNAY=1 taskset -c 0-1 bash -c "$(declare -f line); line $HD --host-threads 2 --no-pin")"
echo "1080p D=256 batch 8, host route, 4 cores: $(JN_GPU_DELAUNAY=0 taskset -c 0-3 bash -c "$(declare -f line); line $HD --host-threads 4 --no-pin")"
JN_GPU_DELAUNAY=1 bash scripts/prof.sh hd_dt $HD | head -8
