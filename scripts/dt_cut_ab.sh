#!/bin/bash
# Would 1280x720 sides gain from the cut form too (subtrees in LDS + top levels in global memory, JN_DT_WHOLE in the hooks build)?  No: profiles/r06_dt_cut_720p_ab.txt
export JN_STEREO_LIB=$PWD/jackal_navigation_amd/libjn_stereo_hooks.so
line() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, host cores busy", j["host_cpu"].get("cores_total"), ", check", j["check"].get("ok"))'; }
echo "720p GPU route, whole sides in LDS: $(JN_GPU_DELAUNAY=1 line)"
for w in 2100 1100 600 300; do echo "720p GPU route, cut below $w vertices: $(JN_GPU_DELAUNAY=1 JN_DT_WHOLE=$w line)"; done
echo "720p GPU route, whole sides in LDS: $(JN_GPU_DELAUNAY=1 line)"
JN_GPU_DELAUNAY=1 JN_DT_WHOLE=1100 bash scripts/prof.sh dt_cut | grep -E "k_delaunay|k_arrange"
JN_GPU_DELAUNAY=1 JN_DT_WHOLE=300 bash scripts/prof.sh dt_cut2 | grep -E "k_delaunay|k_arrange"
