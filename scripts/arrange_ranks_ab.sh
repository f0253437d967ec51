#!/bin/bash
# k_arrange's rank form against its sort forms (JN_ARRANGE_SORTS=1, hooks build) on one box, both triangulation routes, 1280x720 and 1920x1080.
H=$PWD/jackal_navigation_amd/libjn_stereo_hooks.so
line() { JN_STEREO_LIB=$H python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, check", j["check"]["ok"])'; }
HD="--width 1920 --height 1080 --disp 256 --batch 8"
for i in 1 2; do
  for v in 1 0; do echo "1280x720 host route, JN_ARRANGE_SORTS=$v: $(JN_ARRANGE_SORTS=$v JN_GPU_DELAUNAY=0 line)"; done
  for v in 1 0; do echo "1280x720 GPU route, JN_ARRANGE_SORTS=$v: $(JN_ARRANGE_SORTS=$v JN_GPU_DELAUNAY=1 line)"; done
done
for v in 1 0 1 0; do echo "1920x1080 D=256 batch 8 GPU route, JN_ARRANGE_SORTS=$v: $(JN_ARRANGE_SORTS=$v JN_GPU_DELAUNAY=1 line $HD)"; done
for v in 1 0; do echo "1920x1080, one slot under rocprofv3, JN_ARRANGE_SORTS=$v: $(JN_STEREO_LIB=$H JN_ARRANGE_SORTS=$v JN_GPU_DELAUNAY=1 bash scripts/prof.sh arr_hd_$v $HD | grep -E "k_arrange|k_delaunay" | tr -s ' ' | tr '\n' '|')"; done
