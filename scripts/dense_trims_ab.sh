line() { python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, check", j["check"]["ok"])'; }
P=$PWD/jackal_navigation_amd/_ab/libjn_stereo_prev.so
for i in 1 2 3; do
  echo "this tree, host route: $(line)"
  echo "previous commit's kernels, host route: $(JN_STEREO_LIB=$P line)"
done
for i in 1 2; do
  echo "this tree, GPU route: $(JN_GPU_DELAUNAY=1 line)"
  echo "previous commit's kernels, GPU route: $(JN_STEREO_LIB=$P JN_GPU_DELAUNAY=1 line)"
done
