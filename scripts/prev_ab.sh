#!/bin/bash
# Same-box A/B of this tree's library against jackal_navigation_amd/_ab/libjn_stereo_prev.so (built by hand from an earlier commit's
# kernels.hip; not tracked): one slot under rocprofv3, interleaved, the named kernels' averages — read the RATIO to a kernel that did
# not change (the box's clocks move every kernel of a run together) — then the pipelined rate on both routes.
PAT=${1:-"k_dense_row|k_support_lds"}; N=${2:-3}
P=$PWD/jackal_navigation_amd/_ab/libjn_stereo_prev.so
bash scripts/prof.sh warm > /dev/null
for i in $(seq $N); do
  echo "previous: $(JN_STEREO_LIB=$P bash scripts/prof.sh ab_prev | grep -E "$PAT" | tr -s ' ' | tr '\n' '|')"
  echo "this tree: $(bash scripts/prof.sh ab_new | grep -E "$PAT" | tr -s ' ' | tr '\n' '|')"
done
line() { python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, check", j["check"]["ok"])'; }
for i in 1 2; do
  echo "previous, host route: $(JN_STEREO_LIB=$P line)"
  echo "this tree, host route: $(line)"
done
echo "previous, GPU route: $(JN_STEREO_LIB=$P JN_GPU_DELAUNAY=1 line)"
echo "this tree, GPU route: $(JN_GPU_DELAUNAY=1 line)"
