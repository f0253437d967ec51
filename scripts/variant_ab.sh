#!/bin/bash
# A/B of a compile-time variant of kernels.hip on one box (inside gpurun):  bash scripts/variant_ab.sh <-DFLAG> <kernel name pattern> [bench args]
# Builds the library once more with the flag (objects of the other files are reused), runs the driver's command on both libraries twice,
# and one slot under rocprofv3 for the named kernel's average duration.
FLAG=$1; PAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; C=$R/jackal_navigation_amd/csrc; T=/tmp/variant_ab; mkdir -p $T
HIP_RT_DIR=$(python3 -c "import os,torch;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
cd $C
/opt/rocm/bin/hipcc -O3 -mavx2 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -w $FLAG --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -c kernels.hip -o $T/kernels.o || exit 1
OBJS=$(ls _build/*.o | grep -v "/kernels.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $T/libjn_stereo_variant.so $OBJS $T/kernels.o -L$HIP_RT_DIR -lamdhip64 -Wl,-rpath,$HIP_RT_DIR -lpthread -ldl || exit 1
cd $R
line() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, check", j["check"]["ok"])'; }
for i in 1 2; do
  echo "this tree: $(line "$@")"
  echo "with $FLAG: $(JN_STEREO_LIB=$T/libjn_stereo_variant.so line "$@")"
done
echo "one slot under rocprofv3, this tree:   $(bash scripts/prof.sh vab_a "$@" | grep -E "$PAT" | tr -s ' ' | tr '\n' '|')"
echo "one slot under rocprofv3, with $FLAG: $(JN_STEREO_LIB=$T/libjn_stereo_variant.so bash scripts/prof.sh vab_b "$@" | grep -E "$PAT" | tr -s ' ' | tr '\n' '|')"
