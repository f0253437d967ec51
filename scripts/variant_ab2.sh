#!/bin/bash
# A/B of a compile-time variant that touches kernels.hip AND jn_api.cpp (inside gpurun):  bash scripts/variant_ab2.sh "<-DFLAGS>" <kernel pattern>
FLAGS=$1; PAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; C=$R/jackal_navigation_amd/csrc; T=/tmp/variant_ab2; mkdir -p $T
HIP_RT_DIR=$(python3 -c "import os,torch;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
cd $C
/opt/rocm/bin/hipcc -O3 -mavx2 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -w $FLAGS --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -c kernels.hip -o $T/kernels.o || exit 1
/opt/rocm/lib/llvm/bin/clang++ -x c++ -O3 -mavx2 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -w $FLAGS -include stddef.h -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c jn_api.cpp -o $T/jn_api.o || exit 1
OBJS=$(ls _build/*.o | grep -v "/kernels.o\|/jn_api.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $T/libjn_stereo_variant.so $OBJS $T/kernels.o $T/jn_api.o -L$HIP_RT_DIR -lamdhip64 -Wl,-rpath,$HIP_RT_DIR -lpthread -ldl || exit 1
cd $R
line() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, check", j["check"]["ok"])'; }
for i in 1 2 3; do
  echo "this tree: $(line "$@")"
  echo "with $FLAGS: $(JN_STEREO_LIB=$T/libjn_stereo_variant.so line "$@")"
done
echo "GPU route, this tree: $(JN_GPU_DELAUNAY=1 line "$@")"
echo "GPU route, with $FLAGS: $(JN_GPU_DELAUNAY=1 JN_STEREO_LIB=$T/libjn_stereo_variant.so line "$@")"
echo "one slot under rocprofv3, with $FLAGS: $(JN_STEREO_LIB=$T/libjn_stereo_variant.so bash scripts/prof.sh vab_b "$@" | grep -E "$PAT" | tr -s ' ' | tr '\n' '|')"
