#!/bin/bash
# Kernel-time effect of compile-time flags of kernels.hip (-DJN_AB_..., experiment macros) on one box, inside gpurun:
#   bash scripts/exp_ab.sh <kernel name pattern> <flag> [<flag> ...]
# One slot under rocprofv3 for every flag, this tree's library again after each (the box's clocks drift: read the RATIO of the kernel under
# test to one that the flag does not touch, e.g. "k_dense_row|k_support_lds").
PAT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; C=$R/jackal_navigation_amd/csrc; T=/tmp/exp_ab; mkdir -p $T
HIP_RT_DIR=$(python3 -c "import os,torch;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
bash scripts/prof.sh exp_base > /dev/null      # (the box's first run is not comparable)
base() { echo "this tree: $(bash scripts/prof.sh exp_base | grep -E "$PAT" | tr -s ' ' | tr '\n' '|')"; }
base
for FLAG in "$@"; do
  cd $C
  /opt/rocm/bin/hipcc -O3 -mavx2 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -w $FLAG --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -c kernels.hip -o $T/kernels.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $T/libjn_stereo_variant.so $(ls _build/*.o | grep -v "/kernels.o") $T/kernels.o -L$HIP_RT_DIR -lamdhip64 -Wl,-rpath,$HIP_RT_DIR -lpthread -ldl || exit 1
  cd $R
  echo "with $FLAG: $(JN_STEREO_LIB=$T/libjn_stereo_variant.so bash scripts/prof.sh exp_v | grep -E "$PAT" | tr -s ' ' | tr '\n' '|')"
  base
done
