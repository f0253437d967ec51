#!/bin/bash
# is `volatile` on k_delaunay's LDS accessors a workaround for something?  Builds the library with plain (non-volatile) LDS
# pointers in delaunay_gpu.hip (-DJN_DT_NO_VOLATILE), shows what the compiler merges, and runs the triangulation tests against it.
#   gpurun -- bash scripts/probes/dt_no_volatile.sh
# (the same-box timing of the two builds that profiles/r06_dt_no_volatile.txt carries below the test results was taken once, on round 6's first tree)
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/jackal_navigation_amd/csrc
T=/tmp/dt_novol; mkdir -p $T
HIP_RT_DIR=$(python3 -c "import os,torch;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
cd $C
/opt/rocm/bin/hipcc -O3 -mavx2 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -w -DJN_DT_NO_VOLATILE --offload-arch=gfx950 -c delaunay_gpu.hip -o $T/delaunay_gpu.o || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -w -DJN_DT_NO_VOLATILE --offload-arch=gfx950 -S --cuda-device-only delaunay_gpu.hip -o $T/novol.s
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -w --offload-arch=gfx950 -S --cuda-device-only delaunay_gpu.hip -o $T/vol.s
for v in vol novol; do echo "$v: LDS writes by width: $(grep -oE 'ds_write[0-9a-z_]*' $T/$v.s | sort | uniq -c | tr '\n' ' ')| reads: $(grep -oE 'ds_read[0-9a-z_]*' $T/$v.s | sort | uniq -c | tr '\n' ' ')"; done
OBJS=$(ls _build/*.o | grep -v delaunay_gpu.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $T/libjn_stereo_novol.so $OBJS $T/delaunay_gpu.o -L$HIP_RT_DIR -lamdhip64 -Wl,-rpath,$HIP_RT_DIR -lpthread -ldl || exit 1
cd $R
echo "--- tests/test_gpu_delaunay.py against the non-volatile build"
JN_STEREO_LIB=$T/libjn_stereo_novol.so python3 -m pytest tests/test_gpu_delaunay.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | head -8
echo "--- and against the product build"
python3 -m pytest tests/test_gpu_delaunay.py -q -x 2>&1 | grep -E "passed|failed" | head -3
