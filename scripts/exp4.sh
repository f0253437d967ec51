R=$PWD; C=$R/jackal_navigation_amd/csrc; T=/tmp/dt_novol; mkdir -p $T
HIP_RT_DIR=$(python3 -c "import os,torch;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
cd $C
/opt/rocm/bin/hipcc -O3 -mavx2 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -w -DJN_DT_NO_VOLATILE --offload-arch=gfx950 -c delaunay_gpu.hip -o $T/delaunay_gpu.o || exit 1
OBJS=$(ls _build/*.o | grep -v delaunay_gpu.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $T/libjn_stereo_novol.so $OBJS $T/delaunay_gpu.o -L$HIP_RT_DIR -lamdhip64 -Wl,-rpath,$HIP_RT_DIR -lpthread -ldl || exit 1
cd $R
line() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, check", j["check"]["ok"])'; }
for i in 1 2; do
echo "GPU route, volatile: $(JN_GPU_DELAUNAY=1 line)"
echo "GPU route, non-volatile: $(JN_STEREO_LIB=$T/libjn_stereo_novol.so JN_GPU_DELAUNAY=1 line)"
done
JN_STEREO_LIB=$T/libjn_stereo_novol.so JN_GPU_DELAUNAY=1 bash scripts/prof.sh gpu_dt_y | grep -E "k_delaunay|k_arrange"
