R=${GRAFT_REPO_ROOT:-$(pwd)}; C=$R/jackal_navigation_amd/csrc; T=/tmp/variant_ab; mkdir -p $T
HIP_RT_DIR=$(python3 -c "import os,torch;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
line() { python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, check", j["check"]["ok"])'; }
for th in 512 256; do
  cd $C
  /opt/rocm/bin/hipcc -O3 -mavx2 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -w -DJN_AB_ARR_THREADS=$th --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -c kernels.hip -o $T/kernels.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $T/libjn_v$th.so $(ls _build/*.o | grep -v "/kernels.o") $T/kernels.o -L$HIP_RT_DIR -lamdhip64 -Wl,-rpath,$HIP_RT_DIR -lpthread -ldl || exit 1
  cd $R
done
for i in 1 2; do
  echo "GPU route, 1024 threads: $(JN_GPU_DELAUNAY=1 line)"
  for th in 512 256; do echo "GPU route, $th threads: $(JN_STEREO_LIB=$T/libjn_v$th.so JN_GPU_DELAUNAY=1 line)"; done
done
echo "host route, 1024 threads: $(JN_GPU_DELAUNAY=0 line)"
for th in 512 256; do echo "host route, $th threads: $(JN_STEREO_LIB=$T/libjn_v$th.so JN_GPU_DELAUNAY=0 line)"; done
echo "host route, 1024 threads: $(JN_GPU_DELAUNAY=0 line)"
for th in 512 256; do echo "alone, $th threads: $(JN_STEREO_LIB=$T/libjn_v$th.so JN_GPU_DELAUNAY=1 bash scripts/prof.sh arrthr_$th | grep -E "k_arrange" | tr -s ' ')"; done
