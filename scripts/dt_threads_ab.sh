#!/bin/bash
# k_delaunay / k_delaunay_sub with fewer threads a workgroup (-DJN_AB_DT_THREADS=512 / 256) against 1024, GPU route, one box (inside gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}; C=$R/jackal_navigation_amd/csrc; T=/tmp/variant_ab; mkdir -p $T
HIP_RT_DIR=$(python3 -c "import os,torch;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
line() { python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, check", j["check"]["ok"])'; }
for th in 512 256; do
  cd $C
  /opt/rocm/bin/hipcc -O3 -mavx2 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -w -DJN_AB_DT_THREADS=$th --offload-arch=gfx950 -c delaunay_gpu.hip -o $T/delaunay_gpu.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $T/libjn_dt$th.so $(ls _build/*.o | grep -v "/delaunay_gpu.o") $T/delaunay_gpu.o -L$HIP_RT_DIR -lamdhip64 -Wl,-rpath,$HIP_RT_DIR -lpthread -ldl || exit 1
  cd $R
done
for i in 1 2 3; do
  echo "GPU route, 1024 threads: $(JN_GPU_DELAUNAY=1 line)"
  for th in 512 256; do echo "GPU route, $th threads: $(JN_STEREO_LIB=$T/libjn_dt$th.so JN_GPU_DELAUNAY=1 line)"; done
done
HD="--width 1920 --height 1080 --disp 256 --batch 8"
echo "1080p GPU route, 1024 threads: $(JN_GPU_DELAUNAY=1 line $HD)"
for th in 512 256; do echo "1080p GPU route, $th threads: $(JN_STEREO_LIB=$T/libjn_dt$th.so JN_GPU_DELAUNAY=1 line $HD)"; done
echo "alone, 1024 threads: $(JN_GPU_DELAUNAY=1 bash scripts/prof.sh dtthr_1024 | grep -E "k_delaunay|k_arrange" | tr -s ' ' | tr '\n' '|')"
for th in 512 256; do echo "alone, $th threads: $(JN_STEREO_LIB=$T/libjn_dt$th.so JN_GPU_DELAUNAY=1 bash scripts/prof.sh dtthr_$th | grep -E "k_delaunay|k_arrange" | tr -s ' ' | tr '\n' '|')"; JN_STEREO_LIB=$T/libjn_dt$th.so python -m pytest tests/test_gpu_delaunay.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1; done
