#!/bin/bash
# What k_delaunay costs the pipeline and why (hooks build, inside gpurun): a kernel that only waits, queued behind the real one - one wave and no LDS
# (latency alone) or 1024 threads and 152 KB (a CU nothing else fits on).  profiles/r06_dt_dummy_ab.txt
export JN_STEREO_LIB=$PWD/jackal_navigation_amd/libjn_stereo_hooks.so
line() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, host cores busy", j["host_cpu"].get("cores_total"), ", check", j["check"]["ok"])'; }
echo "host route: $(JN_GPU_DELAUNAY=0 line)"
echo "GPU route: $(JN_GPU_DELAUNAY=1 line)"
echo "GPU route + dummy 1 wave, no LDS, 870 us: $(JN_GPU_DELAUNAY=1 JN_DT_DUMMY=1 line)"
echo "GPU route + dummy 16 waves, 152 KB, 870 us: $(JN_GPU_DELAUNAY=1 JN_DT_DUMMY=2 line)"
echo "GPU route + dummy 1 wave, 152 KB, 870 us: $(JN_GPU_DELAUNAY=1 JN_DT_DUMMY=3 line)"
echo "GPU route + dummy 16 waves, no LDS, 870 us: $(JN_GPU_DELAUNAY=1 JN_DT_DUMMY=4 line)"
echo "GPU route + dummy 1 wave, no LDS, 2000 us: $(JN_GPU_DELAUNAY=1 JN_DT_DUMMY=1 JN_DT_DUMMY_US=2000 line)"
echo "GPU route + dummy 16 waves, 152 KB, 400 us: $(JN_GPU_DELAUNAY=1 JN_DT_DUMMY=2 JN_DT_DUMMY_US=400 line)"
echo "GPU route: $(JN_GPU_DELAUNAY=1 line)"
