line() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, host cores busy", j["host_cpu"].get("cores_total"), ", check", j["check"]["ok"])'; }
echo "host route: $(JN_GPU_DELAUNAY=0 line)"
echo "GPU route: $(JN_GPU_DELAUNAY=1 line)"
echo "GPU route: $(JN_GPU_DELAUNAY=1 line)"
JN_GPU_DELAUNAY=1 bash scripts/prof.sh gpu_dt_x | head -12
python3 scripts/dt_levels.py 2>&1 | grep k_delaunay | head -2
