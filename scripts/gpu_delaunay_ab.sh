#!/bin/bash
# Triangulation on the GPU (delaunay_gpu.hip, no host stage) against the host stage, as a function of the host cores the process owns
# (inside gpurun).  taskset narrows the affinity; the library's own choice (no JN_GPU_DELAUNAY) follows it: GPU route below 14 cores.
line() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency-config --no-pin "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, host cores busy", j["host_cpu"].get("cores_total"), ", check", j["check"]["ok"])'; }
for T in 2 4 8 12 16; do
  echo "cores $T  host route: $(JN_GPU_DELAUNAY=0 taskset -c 0-$((T-1)) bash -c "$(declare -f line); line --host-threads $T")"
  echo "cores $T  GPU route:  $(JN_GPU_DELAUNAY=1 taskset -c 0-$((T-1)) bash -c "$(declare -f line); line --host-threads $T")"
done
# what parallel.pin_rank gives a rank on a 2 x 64-core node with 8 GPUs: 16 physical cores with their SMT siblings (the library's rule counts logical CPUs: host route)
SIB=$(python3 - <<'PY'
import os
sib=set()
for c in range(16):
    try: sib.update(int(x) for x in open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list"%c).read().replace("-",",").split(","))
    except OSError: sib.add(c)
print(",".join(str(c) for c in sorted(sib)))
PY
)
echo "16 physical cores + siblings (16 + 16 logical CPUs) host route: $(JN_GPU_DELAUNAY=0 taskset -c $SIB bash -c "$(declare -f line); line --host-threads 16")"
echo "16 physical cores + siblings (16 + 16 logical CPUs) GPU route:  $(JN_GPU_DELAUNAY=1 taskset -c $SIB bash -c "$(declare -f line); line --host-threads 16")"
echo "16 physical cores + siblings (16 + 16 logical CPUs) library's choice: $(taskset -c $SIB bash -c "$(declare -f line); line --host-threads 16")"
echo "all cores, library's choice: $(line)"
echo "all cores, JN_GPU_DELAUNAY=1: $(JN_GPU_DELAUNAY=1 line)"
echo "all cores, JN_GPU_DELAUNAY=0: $(JN_GPU_DELAUNAY=0 line)"
echo "--- one slot, rocprofv3 kernel stats, JN_GPU_DELAUNAY=1"; JN_GPU_DELAUNAY=1 bash scripts/prof.sh gpu_dt_ab | head -8
python3 scripts/dt_levels.py 2>&1 | grep k_delaunay | head -2
