#!/bin/bash
# Pairs per batch x batches in flight (inside gpurun): more kernels in flight than four lower the rate, whatever the batch size.
line() { python3 bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, host cores busy", j["host_cpu"].get("cores_total"), ", check", j["check"].get("ok"))'; }
echo "batch 32 slots 4: $(line)"
echo "batch 32 slots 5: $(line --slots 5)"
echo "batch 16 slots 4: $(line --batch 16 --steps 80)"
echo "batch 16 slots 6: $(line --batch 16 --slots 6 --steps 80)"
echo "batch 16 slots 8: $(line --batch 16 --slots 8 --steps 80)"
echo "batch 8 slots 8: $(line --batch 8 --slots 8 --steps 160)"
echo "batch 32 slots 4 GPU route: $(JN_GPU_DELAUNAY=1 line)"
echo "batch 16 slots 8 GPU route: $(JN_GPU_DELAUNAY=1 line --batch 16 --slots 8 --steps 80)"
echo "batch 16 slots 6 GPU route: $(JN_GPU_DELAUNAY=1 line --batch 16 --slots 6 --steps 80)"
echo "batch 64 slots 4: $(line --batch 64 --steps 20)"
echo "batch 32 slots 4: $(line)"
