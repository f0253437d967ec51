#!/bin/bash
# Release switches of the host side against the headline (inside gpurun): does polling longer / zero-copy payload / no stage events move the rate?
line() { python3 bench.py --gpus 1 --steps 40 --warmup 8 --no-cpu-baseline --no-latency-config --no-alone-leg "$@" 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, host cores busy", j["host_cpu"].get("cores_total"), ", host stage", j["stage_ms_per_batch"].get("host_stage"), "d2h", j["stage_ms_per_batch"].get("d2h"), "h2d", j["stage_ms_per_batch"].get("h2d"))'; }
echo "default: $(line)"
echo "JN_WAIT_SPIN_US=300: $(JN_WAIT_SPIN_US=300 line)"
echo "JN_WAIT_SPIN_US=2000: $(JN_WAIT_SPIN_US=2000 line)"
echo "JN_POOL_SPIN_US=100: $(JN_POOL_SPIN_US=100 line)"
echo "JN_POOL_SPIN_US=400: $(JN_POOL_SPIN_US=400 line)"
echo "JN_ZERO_COPY=1: $(JN_ZERO_COPY=1 line)"
echo "JN_STAGE_EVENTS=0: $(JN_STAGE_EVENTS=0 line)"
echo "JN_WAIT_SPIN_US=300 JN_POOL_SPIN_US=100: $(JN_WAIT_SPIN_US=300 JN_POOL_SPIN_US=100 line)"
echo "host threads 14: $(line --host-threads 14)"
echo "host threads 20: $(line --host-threads 20)"
echo "default: $(line)"
