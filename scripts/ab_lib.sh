#!/bin/bash
# Same-box A/B of two builds of the library (inside gpurun): bash scripts/ab_lib.sh <libA.so> <libB.so> [rounds] [bench args...]
# Alternates the two builds `rounds` times; prints pairs/s and the dominant kernel's alone-leg ms per run.
a=$1; b=$2; rounds=${3:-3}; shift 3
for i in $(seq $rounds); do
  for lib in $a $b; do
    JN_STEREO_LIB=$lib python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-latency-config "$@" 2>/dev/null | tail -1 |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', d['value'], r.get('ms_per_launch'), r.get('ms_per_launch_pipelined'))"
  done
done
