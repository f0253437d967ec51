#!/bin/bash
# lone-pair latency of two builds of the library on one box (inside gpurun): bash scripts/lone_ab_lib.sh <libA> <libB> [rounds]
for i in $(seq ${3:-3}); do
  for lib in $1 $2; do
    echo "$lib: $(JN_STEREO_LIB=$lib HT=8 LONE_ONLY=${LONE_ONLY:-} timeout 200 python3 scripts/latency_check.py 2>/dev/null | grep "device pointers" | cut -c1-100 | tr '\n' '|')"
  done
done
