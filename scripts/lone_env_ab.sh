#!/bin/bash
# lone 640x480 pair (latency-mode handle, device pointers) under a few settings, alternating (inside gpurun): medians of scripts/latency_check.py
run() { env "$@" HT=8 LONE_ONLY=1 timeout 200 python3 scripts/latency_check.py 2>/dev/null | grep "device pointers" | sed 's/.*p10 [0-9.]* median \([0-9.]*\).*/\1/'; }
for i in 1 2 3; do
  echo "round $i: gate=1 $(run JN_GATE_STAGE_B=1)  gate=0 $(run JN_GATE_STAGE_B=0)  gate=1+fused-post $(run JN_GATE_STAGE_B=1 JN_POST_FUSED_MIN_PIXELS=0)  gate=0+fused-post $(run JN_GATE_STAGE_B=0 JN_POST_FUSED_MIN_PIXELS=0)  desc-flow gate=1 $(run JN_GATE_STAGE_B=1 JN_DESC_FLOW=desc)"
done
