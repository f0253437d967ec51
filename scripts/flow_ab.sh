#!/bin/bash
# Same-box A/B of the two descriptor data flows (inside gpurun): the driver's command twice each, alternating, then one-slot kernel tables.
#   planes: k_sobel_planes -> k_support_lds<PL> -> k_owner + k_dense_row      desc: k_descriptor_fused -> k_support_lds -> k_dense2 (JN_DESC_FLOW=desc)
line() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(j["value"], "pairs/s,", j["ms_per_step"], "ms/step, dense stage alone", r.get("ms_per_launch"), "ms, lone 640x480 pair", j["latency_config"]["ms_per_frame"], "ms, check", j["check"]["ok"])'; }
for i in 1 2; do
  echo "planes: $(JN_DENSE_COOP=0 line)"
  echo "desc:   $(JN_DESC_FLOW=desc line)"
done
echo "--- planes, one slot, rocprofv3 kernel stats"; JN_DENSE_COOP=0 bash scripts/prof.sh flow_ab_planes | head -22
echo "--- desc, one slot, rocprofv3 kernel stats"; JN_DESC_FLOW=desc bash scripts/prof.sh flow_ab_desc | head -22
