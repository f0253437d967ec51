#!/bin/bash
# k_dense2's phases by its JN_DENSE_DBG switches (results wrong, timing only): the kernel alone (ms per launch, HIP events) and the pipelined rate.
# Usage (inside gpurun): bash scripts/dense_dbg_times.sh "0 4 12 20 28" [env assignments...]
export JN_STEREO_LIB=${JN_STEREO_LIB:-${GRAFT_REPO_ROOT:-$(pwd)}/jackal_navigation_amd/libjn_stereo_hooks.so}   # the switches used below exist in the hooks build only (csrc/hooks.h)
vals=${1:-"0 1 2 3 4 12 20 28 64 0"}; shift
for d in $vals; do
  echo "JN_DENSE_DBG=$d $*: $(env "$@" JN_DENSE_DBG=$d python3 bench.py --steps 20 --warmup 3 --min-time 0.5 --no-cpu-baseline --no-latency-config 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); r=j["roofline"]; print(r.get("ms_per_launch"), "ms alone,", r.get("ms_per_launch_pipelined"), "ms pipelined,", j["value"], "pairs/s")')"
done
