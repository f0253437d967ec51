#!/bin/bash
# stage times of the SGM mode under the JN_SGM_DBG profiling switches (results are WRONG under any switch; attribution only):
#   bash scripts/sgm_dbg_times.sh "0 1 2 4" [batch] [extra bench args]
# k_sw_w: 1 = no wait for / load of the producer block's columns, 2 = no right-image minima (LDS atomics + flush), 4 = no volume loads / stores,
#         16 = no per-row input fetch, 32 = no waiting on the neighbour strip's LDS counters
export JN_STEREO_LIB=${JN_STEREO_LIB:-${GRAFT_REPO_ROOT:-$(pwd)}/jackal_navigation_amd/libjn_stereo_hooks.so}   # the switches used below exist in the hooks build only (csrc/hooks.h)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
b=${2:-32}
for d in $1; do
  echo "JN_SGM_DBG=$d batch=$b: $(JN_SGM_DBG=$d timeout 200 python bench.py --mode sgm --steps 3 --warmup 1 --no-cpu-baseline --batch $b $3 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["stage_ms_per_batch"])')"
done
