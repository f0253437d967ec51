#!/bin/bash
# stage times of the SGM mode under the JN_SGM_DBG profiling switches:  bash scripts/sgm_dbg_times.sh "0 1 2 3"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for d in $1; do
  echo "JN_SGM_DBG=$d: $(JN_SGM_DBG=$d timeout 200 python bench.py --mode sgm --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["stage_ms_per_batch"])')"
done
