#!/bin/bash
export JN_STEREO_LIB=${JN_STEREO_LIB:-${GRAFT_REPO_ROOT:-$(pwd)}/jackal_navigation_amd/libjn_stereo_hooks.so}   # the switches used below exist in the hooks build only (csrc/hooks.h)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 300 python scripts/sgm_debug.py > gpurun_out/sgm_ns_debug.txt 2>&1; echo "debug rc=$?"; grep -E "differ|ALL OK|FAIL|rror" gpurun_out/sgm_ns_debug.txt | head
for ns in 7 5 3; do
  echo "NS=$ns: $(JN_SGM_NS=$ns timeout 200 python bench.py --mode sgm --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["stage_ms_per_batch"], j["check"]["ok"])')"
done
JN_SGM_NS=5 timeout 200 python scripts/sgm_debug.py 333 101 64 2 subpixel=1 | tail -3
JN_SGM_NS=3 timeout 200 python scripts/sgm_debug.py 200 150 128 2 | tail -3
