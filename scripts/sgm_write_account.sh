#!/bin/bash
# what are k_sw_w<up, final>'s 3.07 GB of writes per batch?  WRITE_SIZE / FETCH_SIZE of the SGM kernels with parts of the final
# sweep switched off (profile build: hooks + -DJN_SGM_PROFILE; results are WRONG under a switch, the counters are what matters).
#   JN_SGM_DBG: 2 = no right-image minima (LDS atomics + the per-row flush with global atomics), 1 = no producer-block columns,
#   JN_SGM_EXP: 8 = no minima memset
R=${GRAFT_REPO_ROOT:-$(pwd)}
export JN_STEREO_LIB=$R/jackal_navigation_amd/libjn_stereo_prof.so
cd /tmp && export TMPDIR=/tmp
for cfg in "0 0" "2 0" "1 0" "3 0"; do
  set -- $cfg
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/swa; JN_SGM_DBG=$1 JN_SGM_EXP=$2 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/swa -- python3 $R/bench.py --mode sgm --sgm-slots 1 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/swa.log 2>&1
    echo "JN_SGM_DBG=$1 JN_SGM_EXP=$2 $c:"; python3 $R/scripts/pmc.py $(ls /tmp/swa/*/*counter_collection.csv | tail -1) "k_sw_" | grep -A1 -E "^[a-z_:A-Z<>0-9, ]*k_sw" | grep -E "k_sw|$c" | paste - - | sed 's/  */ /g'
  done
done
