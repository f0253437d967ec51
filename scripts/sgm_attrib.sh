#!/bin/bash
# Attribution of the SGM mode's batch time (pipelined): what each part is worth when it is switched off (results are then WRONG).
# Needs libjn_stereo_prof.so: the hooks build with the sweeps' profiling branches (make -C jackal_navigation_amd/csrc hooks EXTRA=-DJN_SGM_PROFILE, then copy ../libjn_stereo_hooks.so aside under that name).  Usage: gpurun -- bash scripts/sgm_attrib.sh
R=$(pwd); out=gpurun_out
run() { python3 bench.py --mode sgm --sgm-slots ${SS:-6} --steps 18 --warmup 6 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], "pairs/s,", j["ms_per_step"], "ms per batch")'; }
{
echo "product library, 6 slots: $(run)"
echo "product library, 4 slots: $(SS=4 run)"
export JN_STEREO_LIB=$R/jackal_navigation_amd/libjn_stereo_prof.so
echo "profile library, nothing off: $(run)"
echo "no prefilter (EXP=1): $(JN_SGM_EXP=1 run)"
echo "no L/R kernel (EXP=2): $(JN_SGM_EXP=2 run)"
echo "no u8 + scan tail (EXP=4): $(JN_SGM_EXP=4 run)"
echo "no minima memset (EXP=8): $(JN_SGM_EXP=8 run)"
echo "no prefilter, L/R, tail, memset (EXP=15): $(JN_SGM_EXP=15 run)"
echo "no right-image minima in the final sweep (DBG=2): $(JN_SGM_DBG=2 run)"
echo "no producer-block columns (DBG=1): $(JN_SGM_DBG=1 run)"
echo "no volume loads / stores in the row sweeps (DBG=4): $(JN_SGM_DBG=4 run)"
echo "no waiting on neighbour strips (DBG=32): $(JN_SGM_DBG=32 run)"
echo "all of DBG 1+2+32: $(JN_SGM_DBG=35 run)"
} > $out/sgm_attrib.txt 2>&1
cat $out/sgm_attrib.txt
