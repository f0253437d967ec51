export JN_STEREO_LIB=${JN_STEREO_LIB:-${GRAFT_REPO_ROOT:-$(pwd)}/jackal_navigation_amd/libjn_stereo_hooks.so}   # the switches used below exist in the hooks build only (csrc/hooks.h)
for d in 0 1 2 3 4; do
cd /tmp; export TMPDIR=/tmp
JN_DENSE2=${JN_DENSE2:-1} JN_DENSE_DBG=$d timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dbg$d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --slots 1 --no-cpu-baseline --no-latency-config --no-alone-leg --min-time 0 > /dev/null 2>&1
echo DBG $d; python3 $GRAFT_REPO_ROOT/scripts/pmc.py $(ls $GRAFT_REPO_ROOT/gpurun_out/dbg$d/*/*counter_collection.csv | head -1) "k_dense" | grep -v "^k_"
done
