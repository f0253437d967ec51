export JN_STEREO_LIB=${JN_STEREO_LIB:-${GRAFT_REPO_ROOT:-$(pwd)}/jackal_navigation_amd/libjn_stereo_hooks.so}   # the switches used below exist in the hooks build only (csrc/hooks.h)
for b in 128 90 64 48; do
JN_POST_BAND=$b bash scripts/prof.sh r2p_$b > gpurun_out/r2p_$b.txt 2>&1; echo band $b $(grep "k_gap_mean_fused\|total GPU" gpurun_out/r2p_$b.txt | tr '\n' ' ')
done
