#!/bin/bash
# k_gap_mean_fused: rows per band (JN_POST_BAND) against the kernel's average duration under rocprofv3 (inside gpurun)
export JN_STEREO_LIB=${JN_STEREO_LIB:-${GRAFT_REPO_ROOT:-$(pwd)}/jackal_navigation_amd/libjn_stereo_hooks.so}   # the switches used below exist in the hooks build only (csrc/hooks.h)
cd /tmp; export TMPDIR=/tmp
cat > /tmp/pb_avg.py <<'EOF'
import csv, glob, sys
for f in glob.glob("/tmp/pb/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_gap_mean_fused" in r["Name"]:
            print("%s calls, %.1f us average" % (r["Calls"], float(r["AverageNs"]) / 1e3))
EOF
for b in ${1:-40 48 60 72 80 90 120}; do
  rm -rf /tmp/pb; JN_POST_BAND=$b rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --slots 1 --no-cpu-baseline --no-latency-config --no-alone-leg --min-time 0 > /tmp/pb.log 2>&1
  echo "JN_POST_BAND=$b: $(python3 /tmp/pb_avg.py)"
done
