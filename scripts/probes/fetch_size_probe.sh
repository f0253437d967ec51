#!/bin/bash
# FETCH_SIZE against known bytes for 4 / 8 / 16-byte-per-lane buffer loads (scripts/probes/fetch_size_probe.hip).  Inside gpurun:
#   bash scripts/probes/fetch_size_probe.sh > gpurun_out/fetch_size_probe.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/scripts/probes/fetch_size_probe.hip -o /tmp/fetch_size_probe || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fsp; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fsp -- /tmp/fetch_size_probe > /tmp/fsp.log 2>&1
python3 - <<'PY'
import csv, glob, re
f = sorted(glob.glob("/tmp/fsp/*/*counter_collection.csv"))[-1]
acc = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    m = re.search(r"k_stream<(\d+)>", r["Kernel_Name"])
    if m:
        acc.setdefault(int(m.group(1)), []).append(float(r["Counter_Value"]))
total = float(1 << 30)
print("1 GiB streamed once per launch, fully coalesced; FETCH_SIZE in KB (1024 B) per launch, and known bytes / counted bytes = the factor the counter needs")
for b in sorted(acc):
    v = acc[b][-1]                                  # the second repetition (the first one warms nothing: the buffer is 4x the MALL anyway)
    print("raw_buffer_load of %2d bytes per lane: FETCH_SIZE %.0f KB = %.3f GB counted for 1.074 GB read -> factor %.3f" % (b, v, v * 1024 / 1e9, total / (v * 1024)))
PY
