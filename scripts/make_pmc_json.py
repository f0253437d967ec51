#!/usr/bin/env python3
"""Build profiles/<round>_pmc_traffic.json from the PMC passes of scripts/collect_profiles.sh + scripts/pmc_sq.sh:
  python3 scripts/make_pmc_json.py <tag> <out.json>     (reads gpurun_out/<tag>_pmc_{FETCH,WRITE}_SIZE.txt, <tag>_sq{1,2}.txt,
                                                          <tag>_slots1_summary.txt)
FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B for 16-byte-per-lane streaming loads, MI355X_MICROARCH.md,
HBM section); WRITE_SIZE is taken as is.  The file records the sha256 of kernels.hip it was measured with: bench.py
reports `traffic` only when that still matches."""
import hashlib, json, os, re, subprocess, sys
tag, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = os.path.join(root, "gpurun_out")


def counters(path):
    res, cur = {}, None
    if not os.path.exists(path):
        return res
    for line in open(path):
        if not line.startswith(" "):
            cur = re.sub(r"^void ", "", line.strip()); res[cur] = {}
        else:
            f = line.split()
            res[cur][f[0]] = float(f[1])
    return res


fetch, write = counters(os.path.join(g, tag + "_pmc_FETCH_SIZE.txt")), counters(os.path.join(g, tag + "_pmc_WRITE_SIZE.txt"))
sq = {}
for i in (1, 2):
    for k, v in counters(os.path.join(g, tag + "_sq%d.txt" % i)).items():
        sq.setdefault(k, {}).update(v)
alone = {}
p = os.path.join(g, tag + "_slots1_summary.txt")
if os.path.exists(p):
    for line in open(p):
        m = re.match(r"(?:void )?(\S+(?:<[^>]*>)?)\s+calls\s+\d+ avg\s+([0-9.]+) us", line)
        if m:
            alone[m.group(1)] = float(m.group(2)) / 1e3
doc = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (two passes, scripts/collect_profiles.sh) and SQ passes (scripts/pmc_sq.sh) -- "
               "python3 bench.py --steps 3 --warmup 1 --slots 1 --no-cpu-baseline --no-latency-config",
    "workload": "32 pairs 1280x720 disp_max=127 per launch",
    "correction": "gfx950: FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM section) — settled by a probe for the loads these kernels issue: a buffer streamed once with raw_buffer_load of 4, 8 and 16 bytes per lane "
                  "shows HALF the bytes read in FETCH_SIZE in all three cases (scripts/probes/fetch_size_probe.hip, profiles/%s_fetch_size_probe.txt); WRITE_SIZE as is; KB = 1024 B" % tag,
    "kernels_hip_sha256": hashlib.sha256(open(os.path.join(root, "jackal_navigation_amd", "csrc", "kernels.hip"), "rb").read()).hexdigest(),
    "commit": subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
}
for name in sorted(set(fetch) | set(write)):
    key = re.sub(r"<.*", "", name)
    e = {"kernel": name, "FETCH_SIZE_KB": fetch.get(name, {}).get("FETCH_SIZE"), "WRITE_SIZE_KB": write.get(name, {}).get("WRITE_SIZE")}
    if e["FETCH_SIZE_KB"] is not None and e["WRITE_SIZE_KB"] is not None:
        e["traffic_bytes"] = int(2 * e["FETCH_SIZE_KB"] * 1024 + e["WRITE_SIZE_KB"] * 1024)
    if name in alone:
        e["alone_ms_per_launch"] = round(alone[name], 4)
    for k, v in sq.get(name, {}).items():
        e[k] = int(v)
    if "SQ_ACTIVE_INST_VALU" in e and name in alone:
        # SQ_ACTIVE_INST_* count quad-cycles summed over waves; 256 CUs x 4 SIMDs at 2.4 GHz
        e["valu_issue_frac_alone"] = round(e["SQ_ACTIVE_INST_VALU"] * 4 / (alone[name] * 1e-3 * 1024 * 2.4e9), 3)
    if key.startswith("k_dense"):
        e["algorithmic_bytes"] = 16 * 1280 * 720 * 32
        doc["k_dense"] = e
    else:
        doc[key] = e
# the whole path per batch: every kernel that runs once per batch (k_valid_lut runs once per handle: left out)
per_batch = {k: v for k, v in doc.items() if isinstance(v, dict) and "traffic_bytes" in v and k != "k_valid_lut"}
if len(per_batch) >= 10:
    total = sum(v["traffic_bytes"] for v in per_batch.values())
    alg = int(97 * 1280 * 720 * 32)
    doc["whole_path"] = {"kernels": sorted(v["kernel"] for v in per_batch.values()), "traffic_bytes_per_batch": total, "algorithmic_bytes_per_batch": alg,
                         "traffic_ratio": round(total / alg, 3),
                         "alone_ms_sum": round(sum(v.get("alone_ms_per_launch", 0.0) for v in per_batch.values()), 4),
                         "note": "FETCH_SIZE x 2 + WRITE_SIZE summed over the kernels of one batch (32 pairs 1280x720), one slot; 97 B per pixel and pair is SURVEY 8d's algorithmic figure"}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc.get("k_dense"), indent=1))
