#!/usr/bin/env python3
"""Pretty-print a rocprofv3 *_kernel_stats.csv (per-kernel calls / average / share)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU time %.3f ms" % (tot / 1e6))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 28]:
    name = re.sub(r"jnav::|\(anonymous namespace\)::", "", r["Name"]).split("(")[0]
    print("%-46s calls %4s avg %9.1f us %5.1f%%" % (name[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
