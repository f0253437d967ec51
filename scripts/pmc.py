#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: mean counter value per kernel (per dispatch)."""
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"jnav::|\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in acc.items():
    if pat and not re.search(pat, k):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-24s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
