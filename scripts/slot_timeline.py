#!/usr/bin/env python3
"""A slot's life in the pipelined path (rocprofv3 --kernel-trace of a bench run):
    python3 scripts/slot_timeline.py <kernel_trace.csv>
Per stream (= slot): batches are cut at k_sobel_planes; for every kernel of the chain the mean duration inside the pipeline, the mean gap
in front of it (end of the previous kernel of the same stream -> its start), and the batch's cycle (k_sobel_planes to k_sobel_planes).
What it answers: where the 4 x step milliseconds of a slot's cycle go — kernels stretched by the other slots, or waits between them."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
per = defaultdict(list)
for r in rows:
    name = re.sub(r"^void |jnav::|\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
    if "at::native" in name or name.startswith("__amd"):
        continue
    per[r.get("Stream_Id") or r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
dur = defaultdict(list); gap = defaultdict(list); cycles = []; busy = []
for sid, ev in per.items():
    ev.sort()
    heads = [i for i, e in enumerate(ev) if e[2].startswith("k_sobel_planes")]
    if len(heads) < 8:
        continue
    for a, b in zip(heads[2:-2], heads[3:-1]):                       # steady state: drop the first and last batches of the stream
        cyc = ev[b][0] - ev[a][0]
        if cyc > 30e6:                                               # a synchronisation between timed regions
            continue
        cycles.append(cyc / 1e6)
        busy.append(sum(e[1] - e[0] for e in ev[a:b]) / 1e6)
        for i in range(a, b):
            dur[ev[i][2]].append((ev[i][1] - ev[i][0]) / 1e3)
            if i > a:
                gap[ev[i][2]].append((ev[i][0] - ev[i - 1][1]) / 1e3)
        gap["(end of chain -> next k_sobel_planes)"].append((ev[b][0] - ev[b - 1][1]) / 1e3)
if not cycles:
    sys.exit("no steady-state batches found")
m = lambda v: sum(v) / max(len(v), 1)
print("streams %d, batches %d: cycle %.3f ms, kernels of the chain %.3f ms, waits %.3f ms" % (len(per), len(cycles), m(cycles), m(busy), m(cycles) - m(busy)))
order = sorted(dur, key=lambda k: -sum(dur[k]) / len(cycles))
print("%-44s %8s %9s %9s" % ("kernel", "per batch", "us each", "gap before us"))
for k in order:
    print("%-44s %8.2f %9.1f %9.1f" % (k[:44], len(dur[k]) / len(cycles), m(dur[k]), m(gap.get(k, [0]))))
k = "(end of chain -> next k_sobel_planes)"
print("%-44s %8s %9s %9.1f" % (k, "", "", m(gap[k])))
