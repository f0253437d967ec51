"""Timeline of ONE lone pair from a rocprofv3 kernel trace of scripts/latency_check.py: per kernel its start relative to
the pair's first kernel, its duration and the idle gap before it.
Usage: python3 scripts/lone_timeline.py <kernel_trace.csv> [first kernel name, default k_descriptor_fused] [which pair, default -3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_sobel_planes"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("jnav::", "")) for r in rows)
starts = [i for i, e in enumerate(ev) if first in e[2]]
a = starts[which]; b = starts[which + 1]
t0 = ev[a][0]; last_end = t0; ksum = 0
for s, e, n in ev[a:b]:
    print("%8.1f us  +%6.1f gap  %6.1f us  %s" % ((s - t0) / 1e3, (s - last_end) / 1e3, (e - s) / 1e3, n[:50]))
    last_end = max(last_end, e); ksum += e - s
print("pair: %.1f us first start -> last end, %.1f us in kernels, %d launches; next pair starts %.1f us after this one" %
      ((last_end - t0) / 1e3, ksum / 1e3, b - a, (ev[b][0] - t0) / 1e3))
