"""GPU occupancy of a pipelined run from a rocprofv3 kernel trace: the fraction of wall time during which at
least one kernel ran, how many ran at once on average, and which kernels were in flight alone.
Usage: python3 scripts/busy.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in rows)
t_lo = ev[0][0] + skip * (ev[-1][1] - ev[0][0])          # drop start-up / warm-up
ev = [e for e in ev if e[0] >= t_lo]
points = []
for s, e, n in ev:
    points.append((s, 1, n)); points.append((e, -1, n))
points.sort()
active = defaultdict(int)
depth = 0; last = points[0][0]; busy = 0; weighted = 0; alone = defaultdict(int)
for t, d, n in points:
    dt = t - last
    if depth > 0:
        busy += dt; weighted += dt * depth
        if depth == 1:
            alone[[k for k, v in active.items() if v > 0][0]] += dt
    depth += d; active[n] += d; last = t
wall = points[-1][0] - points[0][0]
print("wall %.2f ms  busy %.1f %%  mean kernels in flight while busy %.2f  sum of kernel time / wall %.2f" %
      (wall / 1e6, 100.0 * busy / wall, weighted / max(busy, 1), sum(e - s for s, e, _ in ev) / wall))
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:8]:
    print("  alone on the GPU: %-42s %5.1f %% of wall" % (k, 100.0 * v / wall))
