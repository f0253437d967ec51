"""Idle gaps of the GPU inside the timed regions of a short-region bench run (rocprofv3 kernel trace):
python3 scripts/region_gaps.py <kernel_trace.csv> [min gap us, default 15]     Regions are separated by the synchronisations (gaps > 300 us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("jnav::", "")) for r in rows)
# merge into busy intervals
busy = []
for s, e, n in ev:
    if busy and s <= busy[-1][1]:
        busy[-1][1] = max(busy[-1][1], e)
    else:
        busy.append([s, e])
# regions: split at gaps > 300 us
regions = [[busy[0]]]
for b in busy[1:]:
    if b[0] - regions[-1][-1][1] > 300e3:
        regions.append([b])
    else:
        regions[-1].append(b)
regions = [r for r in regions if (r[-1][1] - r[0][0]) > 20e6]          # the 20-step regions are > 20 ms
print("%d regions of more than 20 ms" % len(regions))
for r in regions[2:6]:
    t0, t1 = r[0][0], r[-1][1]
    idle = [(r[i][1] - t0, r[i + 1][0] - r[i][1]) for i in range(len(r) - 1) if r[i + 1][0] - r[i][1] > thr * 1e3]
    tot = sum(g for _, g in idle)
    print("region %.2f ms, idle in gaps > %.0f us: %.2f ms;  first 2 ms: %.2f, last 2 ms: %.2f, middle: %.2f" % ((t1 - t0) / 1e6, thr, tot / 1e6,
          sum(g for a, g in idle if a < 2e6) / 1e6, sum(g for a, g in idle if a > (t1 - t0) - 2e6) / 1e6, sum(g for a, g in idle if 2e6 <= a <= (t1 - t0) - 2e6) / 1e6))
    print("   largest gaps (at ms, us):", ", ".join("%.2f:%.0f" % (a / 1e6, g / 1e3) for a, g in sorted(idle, key=lambda x: -x[1])[:10]))
