#!/usr/bin/env python3
"""The numbers README.md quotes, straight from the committed evidence set profiles/CURRENT names.
    python3 scripts/readme_numbers.py            prints the markdown block
    python3 scripts/readme_numbers.py --write    replaces the block between the markers in README.md
tests/test_evidence.py fails when README.md's block differs from what this prints."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
BEGIN, END = "<!-- numbers:begin (scripts/readme_numbers.py --write) -->", "<!-- numbers:end -->"
RBEGIN, REND = "<!-- rates:begin (scripts/readme_numbers.py --write) -->", "<!-- rates:end -->"


def rates_block():
    """INTEGRATION.md's host-pointer rates (PCIe inclusive), from the evidence set's host_pointer_rate.txt"""
    tag = open(os.path.join(P, "CURRENT")).read().strip()
    txt = open(os.path.join(P, "%s_host_pointer_rate.txt" % tag)).read()

    def rate(prefix):
        for l in txt.splitlines():
            if l.startswith(prefix):
                return float(re.search(r"[:=] *([0-9.]+) pairs/s", l).group(1))
        raise KeyError(prefix)
    sync720, sync480 = rate("1280x720 jn_elas_process"), rate("640x480 jn_elas_process")
    best720 = max(float(re.search(r": ([0-9.]+) pairs/s", l).group(1)) for l in txt.splitlines() if l.startswith("1280x720 jn_elas_submit_host"))
    best480 = max(float(re.search(r": ([0-9.]+) pairs/s", l).group(1)) for l in txt.splitlines() if l.startswith("640x480 jn_elas_submit_host"))
    return "\n".join([RBEGIN,
                      "Measured through host pointers, PCIe included (`profiles/%s_host_pointer_rate.txt`, `scripts/host_pointer_rate.py`): `jn_elas_process`, one synchronous call per pair, "
                      "%.2f k pairs/s at 1280x720 and %.2f k at 640x480; `jn_elas_submit_host` on 4 slots (best batch size of the sweep) %.1f k at 1280x720 and %.1f k at 640x480."
                      % (tag, sync720 / 1e3, sync480 / 1e3, best720 / 1e3, best480 / 1e3), REND])



def line(name):
    txt = open(os.path.join(P, name)).read()
    return json.loads([l for l in txt.splitlines() if l.startswith('{"metric"')][-1])


def gpu_dt_row(tag):
    try:
        for l in open(os.path.join(P, "%s_gpu_delaunay_ab.txt" % tag)):
            if l.startswith("all cores, JN_GPU_DELAUNAY=1:"):
                m = re.search(r": ([0-9.]+) pairs/s.*host cores busy ([0-9.]+)", l)
                return "%.1f k pairs/s, %.1f" % (float(m.group(1)) / 1e3, float(m.group(2)))
    except OSError:
        pass
    return "n/a"


def block():
    tag = open(os.path.join(P, "CURRENT")).read().strip()
    b = line("%s_bench_line.json" % tag)
    s = line("%s_sgm_bench_line.json" % tag)
    m = line("%s_bm_bench_line.json" % tag)
    other = [json.loads(l) for l in open(os.path.join(P, "%s_other_configs.jsonl" % tag)) if l.startswith("{")]

    def find(sub):
        for o in other:
            if sub in o["config"]["workload"]:
                return o
        return None
    node = open(os.path.join(P, "%s_node_rate.txt" % tag)).read().splitlines()
    node_ms = [re.search(r"scan: ([0-9.]+) ms per frame", l).group(1) for l in node if "ms per frame" in l]
    merge = [l for l in open(os.path.join(P, "%s_merge_in_worker.txt" % tag)).read().splitlines() if "cost" in l and not l.startswith("#")]
    cost = [float(l.split("cost")[1]) for l in merge]
    pace_file = os.path.join(P, "%s_pace_ab.txt" % tag)
    drv = [float(re.search(r": ([0-9.]+) pairs/s", l).group(1)) for l in open(pace_file) if l.startswith("JN_PACE=1, the driver")] if os.path.exists(pace_file) else []
    rows = [
        ("ELAS 1280x720, D=128, batch 32, one MI355X (`bench.py`, the headline)", "%.1f k pairs/s, %.3f ms per step" % (b["value"] / 1e3, b["ms_per_step"])),
        ("  the same with the driver's command (`--gpus 1 --steps 20 --warmup 5`: 20-step regions)", ("%.1f k pairs/s" % (sum(drv) / len(drv) / 1e3)) if drv else "n/a"),
        ("  roofline of `k_dense_row` alone / whole path (fraction of 8 TB/s)", "%.3f / %.3f" % (b["roofline"]["frac"], b["roofline"]["whole_path_frac"])),
        ("  HBM bytes counted over the whole path / SURVEY's 97 B per pixel and pair", ("%.2f GB per batch, ratio %.2f" % (b["roofline"]["whole_path_traffic"] / 1e9, b["roofline"]["whole_path_traffic_ratio"])) if b["roofline"].get("whole_path_traffic") else "n/a"),
        ("  the same pipeline with NO host stage (`JN_GPU_DELAUNAY=1`: what a rank pinned to <= 16 cores runs), busy host cores", gpu_dt_row(tag)),
        ("ELAS 640x480, D=64, batch 32, four batches in flight (the driver line's `vga_config`)", ("%.1f k pairs/s" % (b["vga_config"]["pairs_per_sec"] / 1e3)) if b.get("vga_config") and b["vga_config"].get("pairs_per_sec") else "n/a"),
        ("  reference CPU path on the same box (%d cores)" % b["cpu_baseline"]["cores"], "%.0f pairs/s" % b["cpu_baseline"]["value"]),
        ("ELAS 640x480, D=64, batch 64", "%.1f k pairs/s" % (find("640x480 rectified pairs (scene disparities <= 64), ELAS disp_max=63 (D=64), batch=64")["value"] / 1e3)),
        ("ELAS 320x180, disp_max 255, batch 128 (the reference's native size)", "%.0f k pairs/s" % (find("320x180")["value"] / 1e3)),
        ("ELAS 1920x1080, D=256, batch 8", "%.1f k pairs/s" % (find("1920x1080 rectified pairs (scene disparities <= 256), ELAS")["value"] / 1e3)),
        ("lone 640x480 pair, ELAS (median of 200 calls)", "%.3f ms" % b["latency_config"]["ms_per_frame"]),
        ("SGM 8 paths 1280x720, D=128, batch 32 (`--mode sgm`, six batches in flight)", "%.2f k pairs/s, frac %.3f on SURVEY's B_sgm" % (s["value"] / 1e3, s["roofline"]["frac"])),
        ("block matching 9x9 1280x720, D=128, batch 32 (`--mode bm`, SAD, `v_qsad`)", "%.1f k pairs/s" % (m["value"] / 1e3)),
        ("  the same with the squared-difference cost on the matrix cores (`--bm-cost ssd`, `v_mfma_i32_32x32x32_i8`)", "%.1f k pairs/s" % (line("%s_bm_ssd_bench_line.json" % tag)["value"] / 1e3)),
        ("  1920x1080, D=256, batch 8: SSD on the matrix cores / SAD", "%.2f k / %.2f k pairs/s" % (line("%s_bm_ssd_1080p_bench_line.json" % tag)["value"] / 1e3, line("%s_bm_sad_1080p_bench_line.json" % tag)["value"] / 1e3)),
        ("SGM 1920x1080, D=256 + 1/16 pixel, batch 8 (config 5's share of one GPU)", "%.0f pairs/s" % find("SGM 8 paths D=256")["value"]),
        ("node path, two 640x360 JPEG frames -> LaserScan, one frame at a time", "%s ms (eyes decoded serially: %s ms)" % (node_ms[-1], node_ms[0])),
        ("cross-rig merge in the slot worker, one-rank communicator", "%.1f-%.1f %% of the pipelined rate" % (100 * min(cost), 100 * max(cost))),
    ]
    out = [BEGIN, "Evidence set `profiles/%s_*` (one MI355X; every row is a committed file there):" % tag, "", "| What | Measured |", "|---|---|"]
    out += ["| %s | %s |" % r for r in rows]
    out.append(END)
    return "\n".join(out)


if __name__ == "__main__":
    blk = block()
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "README.md")
        txt = open(p).read()
        if BEGIN in txt:
            txt = txt[:txt.index(BEGIN)] + blk + txt[txt.index(END) + len(END):]
        else:
            txt += "\n" + blk + "\n"
        open(p, "w").write(txt)
        p = os.path.join(ROOT, "INTEGRATION.md")
        txt = open(p).read()
        txt = txt[:txt.index(RBEGIN)] + rates_block() + txt[txt.index(REND) + len(REND):]
        open(p, "w").write(txt)
    elif "--rates" in sys.argv:
        print(rates_block())
    else:
        print(blk)
