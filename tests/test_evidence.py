"""The evidence chain: what README.md and bench.py quote must be what profiles/ holds.  CPU only."""
import csv
import glob
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def current():
    return open(os.path.join(P, "CURRENT")).read().strip()


def test_manifest_lists_existing_files_and_was_taken_on_this_source():
    tag = current()
    man = json.load(open(os.path.join(P, "%s_manifest.json" % tag)))
    assert man["tag"] == tag and len(man["files"]) >= 20
    for name in man["files"]:
        assert os.path.getsize(os.path.join(P, name)) > 0, name
    # kernel sources: the profiles describe THIS tree's kernels (re-publish after touching them: scripts/round6_profiles.sh + publish_round.py)
    for src in ("jackal_navigation_amd/csrc/kernels.hip", "jackal_navigation_amd/csrc/sgm_sweep.hip", "jackal_navigation_amd/csrc/bm.hip", "jackal_navigation_amd/csrc/bm_mfma.hip", "jackal_navigation_amd/csrc/prefilter.h", "jackal_navigation_amd/csrc/delaunay_gpu.hip"):
        sha = hashlib.sha256(open(os.path.join(ROOT, src), "rb").read()).hexdigest()
        assert man["sources_sha256"][src] == sha, "%s changed after profiles/%s_* were taken" % (src, tag)


def test_every_kernel_with_pmc_traffic_is_in_the_kernel_stats():
    tag = current()
    names = {re.sub(r"jnav::|jnav_sgm::|\(anonymous namespace\)::|^void ", "", r["Name"]).split("(")[0] for r in csv.DictReader(open(os.path.join(P, "%s_slots1_kernel_stats.csv" % tag)))}
    pmc = json.load(open(os.path.join(P, "%s_pmc_traffic.json" % tag.split("_")[0])))
    listed = [v["kernel"] for v in pmc.values() if isinstance(v, dict) and "kernel" in v]
    assert len(listed) >= 15                                          # the whole path, not only the heavy kernels
    wp = pmc["whole_path"]
    assert wp["traffic_bytes_per_batch"] == sum(v["traffic_bytes"] for k, v in pmc.items() if isinstance(v, dict) and "traffic_bytes" in v and k != "k_valid_lut")
    assert abs(wp["traffic_ratio"] - wp["traffic_bytes_per_batch"] / (97 * 1280 * 720 * 32)) < 1e-3
    for k in listed:
        assert re.sub(r"^void ", "", k) in names, (k, sorted(names)[:8])
    assert pmc["kernels_hip_sha256"] == hashlib.sha256(open(os.path.join(ROOT, "jackal_navigation_amd", "csrc", "kernels.hip"), "rb").read()).hexdigest()
    sgm = json.load(open(os.path.join(P, "%s_sgm_pmc_traffic.json" % tag.split("_")[0])))
    snames = {re.sub(r"^void ", "", r["Name"]).split("(")[0] for r in csv.DictReader(open(os.path.join(P, "%s_sgm_kernel_stats.csv" % tag)))}
    for k in sgm["kernels"]:
        assert k in snames, (k, sorted(snames)[:8])
    assert sgm["bytes_per_batch"] == sum(v["traffic_bytes"] for v in sgm["kernels"].values())


def test_roofline_is_recomputable_from_the_tracked_files():
    """frac of the bench line = algorithmic bytes / k_dense_row's alone time / 8 TB/s, and the alone time the line carries agrees with
    the one-slot kernel stats within 8 % (different runs of the same tree on the same box, one of them under rocprofv3)."""
    tag = current()
    b = json.loads([l for l in open(os.path.join(P, "%s_bench_line.json" % tag)) if l.startswith('{"metric"')][-1])
    r = b["roofline"]
    assert abs(r["frac"] - r["algorithmic_bytes_per_launch"] / (r["ms_per_launch"] * 1e-3) / 8e12) < 2e-3
    assert r["algorithmic_bytes_per_launch"] == 16 * 1280 * 720 * 32
    rows = {r_["Name"]: float(r_["AverageNs"]) for r_ in csv.DictReader(open(os.path.join(P, "%s_slots1_kernel_stats.csv" % tag)))}
    alone = [v for k, v in rows.items() if "k_dense_row" in k][0] / 1e6
    assert abs(alone - r["ms_per_launch"]) / alone < 0.08, (alone, r["ms_per_launch"])   # two runs (one under the tracer) of one tree on one box: 2-6 % apart over the rounds' sets
    assert b["check"]["ok"] is True
    assert abs(b["value"] - 32 / (b["ms_per_step"] * 1e-3)) / b["value"] < 1e-3          # value = pairs of a step / time of a step
    under = json.loads(open(os.path.join(P, "%s_default_bench_line_under_rocprof.json" % tag)).read())
    pipelined = [v for k, v in {r_["Name"]: float(r_["AverageNs"]) for r_ in csv.DictReader(open(os.path.join(P, "%s_default_bench_kernel_stats.csv" % tag)))}.items() if "k_dense_row" in k][0] / 1e6
    assert abs(pipelined - under["roofline"]["ms_per_launch_pipelined"]) / pipelined < 0.05, (pipelined, under["roofline"]["ms_per_launch_pipelined"])


def test_readme_quotes_the_committed_numbers():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "readme_numbers.py")], capture_output=True, text=True, check=True).stdout.strip()
    assert out in open(os.path.join(ROOT, "README.md")).read(), "README.md's numbers block is stale: python3 scripts/readme_numbers.py --write"


def test_integration_md_quotes_the_committed_rates():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "readme_numbers.py"), "--rates"], capture_output=True, text=True, check=True).stdout.strip()
    assert out in open(os.path.join(ROOT, "INTEGRATION.md")).read(), "INTEGRATION.md's rates block is stale: python3 scripts/readme_numbers.py --write"


def test_design_names_the_current_evidence_set():
    tag = current()
    txt = open(os.path.join(ROOT, "DESIGN.md")).read()
    cited = set(re.findall(r"profiles/(r\d\d_(?:[a-z]_)?[A-Za-z0-9_]+\.(?:csv|json|txt|jsonl))", txt)) | set(re.findall(r"`(r\d\d_(?:[a-z]_)?[A-Za-z0-9_]+\.(?:csv|json|txt|jsonl))`", txt))
    assert cited, "DESIGN.md cites no evidence file"
    for name in cited:
        assert os.path.exists(os.path.join(P, name)), "DESIGN.md cites profiles/%s, which is not tracked" % name
        assert name.startswith(tag + "_"), "DESIGN.md cites %s but profiles/CURRENT is %s" % (name, tag)
