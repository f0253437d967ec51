"""Generates tests/golden/bench_modes_golden.json: FNV-1a-64 of the SGM / block-matching (SAD, SSD) ORACLES' int16 maps for every pair of
bench.py's headline batch (1280x720, scene 128, D = 128, seeds 12345 .. 12376), so that bench.py's `other_modes` legs can check every frame of
every slot as the ELAS leg does.  SELF-REFERENTIAL goldens (the reference has no such matchers): they pin the HIP path to oracle/sgm_oracle.cpp
and oracle/bm_oracle.cpp, nothing more.  ~10 minutes on 8 cores.  Run from the repo root:  python tests/golden/make_bench_modes_golden.py"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
W, H, SCENE, D, B = 1280, 720, 128, 128, 32


def one(seed):
    from oracle.binding import Oracle, SgmOracle, BmOracle
    o, s, b = Oracle(), SgmOracle(), BmOracle()
    L, R = o.synth_pair(W, H, SCENE, seed)
    out = {}
    out["sgm"] = "%016x" % o.fnv(s.process(s.params(D), L, R).view(np.uint32))
    out["bm"] = "%016x" % o.fnv(b.process(b.params(D, 4), L, R).view(np.uint32))
    out["bm_ssd"] = "%016x" % o.fnv(b.process(b.params(D, 4, cost_function=1), L, R).view(np.uint32))
    return seed, out


if __name__ == "__main__":
    with mp.Pool(int(os.environ.get("JN_GOLDEN_PROCS", "8"))) as pool:
        res = dict(pool.map(one, range(12345, 12345 + B)))
    doc = {"what": "oracle hashes (FNV-1a-64 over the int16 map as uint32 words) per generator seed: sgm = oracle/sgm_oracle.cpp defaults, D = 128; "
                   "bm / bm_ssd = oracle/bm_oracle.cpp 9x9 SAD / SSD, D = 128, no sub-pixel",
           "W": W, "H": H, "scene_disp": SCENE, "D": D, "frames": {str(k): v for k, v in sorted(res.items())}}
    json.dump(doc, open(os.path.join(ROOT, "tests", "golden", "bench_modes_golden.json"), "w"), indent=0)
    print("wrote %d frames" % len(res))
