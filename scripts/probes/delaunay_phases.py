"""Support points of the 720p benchmark frame (via the checker), then the phase timing of csrc/delaunay.cpp on them (CPU only)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle.binding import Oracle
o = Oracle()
L, R = o.synth_pair(1280, 720, 128, 12345)
p = o.params(0, disp_max=127)
sup = np.asarray(o.support(p, o.descriptor(L), o.descriptor(R)))
for side in (0, 1):
    pts = sup.copy()
    if side:
        pts[:, 0] -= pts[:, 2]
    open("/tmp/sup%d.txt" % side, "w").write("\n".join("%d %d %d" % tuple(r) for r in pts))
csrc = os.path.join(ROOT, "jackal_navigation_amd", "csrc")
exe = "/tmp/delaunay_phases"
subprocess.check_call(["g++", "-O3", "-mavx2", "-std=c++17", "-I", csrc, os.path.join(ROOT, "scripts", "probes", "delaunay_phases.cpp"),
                       os.path.join(csrc, "delaunay.cpp"), "-o", exe])
for side in (0, 1):
    print("side", side, subprocess.check_output([exe, "/tmp/sup%d.txt" % side]).decode().strip().replace("\n", " | "))
