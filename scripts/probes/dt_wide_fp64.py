"""Are the FP64 predicates of the GPU triangulation still right beyond (-2048, 2048)?  ("check it, do not assume it".)
Random lattice subsets with columns up to 8 000, each triangulated twice on the device — with the integer predicates the release library takes
for such coordinates, and with the FP64 ones forced (JN_DT_FP64, hooks build) — against the host's int64 replay.
    JN_STEREO_LIB=.../libjn_stereo_hooks.so python3 scripts/probes/dt_wide_fp64.py"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) < 2:                                   # the switch is read per launch from the environment: one child per setting
    env = dict(os.environ, JN_STEREO_LIB=os.path.join(ROOT, "jackal_navigation_amd", "libjn_stereo_hooks.so"))
    for fp in ("0", "1"):
        e = dict(env); e.pop("JN_DT_FP64", None)
        if fp == "1":
            e["JN_DT_FP64"] = "1"
        print(subprocess.run([sys.executable, __file__, fp], env=e, capture_output=True, text=True).stdout.strip())
    sys.exit(0)
import jackal_navigation_amd as jn
L = jn.load()
rng = np.random.default_rng(101)
sides = wrong = 0
worst = None
for rep in range(60):
    n = int(rng.integers(200, 3800)); cw = int(rng.choice([900, 1300, 1600])); ch = int(rng.choice([40, 90, 200]))
    cells = np.sort(rng.choice(cw * ch, size=n, replace=False))
    t = np.stack([cells // ch, cells % ch, rng.integers(0, 256, n)], axis=1).astype(np.int16)
    tl = np.zeros(6 * n, np.int32); tr = np.zeros(6 * n, np.int32); ntri = (C.c_int32 * 2)(); need = C.c_int32()
    assert L.jn_device_triangulate(0, t.ctypes.data, n, 5, tl.ctypes.data, tr.ctypes.data, ntri, C.byref(need)) == 0
    for side, (k, tri) in enumerate(((ntri[0], tl), (ntri[1], tr))):
        if need.value & (1 << side):
            continue
        x = np.ascontiguousarray(t[:, 0].astype(np.int32) * 5 - (t[:, 2].astype(np.int32) if side else 0)); y = np.ascontiguousarray(t[:, 1].astype(np.int32) * 5)
        te = np.zeros(6 * n, np.int32)
        ke = L.jn_host_triangulate(x.ctypes.data, y.ctypes.data, n, te.ctypes.data)
        sides += 1
        if k != ke or not np.array_equal(tri[:3 * k], te[:3 * ke]):
            wrong += 1
            worst = worst or (n, cw, ch, side)
print("%s predicates: %d sides of 200-3800 vertices, columns up to 8000: %d differ from the host's int64 replay%s"
      % ("FP64 (forced)" if sys.argv[1] == "1" else "integer (release)", sides, wrong, "" if not wrong else "  (first: n, cw, ch, side = %s)" % (worst,)))
