"""Where k_delaunay's time goes: microseconds per tree level for one 720p frame's support points (JN_DT_CLOCKS makes jn_device_triangulate print them)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["JN_DT_CLOCKS"] = "1"
os.environ.setdefault("JN_STEREO_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jackal_navigation_amd", "libjn_stereo_hooks.so"))   # JN_DT_CLOCKS: hooks build only
import jackal_navigation_amd as jn
from oracle.binding import Oracle
o = Oracle(); L = jn.load()
for (W, H, sd, dmax) in ((1280, 720, 128, 127), (640, 480, 64, 63)):
    Lm, Rm = o.synth_pair(W, H, sd, 12345)
    sup = np.asarray(o.support(o.params(0, disp_max=dmax), o.descriptor(Lm), o.descriptor(Rm)))
    t = np.ascontiguousarray(np.stack([sup[:, 0] // 5, sup[:, 1] // 5, sup[:, 2]], axis=1).astype(np.int16))
    n = len(t)
    tl = np.zeros(6 * n, np.int32); tr = np.zeros(6 * n, np.int32); ntri = (C.c_int32 * 2)(); need = C.c_int32()
    for rep in range(2):
        L.jn_device_triangulate(0, t.ctypes.data, n, 5, tl.ctypes.data, tr.ctypes.data, ntri, C.byref(need))
