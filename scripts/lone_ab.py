"""Median latency of synchronous jn_elas_process_batch calls on one lone pair (device pointers):  python3 scripts/lone_ab.py [W H D reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jackal_navigation_amd as jn
from jackal_navigation_amd import node
from jackal_navigation_amd.device import DeviceArray
W, H, D, reps = (int(x) for x in (sys.argv[1:5] + ["640", "480", "64", "1500"][len(sys.argv) - 1:]))
l, r = node.synth_pair(W, H, D, 12345)
dl, dr = DeviceArray.from_numpy(l), DeviceArray.from_numpy(r)
d1, d2 = DeviceArray((H, W), np.float32), DeviceArray((H, W), np.float32)
with jn.Elas(jn.Elas.parameters(jn.Elas.ROBOTICS, disp_max=D - 1), W, H, max_batch=1, host_threads=8, slots=1) as e:
    for _ in range(100):
        e.process_batch(1, dl.ptr, dr.ptr, W, H * W, d1.ptr, d2.ptr)
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); e.process_batch(1, dl.ptr, dr.ptr, W, H * W, d1.ptr, d2.ptr); t.append(time.perf_counter() - t0)
t.sort()
print("%dx%d D=%d: median %.3f ms  p10 %.3f  p90 %.3f  (inline=%s spin=%s)" % (W, H, D, t[len(t) // 2] * 1e3, t[len(t) // 10] * 1e3, t[9 * len(t) // 10] * 1e3,
      os.environ.get("JN_INLINE_SYNC", "default"), os.environ.get("JN_POOL_SPIN_US", "default")))
