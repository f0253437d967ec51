"""Quick end-to-end parity check of the HIP path against the CPU oracle (run on the GPU box)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jackal_navigation_amd as jn
from jackal_navigation_amd.device import DeviceArray
from oracle.binding import Oracle

o = Oracle()
cfgs = [(320, 180, 48, 256), (640, 480, 64, 64), (1280, 720, 128, 128)]
if len(sys.argv) > 1:
    cfgs = cfgs[:int(sys.argv[1])]
for (W, H, sd, pd) in cfgs:
    L, R = jn.node.synth_pair(W, H, sd)
    p = jn.Elas.parameters(0, disp_max=pd - 1)
    po = o.params(disp_max=pd - 1)
    t = time.time(); st, D1o, D2o = o.process(po, L, R); t_or = time.time() - t
    with jn.Elas(p, W, H) as e:
        D1 = np.zeros((H, W), np.float32); D2 = np.zeros((H, W), np.float32)
        e.process(L, R, D1, D2)
        t = time.time(); st2 = e.process(L, R, D1, D2); t_gpu = time.time() - t
        times = e.last_times()
    ok1 = np.array_equal(D1.view(np.uint32), D1o.view(np.uint32)); ok2 = np.array_equal(D2.view(np.uint32), D2o.view(np.uint32))
    print("%dx%d D=%d: status %d/%d  D1 %s D2 %s  hash %016x  oracle %.1f ms  gpu(e2e, host ptrs) %.1f ms" %
          (W, H, pd, st, st2, "OK" if ok1 else "MISMATCH", "OK" if ok2 else "MISMATCH", o.fnv(D1), t_or * 1e3, t_gpu * 1e3))
    print("   stage ms:", {k: round(v, 3) for k, v in times.items()})
    for name, a, b in (("D1", D1, D1o), ("D2", D2, D2o)):
        if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
            idx = np.argwhere(a != b)
            print("   %s: %d differing px, first %s gpu=%s oracle=%s" % (name, len(idx), idx[:6].tolist(),
                  [float(a[tuple(i)]) for i in idx[:6]], [float(b[tuple(i)]) for i in idx[:6]]))
    # node side
    sp = jn.node.scan_params(W, H); spo = o.scan_params(W, H)
    lut = jn.node.build_valid_disp_lut(sp, W, H)
    luto = o.valid_lut(spo, W, H)
    print("   lut", "OK" if np.array_equal(lut.numpy(), luto) else "MISMATCH %d" % (lut.numpy() != luto).sum())
    dD = DeviceArray.from_numpy(D1o); du8 = DeviceArray((H, W), np.uint8)
    bins = DeviceArray((1, sp.bins), np.float64); meta = DeviceArray((1, 4), np.float64)
    jn.node.disparity_scan(sp, 1, dD.ptr, lut.ptr, W, H, du8.ptr, bins.ptr, meta.ptr)
    u8o = o.to_u8(D1o); binso, metao, used = o.scan(spo, u8o, luto)
    b = bins.numpy()[0]; m = meta.numpy()[0]
    print("   u8", "OK" if np.array_equal(du8.numpy(), u8o) else "MISMATCH", " bins maxabs %.3g" % np.abs(b - binso).max(),
          " meta maxabs %.3g" % np.abs(m - metao).max(), " used", used, " nonempty", int((binso < 1e9 - 1).sum()))
    pc = jn.node.point_cloud(sp, du8.ptr, W, H); pco = o.point_cloud(spo, u8o)
    print("   point cloud", pc.shape, pco.shape, "maxabs %.3g" % (np.abs(pc - pco).max() if pc.shape == pco.shape and len(pc) else -1))
