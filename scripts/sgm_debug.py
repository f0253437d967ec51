#!/usr/bin/env python3
"""GPU check of the sweep SGM kernels stage by stage against oracle/sgm_oracle.cpp (checker): the two horizontal volumes,
the downward three-path volume, the winners, the final map.  python scripts/sgm_debug.py [W H D n [key=value ...]]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jackal_navigation_amd as jn                       # noqa: E402
from jackal_navigation_amd import _lib                    # noqa: E402
from jackal_navigation_amd.device import DeviceArray     # noqa: E402
from oracle.binding import Oracle, SgmOracle             # noqa: E402


def d2h(ptr, shape, dtype):
    out = np.empty(shape, dtype)
    _lib.check(_lib.load().jn_memcpy_d2h(0, out.ctypes.data_as(C.c_void_p), ptr, out.nbytes), "d2h")
    return out


def cost_volume(gL, gR, D):
    H, W = gL.shape
    Cv = np.zeros((H, W, D), np.int32)
    xs = np.arange(W)
    for d in range(D):
        for i in (-1, 0, 1):
            xl = np.clip(xs + i, 0, W - 1); xr = np.clip(xs + i - d, 0, W - 1)
            Cv[:, :, d] += np.abs(gL[:, xl].astype(np.int32) - gR[:, xr].astype(np.int32))
    return Cv


def volume_order(D, wide):
    """d of every stored element of a pixel (sgm_sweep.hip VOLUME LAYOUT / REGISTER LAYOUT)."""
    NR, DPL = D // 8, D // 4
    out = np.zeros(D, np.int64)
    for e in range(D):
        if wide:
            c, q, w = e // 32, (e % 32) // 8, e % 8
            r, half = 4 * c + w // 2, w % 2
        else:
            c, q, b = e // 64, (e % 64) // 16, e % 16
            r, half = 8 * c + 2 * (b // 4) + ((b % 4) >> 1), b & 1
        out[e] = DPL * q + r + half * NR
    return out


def report(name, got, exp):
    bad = got != exp
    if not bad.any():
        print("  %-10s ok" % name)
        return True
    idx = np.argwhere(bad)
    print("  %-10s %d / %d differ; first %s got %s exp %s; rows %d..%d cols(k) %d..%d" % (
        name, bad.sum(), bad.size, idx[0].tolist(), got[tuple(idx[0])], exp[tuple(idx[0])], idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max()))
    return False


def one(W, H, D, n, kw, scene=None):
    o, so = Oracle(), SgmOracle()
    scene = scene or min(D, max(8, W // 4))
    Ls = np.stack([o.synth_pair(W, H, scene, 700 + b)[0] for b in range(n)]); Rs = np.stack([o.synth_pair(W, H, scene, 700 + b)[1] for b in range(n)])
    p = jn.Sgm.parameters(num_disparities=D, **kw)
    po = so.params(D, **kw)
    dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
    dD = DeviceArray((n, H, W), np.int16)
    ok = True
    with jn.Sgm(p, W, H, max_batch=n) as s:
        s.process_batch(n, dL.ptr, dR.ptr, W, H * W, dD.ptr)
        print("%dx%d D=%d n=%d %s times %s" % (W, H, D, n, kw, {k: round(v, 3) for k, v in s.last_times().items()}))
        out = dD.numpy()
        ptr, info = s.debug_ptr(0)
        if info[4] != 0:
            wide = info[0]
            vF = d2h(ptr, (n, H, W, D), np.uint16 if wide else np.uint8).astype(np.int32)
            vH0 = d2h(s.debug_ptr(1)[0], (n, H, W, D), np.uint8).astype(np.int32)
            vH1 = d2h(s.debug_ptr(2)[0], (n, H, W, D), np.uint8).astype(np.int32)
            minr = d2h(s.debug_ptr(3)[0], (n, H, W), np.uint32)
            dl = d2h(s.debug_ptr(4)[0], (n, H, W), np.uint32)
            for b in range(min(n, 2)):
                gL, gR = so.prefilter(Ls[b], po.prefilter_cap), so.prefilter(Rs[b], po.prefilter_cap)
                Cv = cost_volume(gL, gR, D)
                m = {dxy: so.path(gL, gR, D, po.P1, po.P2, *dxy).astype(np.int32) - Cv for dxy in ((1, 0), (-1, 0), (0, 1), (1, 1), (-1, 1), (0, -1), (-1, -1), (1, -1))}
                print(" frame", b)
                ob, of = volume_order(D, 0), volume_order(D, wide)           # the volumes hold sums of Y = P2 - (L - C), in the kernels' own order
                ok &= report("H0 (-1,0)", vH0[b], (po.P2 - m[(-1, 0)])[:, ::-1][:, :, ob])
                ok &= report("H1 (+1,0)", vH1[b], (po.P2 - m[(1, 0)])[:, ::-1][:, :, ob])
                ok &= report("F down", vF[b], (3 * po.P2 - (m[(0, 1)] + m[(1, 1)] + m[(-1, 1)]))[:, ::-1][:, :, of])
                S = 8 * Cv + sum(m.values())
                dLexp = S.argmin(axis=2)
                ok &= report("dL", (dl[b] & 0xFFFF).astype(np.int64), dLexp[:, ::-1])
                dRexp = np.full((H, W), -1, np.int64)
                for x in range(W):
                    dm = min(D, W - x)
                    dRexp[:, x] = np.stack([S[:, x + d, d] for d in range(dm)], 1).argmin(axis=1)
                ok &= report("dR", (minr[b] & 0xFFFF).astype(np.int64), dRexp[:, ::-1])
        for b in range(n):
            exp = so.process(po, Ls[b], Rs[b])
            bad = out[b] != exp
            if bad.any():
                ok = False
                idx = np.argwhere(bad)
                print("  final frame %d: %d differ, first %s got %d exp %d" % (b, bad.sum(), idx[0].tolist(), out[b][tuple(idx[0])], exp[tuple(idx[0])]))
            else:
                print("  final frame %d ok" % b)
    for a in (dL, dR, dD):
        a.free()
    return ok


if __name__ == "__main__":
    if len(sys.argv) > 1:
        W, H, D, n = (int(v) for v in sys.argv[1:5])
        kw = {k: int(v) for k, v in (a.split("=") for a in sys.argv[5:])}
        sys.exit(0 if one(W, H, D, n, kw) else 1)
    allok = True
    for (W, H, D, n, kw) in ((64, 40, 64, 1, {}), (200, 150, 128, 2, {}), (333, 101, 64, 2, {"subpixel": 1, "P1": 4, "P2": 30, "prefilter_cap": 15}),
                             (96, 64, 128, 1, {"lr_max_diff": 0}), (160, 120, 256, 1, {"subpixel": 1}), (150, 90, 64, 1, {"P1": 20, "P2": 120, "prefilter_cap": 20})):
        allok &= one(W, H, D, n, kw)
    print("ALL OK" if allok else "FAILURES")
    sys.exit(0 if allok else 1)
