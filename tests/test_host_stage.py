"""The host stage of the product (support-point filters, support list, Delaunay) against the oracle.
CPU only: goes through the jn_host_* hooks of the C-ABI."""
import ctypes as C

import numpy as np
import pytest


class FrameInfo(C.Structure):
    _fields_ = [("ok", C.c_int32), ("nsup", C.c_int32), ("ntri", C.c_int32 * 2), ("sup_offset", C.c_int64),
                ("corner_offset", C.c_int64 * 2), ("reserved", C.c_int64)]


def host_stage(jn, p, W, H, d_can):
    L = jn.load()
    cap = 16 << 20
    payload = np.zeros(cap, np.uint8)
    fi = FrameInfo()
    dc = np.ascontiguousarray(d_can.copy())
    used = L.jn_host_stage(C.byref(p), W, H, dc.ctypes.data, payload.ctypes.data, cap, C.byref(fi))
    assert used >= 0
    sup = np.frombuffer(payload[fi.sup_offset:fi.sup_offset + 12 * fi.nsup].tobytes(), np.int32).reshape(-1, 3)
    corners = [np.frombuffer(payload[fi.corner_offset[s]:fi.corner_offset[s] + 12 * fi.ntri[s]].tobytes(), np.int32).reshape(-1, 3)
               for s in (0, 1)]
    return fi, sup, corners, dc


@pytest.mark.parametrize("W,H,sd,dmax,seed", [(320, 180, 48, 255, 12345), (640, 480, 64, 63, 12345), (333, 201, 30, 95, 5),
                                               (1280, 720, 128, 127, 12345)])
def test_host_stage_matches_oracle(jn, oracle, same, W, H, sd, dmax, seed):
    L, R = oracle.synth_pair(W, H, sd, seed)
    po = oracle.params(0, disp_max=dmax)
    d1, d2 = oracle.descriptor(L), oracle.descriptor(R)
    d_can = oracle.candidates(po, d1, d2)
    sup_o = oracle.support(po, d1, d2)
    fi, sup, corners, _ = host_stage(jn, jn.Elas.parameters(0, disp_max=dmax), W, H, d_can)
    assert fi.ok == 1 and same(sup, sup_o)
    for side in (0, 1):
        c, _ = oracle.triangles(sup_o, side)
        assert same(corners[side], c), "triangle list / corner order differs on side %d" % side


def test_filters_match_oracle_on_random_lattices(jn, oracle, same):
    rng = np.random.default_rng(5)
    W, H = 400, 300          # lattice 80 x 60
    po = oracle.params(0)
    p = jn.Elas.parameters(0)
    for trial in range(10):
        D = rng.integers(0, 60, (60, 80)).astype(np.int16)
        D[rng.random(D.shape) < rng.uniform(0.05, 0.6)] = -1
        D[0, :] = 0; D[:, 0] = 0                                 # row/column 0 stay 0 as in elas.cpp:388-397
        exp = oracle.remove_inconsistent(po, D)
        exp = oracle.remove_redundant(exp, 5, 1, True)
        exp = oracle.remove_redundant(exp, 5, 1, False)
        _, sup, _, filtered = host_stage(jn, p, W, H, D)
        assert same(filtered, exp)
        exp_sup = [(5 * u, 5 * v, exp[v, u]) for u in range(1, 80) for v in range(1, 60) if exp[v, u] >= 0]
        assert same(sup, np.array(exp_sup, np.int32).reshape(-1, 3))


def test_fewer_than_three_support_points(jn):
    D = np.full((36, 64), -1, np.int16)
    D[0, :] = 0; D[:, 0] = 0
    D[10, 10] = 5; D[11, 10] = 5           # two isolated points: removed as inconsistent anyway
    fi, sup, corners, _ = host_stage(jn, jn.Elas.parameters(0), 320, 180, D)
    assert fi.ok == 0 and fi.nsup < 3


def triangulate(jn, pts):
    x = np.ascontiguousarray(pts[:, 0], np.int32)
    y = np.ascontiguousarray(pts[:, 1], np.int32)
    tri = np.zeros((2 * len(pts) + 8, 3), np.int32)
    nt = jn.load().jn_host_triangulate(x.ctypes.data, y.ctypes.data, len(pts), tri.ctypes.data)
    return nt, tri[:max(nt, 0)]


def test_product_delaunay_equals_oracle_delaunay(jn, oracle, same):
    """The product computes the alternating-cut arrangement deterministically (kd-style) instead of
    with Triangle's randomised quick-select; the result must be identical, duplicates included."""
    rng = np.random.default_rng(11)
    checked = 0
    for trial in range(300):
        kind = trial % 4
        if kind == 0:
            gw, gh = int(rng.integers(2, 60)), int(rng.integers(2, 40))
            pts = np.array([(5 * x, 5 * y) for x in range(gw) for y in range(gh)], np.int32)
            pts = pts[rng.random(len(pts)) < rng.uniform(0.1, 1.0)]
        elif kind == 1:
            gw, gh = int(rng.integers(2, 60)), int(rng.integers(2, 30))
            pts = np.array([(5 * x - int(rng.integers(0, 14)), 5 * y) for x in range(gw) for y in range(gh)], np.int32)
        elif kind == 2:
            pts = rng.integers(0, 40, (int(rng.integers(3, 500)), 2)).astype(np.int32)
        else:
            pts = rng.integers(0, 3000, (int(rng.integers(3, 2000)), 2)).astype(np.int32)
        if len(pts) < 3 or len(np.unique(pts, axis=0)) < 2:
            continue
        nt, tri = triangulate(jn, pts)
        exp = oracle.triangulate(pts.astype(np.float32))
        assert nt == len(exp) and same(tri, exp), "trial %d kind %d n %d" % (trial, kind, len(pts))
        checked += 1
    assert checked > 250


def test_delaunay_degenerate_inputs(jn, oracle, same):
    line = np.array([(i * 5, 20) for i in range(12)], np.int32)
    nt, tri = triangulate(jn, line)
    assert nt == 0 and len(oracle.triangulate(line.astype(np.float32))) == 0
    dup = np.array([(5, 5), (5, 5), (10, 5), (5, 10), (10, 5)], np.int32)
    nt, tri = triangulate(jn, dup)
    assert same(tri, oracle.triangulate(dup.astype(np.float32)))
    same_pt = np.array([(7, 7)] * 4, np.int32)
    assert triangulate(jn, same_pt)[0] == -1


def test_delaunay_cut_into_parts_equals_the_sequential_run(jn, hooks, oracle, same, monkeypatch):
    """The phased triangulation (2 or 4 parts on their own threads, merged afterwards; what jn_elas does when its pool has
    idle threads) must reproduce the sequential output exactly, triangle ORDER included — the order decides doubly
    covered pixels downstream.  Lattice points with and without duplicates, random points, sizes around the split
    thresholds, collinear runs."""
    monkeypatch.setenv("JN_DELAUNAY_MIN_POINTS", "64")       # read once per process, at the first phased call: cuts from 64 / 128 points on
    L = jn.load()
    rng = np.random.default_rng(5)
    checked = 0
    for trial in range(160):
        kind = trial % 4
        if kind == 0:      # ELAS-like: 5-px lattice, sparse
            gw, gh = int(rng.integers(20, 256)), int(rng.integers(10, 144))
            pts = np.array([(5 * x, 5 * y) for x in range(1, gw) for y in range(1, gh)], np.int32)
            pts = pts[rng.random(len(pts)) < rng.uniform(0.03, 0.5)]
        elif kind == 1:    # right-image coordinates: columns shifted by a disparity -> duplicates
            gw, gh = int(rng.integers(20, 120)), int(rng.integers(10, 60))
            pts = np.array([(5 * x - int(rng.integers(0, 14)), 5 * y) for x in range(gw) for y in range(gh)], np.int32)
        elif kind == 2:
            pts = rng.integers(0, 60, (int(rng.integers(100, 700)), 2)).astype(np.int32)       # many duplicates
        else:
            pts = rng.integers(0, 3000, (int(rng.integers(120, 4000)), 2)).astype(np.int32)
        if len(pts) < 3:
            continue
        x = np.ascontiguousarray(pts[:, 0]); y = np.ascontiguousarray(pts[:, 1])
        ref = np.zeros((2 * len(pts) + 8, 3), np.int32)
        nref = L.jn_host_triangulate(x.ctypes.data, y.ctypes.data, len(pts), ref.ctypes.data)
        for parts in (2, 4):
            got = np.zeros_like(ref)
            n = L.jn_host_triangulate_parts(x.ctypes.data, y.ctypes.data, len(pts), got.ctypes.data, parts)
            assert n == nref and same(got[:max(n, 0)], ref[:max(nref, 0)]), (trial, kind, len(pts), parts)
        checked += 1
    assert checked > 150
    # against the oracle as well (which emulates Triangle's randomised quick-select), on one large lattice set
    pts = np.array([(5 * x, 5 * y) for x in range(1, 256) for y in range(1, 144)], np.int32)
    pts = pts[rng.random(len(pts)) < 0.09]
    got = np.zeros((2 * len(pts) + 8, 3), np.int32)
    x = np.ascontiguousarray(pts[:, 0]); y = np.ascontiguousarray(pts[:, 1])
    n = L.jn_host_triangulate_parts(x.ctypes.data, y.ctypes.data, len(pts), got.ctypes.data, 4)
    exp = oracle.triangulate(pts.astype(np.float32))
    assert n == len(exp) and same(got[:n], exp)
