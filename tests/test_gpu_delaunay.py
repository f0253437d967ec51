"""The hull recursion of the Delaunay triangulation on the GPU (csrc/delaunay_gpu.hip: k_arrange + k_delaunay, what a batch handle runs instead of
the host stage) against the host's exact replay of Triangle (csrc/delaunay.cpp, itself pinned against the compiled reference's Triangle on 200
tie-break sets in tests/test_oracle_vs_reference.py): the same triangles in the same order, both image sides."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def host_tri(L, x, y):
    n = len(x)
    tri = np.zeros(6 * max(n, 1), np.int32)
    k = L.jn_host_triangulate(np.ascontiguousarray(x, np.int32).ctypes.data, np.ascontiguousarray(y, np.int32).ctypes.data, n, tri.ctypes.data)
    return k, tri[:3 * max(k, 0)].reshape(-1, 3)


def device_tri(L, triples, step):
    n = len(triples)
    tl = np.zeros(6 * max(n, 1), np.int32); tr = np.zeros(6 * max(n, 1), np.int32)
    ntri = (C.c_int32 * 2)(); need = C.c_int32()
    t = np.ascontiguousarray(triples, np.int16)
    assert L.jn_device_triangulate(0, t.ctypes.data, n, step, tl.ctypes.data, tr.ctypes.data, ntri, C.byref(need)) == 0
    return (ntri[0], tl[:3 * ntri[0]].reshape(-1, 3)), (ntri[1], tr[:3 * ntri[1]].reshape(-1, 3)), need.value


def lattice_case(rng, n, cw, ch, dmax, step=5, row_d=False):
    cells = rng.choice(cw * ch, size=n, replace=False)
    cells.sort()                                            # uc-major, vc ascending = the support list's order
    uc, vc = cells // ch, cells % ch
    d = (vc * 7) % (dmax + 1) if row_d else rng.integers(0, dmax + 1, n)
    return np.stack([uc, vc, d], axis=1).astype(np.int16)


def test_gpu_triangulation_equals_the_hosts_on_lattices(jn):
    """Random subsets of the support lattice (collinear runs, co-circular quadruples and equal coordinates everywhere: every tie-break of
    the recursion is exercised), sizes from 3 vertices to what 156 KB of LDS hold; the right side's x = u - d scatters the columns."""
    L = jn.load()
    rng = np.random.default_rng(7)
    sides_checked = 0
    for n in (3, 4, 5, 6, 7, 8, 9, 12, 13, 31, 64, 100, 257, 819, 1024, 2047, 3232, 3350, 4200, 4600):
        for rep in range(3 if n < 2000 else 1):
            t = lattice_case(rng, n, 256, 144, 127, row_d=(rep == 1))
            (kl, tl), (kr, tr), need = device_tri(L, t, 5)
            for side, (k, tri) in ((0, (kl, tl)), (1, (kr, tr))):
                x = t[:, 0].astype(np.int32) * 5 - (t[:, 2].astype(np.int32) if side else 0); y = t[:, 1].astype(np.int32) * 5
                if need & (1 << side):
                    assert len(set(zip(x.tolist(), y.tolist()))) < n, (n, side)       # handed back only because vertices coincide
                    continue
                ke, te = host_tri(L, x, y)
                assert k == ke, (n, side, k, ke)
                assert np.array_equal(tri, te), (n, side, int((tri != te).any(axis=1).sum()))
                sides_checked += 1
    assert sides_checked >= 60


def test_gpu_triangulation_on_support_points_of_real_frames(jn, oracle):
    """The support points the matcher really produces (Appendix-A pairs and two scene kinds, 720p included): the GPU's triangles equal the host's."""
    from scenes import make_scene
    L = jn.load()
    cases = [oracle.synth_pair(1280, 720, 128, 12345) + (127,), oracle.synth_pair(640, 480, 64, 5) + (63,), make_scene("blobs", 640, 360, 95, 3) + (95,),
             make_scene("grain", 640, 360, 95, 4) + (95,)]
    for Lm, Rm, dmax in cases:
        p = oracle.params(0, disp_max=dmax)
        sup = np.asarray(oracle.support(p, oracle.descriptor(Lm), oracle.descriptor(Rm)))          # (u, v, d)
        t = np.stack([sup[:, 0] // 5, sup[:, 1] // 5, sup[:, 2]], axis=1).astype(np.int16)
        (kl, tl), (kr, tr), need = device_tri(L, t, 5)
        assert need == 0
        for side, (k, tri) in ((0, (kl, tl)), (1, (kr, tr))):
            x = sup[:, 0].astype(np.int32) - (sup[:, 2].astype(np.int32) if side else 0); y = sup[:, 1].astype(np.int32)
            ke, te = host_tri(L, x, y)
            assert k == ke and np.array_equal(tri, te), (len(sup), side)


def test_sides_the_gpu_cannot_take_are_handed_back(jn):
    L = jn.load()
    rng = np.random.default_rng(3)
    t = lattice_case(rng, 5200, 384, 216, 255, row_d=True)                 # more vertices than the LDS holds
    _, _, need = device_tri(L, t, 5)
    assert need == 3
    t = lattice_case(rng, 300, 256, 144, 127)
    t[10] = (t[9][0] + 1, t[9][1], t[9][2] + 5)                             # (u - d, v) of two right-image vertices coincide; the left side is fine
    t = t[np.lexsort((t[:, 1], t[:, 0]))]
    (kl, tl), _, need = device_tri(L, t, 5)
    assert need & 2
    if not need & 1:
        ke, te = host_tri(L, t[:, 0].astype(np.int32) * 5, t[:, 1].astype(np.int32) * 5)
        assert kl == ke and np.array_equal(tl, te)
    for n in (0, 1, 2):                                                     # fewer than three support points: no triangles, nothing for the host either (elas.cpp:66-71)
        (kl, _), (kr, _), need = device_tri(L, lattice_case(rng, n, 256, 144, 127) if n else np.zeros((0, 3), np.int16), 5)
        assert (kl, kr, need) == (0, 0, 0)
