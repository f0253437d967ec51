"""Pins the CPU oracle against the REAL reference library compiled from /root/reference
(oracle/_ref).  Skipped where that library was not built; the committed golden vectors
(test_oracle_golden.py) carry the same evidence everywhere else."""
import numpy as np
import pytest


@pytest.mark.parametrize("W,H,sd,dmax,seed,both", [
    (320, 180, 48, 255, 12345, False), (256, 200, 40, 63, 1, True), (333, 201, 30, 95, 5, False),
    (640, 480, 64, 63, 12345, False), (200, 150, 20, 40, 9, True),
])
def test_process_bit_exact(oracle, reference, same, W, H, sd, dmax, seed, both):
    L, R = oracle.synth_pair(W, H, sd, seed)
    p = oracle.params(0, disp_max=dmax, postprocess_only_left=0 if both else 1)
    st, D1, D2 = oracle.process(p, L, R)
    D1r, D2r = reference.process(p, L, R)
    assert st == 0 and same(D1, D1r) and same(D2, D2r)


@pytest.mark.parametrize("kw", [
    {"filter_median": 1}, {"filter_median": 1, "filter_adaptive_mean": 0, "postprocess_only_left": 0},
    {"incon_window_size": 3, "incon_min_support": 3}, {"ipol_gap_width": 7, "speckle_size": 50},
])
def test_process_bit_exact_with_other_parameters(oracle, reference, same, kw):
    """The optional median filter (elas.cpp:1494-1560, off in the node's preset) and a few non-default tunables."""
    L, R = oracle.synth_pair(320, 240, 40, 21)
    p = oracle.params(0, disp_max=79, **kw)
    st, D1, D2 = oracle.process(p, L, R)
    D1r, D2r = reference.process(p, L, R)
    assert st == 0 and same(D1, D1r) and same(D2, D2r), kw


@pytest.mark.parametrize("kind", ["strips", "patches", "slanted", "photometric", "blobs", "shallow"])
def test_process_bit_exact_on_other_scenes(oracle, reference, same, kind):
    """Scenes unlike the survey's plane-and-box generator (tests/scenes.py): depth jumps and occlusions, textureless
    patches, slanted surfaces, gain / offset / noise between the images, random blobs."""
    from scenes import make_scene
    for (W, H, dmax, seed) in ((320, 240, 79, 5), (400, 304, 127, 6)):
        L, R = make_scene(kind, W, H, dmax, seed)
        p = oracle.params(0, disp_max=dmax, postprocess_only_left=0)
        st, D1, D2 = oracle.process(p, L, R)
        D1r, D2r = reference.process(p, L, R)
        assert st == 0 and same(D1, D1r) and same(D2, D2r), (kind, W, H)


def test_stagewise_bit_exact_720p(oracle, reference, same):
    W, H = 1280, 720
    L, R = oracle.synth_pair(W, H, 128, 12345)
    p = oracle.params(0, disp_max=127)
    d1, d2 = oracle.descriptor(L), oracle.descriptor(R)
    with reference.open(p, L, R) as s:
        assert same(d1[3:H - 3, 3:W - 3], s.descriptor(0)[3:H - 3, 3:W - 3])
        sup = s.support()
        assert same(oracle.support(p, d1, d2), sup)
        for side in (0, 1):
            cr, plr = s.triangles(side, len(sup))
            c, pl = oracle.triangles(sup, side)
            assert same(c, cr) and same(pl, plr)
            gr = s.grid(side)
            assert same(oracle.grid(p, sup, W, H, side), gr)
            assert same(oracle.dense(p, sup, cr, plr, gr, d1, d2, side), s.dense(side))


def test_support_filters_in_place_order(oracle, reference, same):
    """removeInconsistent/removeRedundant are scan-order dependent (elas.cpp:153-235)."""
    rng = np.random.default_rng(0)
    p = oracle.params(0)
    L, R = oracle.synth_pair(160, 120, 20, 3)
    with reference.open(p, L, R) as s:
        for _ in range(20):
            D = rng.integers(-1, 40, (30, 50)).astype(np.int16)
            D[rng.random(D.shape) < 0.3] = -1
            assert same(oracle.remove_inconsistent(p, D), s.remove_inconsistent(D))
            for vert in (True, False):
                assert same(oracle.remove_redundant(D, 5, 1, vert), s.remove_redundant(D, 5, 1, vert))


def test_delaunay_tie_breaks(oracle, reference, same):
    """Lattice points are cocircular everywhere: the triangulation is whatever Triangle's D&C
    produces, corner order included.  Also right-image style inputs with duplicate vertices."""
    rng = np.random.default_rng(7)
    n_checked = 0
    for trial in range(200):
        kind = trial % 4
        if kind == 0:
            gw, gh = int(rng.integers(2, 30)), int(rng.integers(2, 30))
            pts = np.array([(5 * x, 5 * y) for x in range(gw) for y in range(gh)], np.float32)
            pts = pts[rng.random(len(pts)) < rng.uniform(0.2, 1.0)]
        elif kind == 1:
            gw, gh = int(rng.integers(2, 40)), int(rng.integers(2, 20))
            pts = np.array([(5 * x - int(rng.integers(0, 12)), 5 * y) for x in range(gw) for y in range(gh)], np.float32)
        elif kind == 2:
            pts = rng.integers(0, 30, (int(rng.integers(3, 300)), 2)).astype(np.float32)
        else:
            n = int(rng.integers(3, 60))
            pts = np.stack([rng.integers(0, 50, n), np.full(n, 7)], 1).astype(np.float32)   # collinear
            if trial % 8 == 3:
                pts[0] = (3, 9)
        if len(pts) < 3 or len(np.unique(pts, axis=0)) < 2:
            continue
        assert same(oracle.triangulate(pts), reference.triangulate(pts)), "trial %d kind %d" % (trial, kind)
        n_checked += 1
    assert n_checked > 150


def test_sobel_rows(oracle, reference, same):
    rng = np.random.default_rng(1)
    I = rng.integers(0, 256, (40, 64)).astype(np.uint8)
    du, dv = oracle.sobel(I)
    dur, dvr = reference.sobel(I)
    assert same(du[1:-1, 1:-1], dur[1:-1, 1:-1]) and same(dv[1:-2, 1:-1], dvr[1:-2, 1:-1])
