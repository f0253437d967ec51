"""Pins the CPU oracle against the REAL reference library compiled from /root/reference
(oracle/_ref).  Skipped where that library was not built; the committed golden vectors
(test_oracle_golden.py) carry the same evidence everywhere else.

Every reference call here goes through oracle.binding.Reference: a worker process whose heap hands out zero-filled
memory, because libelas reads descriptor bytes it never wrote (test_reference_reads_the_uninitialised_descriptor_border
below names the lines).  The suite therefore gives the same answers whatever MALLOC_PERTURB_ / heap history the pytest
process itself has."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    {"disp_min": 6}, {"disp_min": 20, "postprocess_only_left": 0}, {"disp_min": -5},      # elas.cpp:323-333: only the support matching reads disp_min
])
def test_process_bit_exact_with_other_parameters(oracle, reference, same, kw):
    """The optional median filter (elas.cpp:1494-1560, off in the node's preset) and a few non-default tunables."""
    L, R = oracle.synth_pair(320, 240, 40, 21)
    p = oracle.params(0, disp_max=79, **kw)
    st, D1, D2 = oracle.process(p, L, R)
    D1r, D2r = reference.process(p, L, R)
    assert st == 0 and same(D1, D1r) and same(D2, D2r), kw


@pytest.mark.parametrize("kind", ["strips", "patches", "slanted", "photometric", "blobs", "shallow", "grain", "periodic"])
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


def test_oracle_equals_reference_on_the_middlebury_preset_with_zero_filled_memory(oracle, reference, same):
    """MIDDLEBURY (elas.h:118-145): add_corners, unbounded gap interpolation with border extrapolation, median filter, both
    sides post-processed.  With the uninitialised descriptor bytes zero (the `reference` fixture's worker) the reference is
    deterministic and the oracle (which defines them as zero) must equal it bit for bit.
    (add_corners together with the ADAPTIVE MEAN — no preset does that — is left out: the reference's vertical pass then also
    reads rows of its scratch image that nothing wrote, elas.cpp:1298 / :1436-1446, and the oracle defines those as a copy.)"""
    from scenes import make_scene
    pairs = [oracle.synth_pair(320, 180, 48, 12345), oracle.synth_pair(400, 300, 60, 7), make_scene("strips", 320, 240, 79, 21),
             make_scene("blobs", 320, 240, 79, 21)]
    for (L, R), d in zip(pairs, [255, 127, 79, 79]):
        for kw in ({}, {"ipol_gap_width": 7}, {"filter_median": 0}, {"sradius": 2.0, "gamma": 3.0, "match_texture": 1}):
            D1r, D2r = reference.process(reference.params(1, disp_max=d, **kw), L, R)
            st, D1o, D2o = oracle.process(oracle.params(1, disp_max=d, **kw), L, R)
            assert st == 0 and same(D1r, D1o) and same(D2r, D2o), (d, kw)


NODE_SCENES = (("strips", 320, 240, 79, 5), ("slanted", 400, 304, 127, 6), ("photometric", 400, 304, 127, 6))


def test_reference_reads_the_uninitialised_descriptor_border(oracle, reference, same):
    """What the reference is pinned TO.  libelas' descriptor image comes from _mm_malloc and is never initialised
    (descriptor.cpp:29); createDescriptor writes u in [3,W-4], v in [3,H-4] only (descriptor.cpp:84-88).  Two places read
    outside that, also with the node's own configuration (ROBOTICS, postprocess_only_left=1, point_cloud.cpp:416-418):
      * the right-image support match: disp_max_valid = W-u-5 (elas.cpp:326) allows u+d = W-5, whose +2 tap is column W-3
        (loads at elas.cpp:340-349);
      * findMatch: `u_warp<window_size || u_warp>=width-window_size` with window_size 2 (elas.cpp:744-746, :752-754,
        :763-765, :770-772) admits warp columns 2 and W-3.
    Shown here: (1) the reference's D1 moves when its heap is filled with another byte; (2) an oracle whose descriptor
    border holds that same byte reproduces the reference bit for bit under BOTH fills — so this border is the only
    uninitialised memory the result depends on; (3) zero fill — freshly mapped pages, the product's and the oracle's
    definition (include/jn_stereo.h) — is what the goldens and every other test are pinned to."""
    from oracle.binding import Reference
    from scenes import make_scene
    other = Reference(fill=0x7f)
    try:
        moved = 0
        for kind, W, H, dmax, seed in NODE_SCENES:
            L, R = make_scene(kind, W, H, dmax, seed)
            p = oracle.params(0, disp_max=dmax)                                 # the node's parameters
            assert p.postprocess_only_left == 1 and p.add_corners == 0
            D1z, D2z = reference.process(p, L, R)
            D1f, D2f = other.process(p, L, R)
            moved += int((D1z.view(np.uint32) != D1f.view(np.uint32)).sum())
            st, D1o, D2o = oracle.process(p, L, R)
            assert st == 0 and same(D1o, D1z) and same(D2o, D2z), ("zero fill", kind)
            oracle.set_uninit_fill(0x7f)
            try:
                st, D1o, D2o = oracle.process(p, L, R)
            finally:
                oracle.set_uninit_fill(0)
            assert st == 0 and same(D1o, D1f) and same(D2o, D2f), ("0x7f fill", kind)
        assert moved > 100, "the reference's D1 no longer depends on the fill byte?"
    finally:
        other.close()


def test_reference_worker_ignores_the_callers_allocator_settings(oracle, same):
    """The pin must not depend on the pytest process: a Reference() created under MALLOC_PERTURB_=128 still zero-fills."""
    from oracle.binding import Reference
    from scenes import make_scene
    if not Reference.available():
        pytest.skip("oracle/_ref/libelas_ref.so not built (needs /root/reference)")
    old = os.environ.get("MALLOC_PERTURB_")
    os.environ["MALLOC_PERTURB_"] = "128"
    try:
        r = Reference()
    finally:
        if old is None:
            del os.environ["MALLOC_PERTURB_"]
        else:
            os.environ["MALLOC_PERTURB_"] = old
    try:
        kind, W, H, dmax, seed = NODE_SCENES[0]
        L, R = make_scene(kind, W, H, dmax, seed)
        p = oracle.params(0, disp_max=dmax)
        D1r, D2r = r.process(p, L, R)
        st, D1o, D2o = oracle.process(p, L, R)
        assert st == 0 and same(D1o, D1r) and same(D2o, D2r)
    finally:
        r.close()


def test_oracle_reproduces_the_committed_middlebury_hashes(oracle):
    rows = [l.split() for l in open(os.path.join(ROOT, "tests", "golden", "reference_middlebury_hashes.txt")) if not l.startswith("#")]
    sys_path_scenes = os.path.join(ROOT, "tests")
    import sys
    if sys_path_scenes not in sys.path:
        sys.path.insert(0, sys_path_scenes)
    from scenes import make_scene
    for kind, W, H, sd, dmax, seed, h1, h2 in rows:
        W, H, sd, dmax, seed = int(W), int(H), int(sd), int(dmax), int(seed)
        if W * H > 640 * 480:
            continue                                                   # the 720p row is for the GPU test
        L, R = oracle.synth_pair(W, H, sd, seed) if kind == "synth" else make_scene(kind, W, H, dmax, seed)
        st, D1, D2 = oracle.process(oracle.params(1, disp_max=dmax), L, R)
        assert st == 0 and oracle.fnv(D1) == int(h1, 16) and oracle.fnv(D2) == int(h2, 16), (kind, W, H)


@pytest.mark.parametrize("W,H,sd,dmax,seed,kw", [
    (320, 240, 40, 79, 21, {}), (320, 240, 40, 79, 21, {"postprocess_only_left": 0}),
    (640, 480, 64, 63, 12345, {"postprocess_only_left": 0}), (256, 200, 40, 63, 1, {"filter_median": 1, "postprocess_only_left": 0}),
    (400, 304, 60, 127, 6, {"candidate_stepsize": 4, "ipol_gap_width": 7, "speckle_size": 50, "postprocess_only_left": 0}),
    (322, 182, 48, 255, 12345, {"filter_adaptive_mean": 0}),
])
def test_process_bit_exact_with_subsampling(oracle, reference, same, W, H, sd, dmax, seed, kw):
    """subsampling = 1 (elas.h:82; elas.cpp:380, 693, 793, 877-896, 914-941, 987-992, 1107-1112, 1292-1297, 1323-1391, 1499-1504;
    descriptor.cpp:47-78): half-size maps.  The node never sets it; the oracle restates it for even image sizes."""
    L, R = oracle.synth_pair(W, H, sd, seed)
    p = oracle.params(0, disp_max=dmax, subsampling=1, **kw)
    st, D1, D2 = oracle.process(p, L, R)
    D1r, D2r = reference.process(p, L, R)
    assert D1.shape == (H // 2, W // 2) == D1r.shape
    assert st == 0 and same(D1, D1r) and same(D2, D2r), (W, H, kw, int((D1 != D1r).sum()), int((D2 != D2r).sum()))
