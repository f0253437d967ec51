"""The block-matching mode's checker (oracle/bm_oracle.cpp) and its C-ABI surface, on CPU.

SELF-REFERENTIAL: the reference has no block matcher, so nothing here is pinned against reference output.  The oracle is
checked against what CAN be known independently: a literal evaluation of every formula in include/jn_bm.h on small images,
the ground-truth disparities of the synthetic scenes, and structural properties."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bm():
    from oracle.binding import BmOracle
    return BmOracle()


@pytest.fixture(scope="module")
def sgm():
    from oracle.binding import SgmOracle
    return SgmOracle()


def _cost(gA, gB, r, x, y, shift, squared=False):
    """sum over the block of |a(cl(x+i), cr(y+j)) - b(cl(x+i+shift), cr(y+j))| (or its square: JN_BM_COST_SSD) with numpy index clamps"""
    H, W = gA.shape
    ys = np.clip(np.arange(y - r, y + r + 1), 0, H - 1)
    xa = np.clip(np.arange(x - r, x + r + 1), 0, W - 1)
    xb = np.clip(np.arange(x - r, x + r + 1) + shift, 0, W - 1)
    e = gA[np.ix_(ys, xa)].astype(int) - gB[np.ix_(ys, xb)].astype(int)
    return int((e * e).sum() if squared else np.abs(e).sum())


@pytest.mark.parametrize("r", [2, 3, 4])
def test_cost_function_against_numpy(bm, r):
    rng = np.random.default_rng(r)
    H, W = 13, 17
    gL = rng.integers(0, 63, (H, W)).astype(np.uint8); gR = rng.integers(0, 63, (H, W)).astype(np.uint8)
    for (x, y, d) in [(0, 0, 0), (0, 0, 5), (16, 12, 3), (8, 6, 7), (3, 11, 15), (16, 0, 2)]:
        assert bm.cost(gL, gR, r, 0, x, y, d) == _cost(gL, gR, r, x, y, -d)
        assert bm.cost(gL, gR, r, 1, x, y, d) == _cost(gR, gL, r, x, y, d)
        assert bm.cost(gL, gR, r, 0, x, y, d, squared=True) == _cost(gL, gR, r, x, y, -d, True)
        assert bm.cost(gL, gR, r, 1, x, y, d, squared=True) == _cost(gR, gL, r, x, y, d, True)


@pytest.mark.parametrize("r,sub,lr,sq", [(2, 0, 1, 0), (3, 1, 1, 0), (4, 1, 0, 0), (4, 0, -1, 0), (4, 1, 1, 1), (2, 0, 0, 1), (3, 1, -1, 1)])
def test_whole_mode_on_a_small_case_restated(bm, sgm, r, sub, lr, sq):
    """Prefilter, both cost volumes by the literal five-loop cost, first-minimum WTA on each side, L/R check, 1/16 formula."""
    rng = np.random.default_rng(11 + r)
    H, W, D = 12, 26, 8
    base = rng.integers(0, 256, (H, W + D)).astype(np.uint8)
    R = base[:, D:].copy(); L = base[:, D - 3:W + D - 3].copy()          # true disparity 3
    got = bm.process(bm.params(D, r, 31, lr, sub, sq), L, R)
    gL, gR = sgm.prefilter(L), sgm.prefilter(R)
    CL = np.array([[[_cost(gL, gR, r, x, y, -d, bool(sq)) for d in range(D)] for x in range(W)] for y in range(H)])
    CR = np.array([[[_cost(gR, gL, r, x, y, d, bool(sq)) for d in range(D)] for x in range(W)] for y in range(H)])
    dL, dR = CL.argmin(axis=2), CR.argmin(axis=2)                        # numpy argmin = first minimum
    exp = np.zeros((H, W), int)
    for y in range(H):
        for x in range(W):
            d = dL[y, x]
            ok = lr < 0 or (x - d >= 0 and abs(d - dR[y, x - d]) <= lr)
            v = -(16 if sub else 1)
            if ok:
                v = d * (16 if sub else 1)
                if sub and 0 < d < D - 1:
                    den = max(CL[y, x, d - 1] + CL[y, x, d + 1] - 2 * CL[y, x, d], 1)
                    v = 16 * d + int((16 * (CL[y, x, d - 1] - CL[y, x, d + 1]) + den) / (2 * den))     # C division truncates toward zero
            exp[y, x] = v
    assert np.array_equal(got.astype(int), exp)
    inner = got[:, 8:-2]
    assert (np.abs(inner / (16.0 if sub else 1.0) - 3) <= 0.5).mean() > 0.9


@pytest.mark.parametrize("W,H,D,scene", [(320, 180, 64, 48), (333, 101, 64, 30)])
def test_oracle_recovers_the_synthetic_ground_truth(bm, sgm, oracle, W, H, D, scene):
    L, R = oracle.synth_pair(W, H, scene, 12345)
    disp = bm.process(bm.params(D), L, R)
    yy, xx = np.mgrid[0:H, 0:W]
    gt = (yy / H * (scene * 0.6)).astype(int) + 2
    gt[(xx > W // 3) & (xx < W // 2) & (yy > H // 3) & (yy < 2 * H // 3)] = int(scene * 0.7)
    v = disp >= 0
    assert v.mean() > 0.85 and (np.abs(disp[v] - gt[v]) <= 1).mean() > 0.97
    sub = bm.process(bm.params(D, subpixel=1), L, R)
    assert np.array_equal(sub < 0, disp < 0)
    assert np.abs(sub[v] / 16.0 - disp[v]).max() <= 0.5 + 1e-9               # the refinement moves by at most half a pixel
    u8 = sgm.to_u8(sub, 1)                                                   # the output format is the SGM mode's
    assert u8[~v].max(initial=0) == 0 and np.abs(u8[v].astype(int) - disp[v]).max() <= 1


@pytest.mark.parametrize("name,cf", [("bm_hashes.txt", 0), ("bm_ssd_hashes.txt", 1)])
def test_committed_hashes_are_the_oracles(bm, oracle, name, cf):
    rows = [l.split() for l in open(os.path.join(ROOT, "tests", "golden", name)) if not l.startswith("#")]
    W, H, scene, D, r, sub, seed, h = rows[0]
    L, R = oracle.synth_pair(int(W), int(H), int(scene), int(seed))
    d = bm.process(bm.params(int(D), int(r), subpixel=int(sub), cost_function=cf), L, R)
    assert "%016x" % oracle.fnv(d.view(np.uint32)) == h


def test_ssd_cost_recovers_the_synthetic_ground_truth_too(bm, oracle):
    W, H, D, scene = 320, 180, 64, 48
    L, R = oracle.synth_pair(W, H, scene, 12345)
    disp = bm.process(bm.params(D, cost_function=1), L, R)
    yy, xx = np.mgrid[0:H, 0:W]
    gt = (yy / H * (scene * 0.6)).astype(int) + 2
    gt[(xx > W // 3) & (xx < W // 2) & (yy > H // 3) & (yy < 2 * H // 3)] = int(scene * 0.7)
    v = disp >= 0
    assert v.mean() > 0.85 and (np.abs(disp[v] - gt[v]) <= 1).mean() > 0.97


def test_parameters_outside_the_definition_are_refused(bm):
    L = np.zeros((16, 16), np.uint8)
    for kw in ({"prefilter_cap": 40}, {"block_radius": 0}, {"num_disparities": 300}, {"cost_function": 2}):
        with pytest.raises(ValueError):
            bm.process(bm.params(**kw), L, L)


def test_bm_header_symbols_are_exported(jn):
    text = open(os.path.join(ROOT, "include", "jn_bm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(jn_bm_[a-z0-9_]+)\s*\(", text)))
    lib = jn.load()
    assert declared == sorted(jn.BM_EXPORTS) and all(hasattr(lib, n) for n in declared)
    p = jn.Bm.parameters()
    assert (p.num_disparities, p.block_radius, p.prefilter_cap, p.lr_max_diff, p.subpixel) == (64, 4, 31, 1, 0)
    import ctypes as C
    from jackal_navigation_amd import _lib
    from jackal_navigation_amd.device import device_count
    if device_count() == 0:
        with pytest.raises(_lib.JnError) as e:
            jn.Bm(p, 64, 48)
        assert e.value.status == _lib.JN_ERR_NO_DEVICE                       # no CPU fallback
    h = C.c_void_p()
    for bad in (jn.Bm.parameters(num_disparities=100), jn.Bm.parameters(block_radius=5), jn.Bm.parameters(block_radius=1)):
        assert jn.load().jn_bm_create(C.byref(bad), 64, 48, 1, 0, C.byref(h)) == _lib.JN_ERR_UNSUPPORTED
