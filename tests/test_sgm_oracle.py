"""The SGM mode's checker (oracle/sgm_oracle.cpp) and its C-ABI surface, on CPU.

SELF-REFERENTIAL: the reference has no SGM, so nothing here is pinned against reference output.  The oracle is checked
against what CAN be known independently: hand-computed tiny cases of every formula in include/jn_sgm.h, the ground-truth
disparities of the synthetic scenes, and structural properties."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sgm():
    from oracle.binding import SgmOracle
    return SgmOracle()


def test_prefilter_is_clipped_sobel_x_with_replicated_borders(sgm):
    rng = np.random.default_rng(0)
    I = rng.integers(0, 256, (9, 11)).astype(np.uint8)
    g = sgm.prefilter(I, 31)
    P = np.pad(I.astype(int), 1, mode="edge")
    sx = (P[:-2, 2:] - P[:-2, :-2]) + 2 * (P[1:-1, 2:] - P[1:-1, :-2]) + (P[2:, 2:] - P[2:, :-2])
    assert np.array_equal(g, (np.clip(sx, -31, 31) + 31).astype(np.uint8))
    assert np.array_equal(sgm.prefilter(I, 5), (np.clip(sx, -5, 5) + 5).astype(np.uint8))


def _cost(gL, gR, x, y, d):
    W = gL.shape[1]
    return sum(abs(int(gL[y, min(max(x + i, 0), W - 1)]) - int(gR[y, min(max(x + i - d, 0), W - 1)])) for i in (-1, 0, 1))


@pytest.mark.parametrize("dx,dy", [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (-1, 1), (1, -1)])
def test_path_recurrence_against_a_python_restatement(sgm, dx, dy):
    rng = np.random.default_rng(3)
    H, W, D, P1, P2 = 7, 9, 6, 3, 11
    gL = rng.integers(0, 63, (H, W)).astype(np.uint8); gR = rng.integers(0, 63, (H, W)).astype(np.uint8)
    got = sgm.path(gL, gR, D, P1, P2, dx, dy)
    exp = np.zeros((H, W, D), int)
    for y0 in range(H):
        for x0 in range(W):
            if 0 <= x0 - dx < W and 0 <= y0 - dy < H:
                continue
            x, y, prev = x0, y0, None
            while 0 <= x < W and 0 <= y < H:
                c = [_cost(gL, gR, x, y, d) for d in range(D)]
                if prev is None:
                    cur = c
                else:
                    mp = min(prev)
                    cur = [c[d] + min([prev[d], mp + P2] + ([prev[d - 1] + P1] if d > 0 else []) + ([prev[d + 1] + P1] if d + 1 < D else [])) - mp
                           for d in range(D)]
                exp[y, x] = cur
                prev = cur
                x += dx; y += dy
    assert np.array_equal(got.astype(int), exp)


def test_whole_mode_on_a_tiny_case_by_hand(sgm):
    """Sum of the 8 paths, first-minimum WTA, right-image WTA along x+d, L/R check and the 1/16 refinement, restated."""
    rng = np.random.default_rng(11)
    H, W, D = 10, 24, 8
    base = rng.integers(0, 256, (H, W + D)).astype(np.uint8)
    R = base[:, D:].copy(); L = base[:, D - 3:W + D - 3].copy()          # true disparity 3
    for sub in (0, 1):
        p = sgm.params(D, 4, 20, 31, 1, sub)
        got = sgm.process(p, L, R)
        gL, gR = sgm.prefilter(L), sgm.prefilter(R)
        S = np.zeros((H, W, D), int)
        for dx, dy in [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (-1, 1), (1, -1)]:
            S += sgm.path(gL, gR, D, 4, 20, dx, dy)
        dL = S.argmin(axis=2)                                                # numpy argmin = first minimum
        exp = np.zeros((H, W), int)
        for y in range(H):
            dR = [min(range(min(D, W - x)), key=lambda d: (S[y, x + d, d], d)) for x in range(W)]
            for x in range(W):
                d = dL[y, x]
                ok = x - d >= 0 and abs(d - dR[x - d]) <= 1
                v = -(16 if sub else 1)
                if ok:
                    v = d * (16 if sub else 1)
                    if sub and 0 < d < D - 1:
                        den = max(S[y, x, d - 1] + S[y, x, d + 1] - 2 * S[y, x, d], 1)
                        num = 16 * (S[y, x, d - 1] - S[y, x, d + 1]) + den
                        v = 16 * d + int(num / (2 * den))                   # C division truncates toward zero
                exp[y, x] = v
        assert np.array_equal(got.astype(int), exp), sub
        inner = got[2:-2, 6:-2]
        assert (np.abs(inner / (16.0 if sub else 1.0) - 3) <= 0.5).mean() > 0.9


@pytest.mark.parametrize("W,H,D,scene", [(320, 180, 64, 48), (333, 101, 64, 30)])
def test_oracle_recovers_the_synthetic_ground_truth(sgm, oracle, W, H, D, scene):
    L, R = oracle.synth_pair(W, H, scene, 12345)
    disp = sgm.process(sgm.params(D), L, R)
    yy, xx = np.mgrid[0:H, 0:W]
    gt = (yy / H * (scene * 0.6)).astype(int) + 2
    gt[(xx > W // 3) & (xx < W // 2) & (yy > H // 3) & (yy < 2 * H // 3)] = int(scene * 0.7)
    v = disp >= 0
    assert v.mean() > 0.85 and (np.abs(disp[v] - gt[v]) <= 1).mean() > 0.97
    sub = sgm.process(sgm.params(D, subpixel=1), L, R)
    assert np.array_equal(sub < 0, disp < 0)
    assert np.abs(sub[v] / 16.0 - disp[v]).max() <= 0.5 + 1e-9               # the refinement moves by at most half a pixel
    u8 = sgm.to_u8(sub, 1)
    assert u8[~v].max(initial=0) == 0 and np.abs(u8[v].astype(int) - disp[v]).max() <= 1


def test_parameters_outside_the_definition_are_refused(sgm):
    L = np.zeros((16, 16), np.uint8)
    for kw in ({"P2": 80}, {"prefilter_cap": 40}, {"P1": 30, "P2": 20}):
        with pytest.raises(ValueError):
            sgm.process(sgm.params(64, **kw), L, L)


def test_sgm_header_symbols_are_exported(jn):
    text = open(os.path.join(ROOT, "include", "jn_sgm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(jn_sgm_[a-z0-9_]+)\s*\(", text)))
    lib = jn.load()
    assert declared == sorted(jn.SGM_EXPORTS) and all(hasattr(lib, n) for n in declared)
    p = jn.Sgm.parameters()
    assert (p.num_disparities, p.P1, p.P2, p.prefilter_cap, p.lr_max_diff, p.subpixel) == (128, 10, 60, 31, 1, 0)
    import ctypes as C
    from jackal_navigation_amd import _lib
    from jackal_navigation_amd.device import device_count
    if device_count() == 0:
        with pytest.raises(_lib.JnError) as e:
            jn.Sgm(p, 64, 48)
        assert e.value.status == _lib.JN_ERR_NO_DEVICE                       # no CPU fallback
    h = C.c_void_p()
    bad = jn.Sgm.parameters(num_disparities=100)
    assert jn.load().jn_sgm_create(C.byref(bad), 64, 48, 1, 0, C.byref(h)) == _lib.JN_ERR_UNSUPPORTED
