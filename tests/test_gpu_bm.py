"""Block-matching mode on the GPU: bit-exact against its scalar definition (oracle/bm_oracle.cpp; self-referential — the
reference has no block matcher), through the C-ABI (jn_bm_*)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bm():
    from oracle.binding import BmOracle
    return BmOracle()


@pytest.fixture(scope="module")
def sgm():
    from oracle.binding import SgmOracle
    return SgmOracle()


def run(jn, p, Ls, Rs):
    from jackal_navigation_amd.device import DeviceArray
    n, H, W = Ls.shape
    dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
    dD = DeviceArray((n, H, W), np.int16)
    with jn.Bm(p, W, H, max_batch=n) as s:
        s.process_batch(n, dL.ptr, dR.ptr, W, H * W, dD.ptr)
        t = s.last_times()
        du8 = DeviceArray((n, H, W), np.uint8)
        s.to_u8(dD.ptr, du8.ptr, n * H * W)
    out, u8 = dD.numpy(), du8.numpy()
    for a in (dL, dR, dD, du8):
        a.free()
    return out, u8, t


@pytest.mark.parametrize("W,H,D,scene,n,kw", [
    (640, 480, 64, 64, 2, {}),                                     # BASELINE config 2's frame, range and matcher
    (1280, 720, 128, 128, 1, {"subpixel": 1}),                     # BASELINE config 3's frame and range, 1/16 pixel
    (320, 180, 256, 48, 2, {"subpixel": 1, "block_radius": 3}),    # D = 256, 7x7
    (333, 101, 64, 30, 3, {"block_radius": 2, "prefilter_cap": 15}),   # ragged size (partial column group, partial band), 5x5
    (200, 150, 128, 90, 2, {"lr_max_diff": -1}),                   # no L/R check (no right-referenced pass)
    (96, 64, 128, 20, 1, {"lr_max_diff": 0, "subpixel": 1}),       # image narrower than the disparity range
    (70, 9, 8, 5, 2, {"block_radius": 4}),                         # fewer rows than the block is high, smallest range
    (640, 480, 72, 64, 1, {"block_radius": 3, "subpixel": 1}),     # D not a multiple of 32: uneven rounds over the four waves
])
def test_bm_bit_exact_vs_its_definition(jn, bm, sgm, oracle, W, H, D, scene, n, kw):
    Ls = np.stack([oracle.synth_pair(W, H, scene, 700 + b)[0] for b in range(n)])
    Rs = np.stack([oracle.synth_pair(W, H, scene, 700 + b)[1] for b in range(n)])
    out, u8, t = run(jn, jn.Bm.parameters(num_disparities=D, **kw), Ls, Rs)
    po = bm.params(D, **kw)
    for b in range(n):
        exp = bm.process(po, Ls[b], Rs[b])
        assert np.array_equal(out[b], exp), (b, int((out[b] != exp).sum()))
        assert np.array_equal(u8[b], sgm.to_u8(exp, kw.get("subpixel", 0)))
    assert t["match"] > 0 and t["total"] >= t["match"]


@pytest.mark.parametrize("band", ["8", "16", "37", "5", "64"])
def test_bm_band_heights_give_the_same_map(jn, hooks, bm, oracle, monkeypatch, band):
    """The launch picks the rows per workgroup band from the batch size (JN_BM_BAND overrides): any height, the same bits."""
    monkeypatch.setenv("JN_BM_BAND", band)
    L, R = oracle.synth_pair(320, 200, 40, 9)
    for sub in (0, 1):
        out, _, _ = run(jn, jn.Bm.parameters(num_disparities=48, subpixel=sub), L[None], R[None])
        assert np.array_equal(out[0], bm.process(bm.params(48, subpixel=sub), L, R))


def test_bm_on_other_scenes_and_random_images(jn, bm):
    from scenes import make_scene
    W, H, D = 320, 240, 64
    pairs = [make_scene(k, W, H, 60, 5 + i) for i, k in enumerate(["strips", "patches", "slanted", "photometric", "blobs"])]
    rng = np.random.default_rng(9)
    pairs.append((rng.integers(0, 256, (H, W)).astype(np.uint8), rng.integers(0, 256, (H, W)).astype(np.uint8)))   # no structure at all
    pairs.append((np.full((H, W), 77, np.uint8), np.full((H, W), 77, np.uint8)))                                    # flat: every cost ties
    Ls = np.stack([p[0] for p in pairs]); Rs = np.stack([p[1] for p in pairs])
    for sub in (0, 1):
        out, _, _ = run(jn, jn.Bm.parameters(num_disparities=D, subpixel=sub), Ls, Rs)
        for b in range(len(pairs)):
            assert np.array_equal(out[b], bm.process(bm.params(D, subpixel=sub), Ls[b], Rs[b])), (sub, b)


def test_bm_committed_hashes(jn, oracle):
    rows = [l.split() for l in open(os.path.join(ROOT, "tests", "golden", "bm_hashes.txt")) if not l.startswith("#")]
    for W, H, scene, D, r, sub, seed, h in rows:
        L, R = oracle.synth_pair(int(W), int(H), int(scene), int(seed))
        out, _, _ = run(jn, jn.Bm.parameters(num_disparities=int(D), block_radius=int(r), subpixel=int(sub)), L[None], R[None])
        assert "%016x" % oracle.fnv(out[0].view(np.uint32)) == h, (W, H, sub)


@pytest.mark.parametrize("sub", [0, 1])
def test_bm_process_scan_equals_the_separate_calls(jn, bm, sgm, oracle, sub):
    """jn_bm_process_scan (matcher, u8 map and LUT scan on one stream, one synchronisation) against the oracle chain, batch of 3."""
    from jackal_navigation_amd.device import DeviceArray
    from jackal_navigation_amd import node
    W, H, D, n = 320, 180, 64, 3
    Ls = np.stack([oracle.synth_pair(W, H, 48, 40 + b)[0] for b in range(n)]); Rs = np.stack([oracle.synth_pair(W, H, 48, 40 + b)[1] for b in range(n)])
    sp, spo = node.scan_params(W, H), oracle.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
    dD = DeviceArray((n, H, W), np.int16); du8 = DeviceArray((n, H, W), np.uint8)
    bins = DeviceArray((n, sp.bins), np.float64); meta = DeviceArray((n, 4), np.float64)
    with jn.Bm(jn.Bm.parameters(num_disparities=D, subpixel=sub), W, H, max_batch=n) as m:
        for _ in range(2):                                        # twice: the scan's scratch is reused
            m.process_scan(n, dL.ptr, dR.ptr, W, H * W, dD.ptr, sp, lut.ptr, du8.ptr, bins.ptr, meta.ptr)
    luto = oracle.valid_lut(spo, W, H)
    for b in range(n):
        exp = bm.process(bm.params(D, subpixel=sub), Ls[b], Rs[b])
        u8o = sgm.to_u8(exp, sub)
        assert np.array_equal(dD.numpy()[b], exp) and np.array_equal(du8.numpy()[b], u8o)
        bo, mo, used = oracle.scan(spo, u8o, luto)
        assert used > 0 and np.allclose(bins.numpy()[b], bo, rtol=0, atol=1e-4) and np.allclose(meta.numpy()[b], mo, rtol=0, atol=1e-4)


def test_bm_feeds_the_node_tail(jn, bm, sgm, oracle):
    """Block-matching disparity -> u8 depth map -> the same LUT scan the ELAS path uses (jn_obstacle_scan), against the oracle chain."""
    from jackal_navigation_amd.device import DeviceArray
    from jackal_navigation_amd import node
    W, H, D = 320, 180, 64
    L, R = oracle.synth_pair(W, H, 48, 31)
    out, u8, _ = run(jn, jn.Bm.parameters(num_disparities=D), L[None], R[None])
    sp, spo = node.scan_params(W, H), oracle.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    du8 = DeviceArray.from_numpy(u8)
    bins = DeviceArray((1, sp.bins), np.float64); meta = DeviceArray((1, 4), np.float64)
    node.obstacle_scan(sp, 1, du8.ptr, lut.ptr, W, H, bins.ptr, meta.ptr)
    bo, mo, used = oracle.scan(spo, sgm.to_u8(bm.process(bm.params(D), L, R), 0), oracle.valid_lut(spo, W, H))
    assert used > 0 and np.allclose(bins.numpy()[0], bo, rtol=0, atol=1e-4) and np.allclose(meta.numpy()[0], mo, rtol=0, atol=1e-4)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode,extra,expected", [
    ("bm", ["--block-radius", "4"], "60300a7da785c9e7"),
    ("sgm", [], "9fc63c10dfa5ad4c"),
])
def test_bench_modes_on_baseline_config_2(mode, extra, expected):
    """`bench.py --mode bm|sgm` on BASELINE config 2's shape (640x480, D=64, batch 1): one JSON line with the contract's
    fields, and its self-check ties the timed output to the mode's golden hash."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", mode, "--width", "640", "--height", "480", "--disp", "64", "--batch", "1",
           "--steps", "5", "--warmup", "1", "--min-time", "0", "--no-cpu-baseline"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=500, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["metric"] == "stereo_pairs_per_sec" and j["n_gpus"] == 1 and j["value"] > 0 and j["config"]["mode"] == mode
    assert j["check"]["ok"] is True and j["check"]["expected"] == expected
    assert j["roofline"]["bound"] == "hbm" and j["roofline"]["ms_per_launch"] > 0 and 0 < j["roofline"]["frac"] < 1


# ---- JN_BM_COST_SSD: the banded int8 contraction on the matrix cores (csrc/bm_mfma.hip; BASELINE config 5's "int8 cost volume (CDNA4 MFMA path)") ----
@pytest.mark.parametrize("W,H,D,scene,n,kw", [
    (640, 480, 64, 64, 2, {}),                                     # 640x480 D=64
    (1280, 720, 128, 128, 1, {"subpixel": 1}),                     # config 3's frame and range, 1/16 pixel
    (320, 180, 256, 48, 2, {"subpixel": 1, "block_radius": 3}),    # D = 256 (9 tiles of candidates), 7x7
    (333, 101, 64, 30, 3, {"block_radius": 2, "prefilter_cap": 15}),   # ragged size (partial tile of 32 columns, partial band), 5x5
    (200, 150, 128, 90, 2, {"lr_max_diff": -1}),                   # no L/R check (no right-referenced pass)
    (96, 64, 128, 20, 1, {"lr_max_diff": 0, "subpixel": 1}),       # image narrower than the disparity range
    (70, 9, 32, 5, 2, {"block_radius": 4}),                        # fewer rows than the block is high, smallest range (two tiles)
    (640, 480, 96, 64, 1, {"block_radius": 3, "subpixel": 1}),     # an even number of tiles
])
def test_bm_ssd_on_the_matrix_cores_bit_exact_vs_its_definition(jn, bm, sgm, oracle, W, H, D, scene, n, kw):
    Ls = np.stack([oracle.synth_pair(W, H, scene, 700 + b)[0] for b in range(n)])
    Rs = np.stack([oracle.synth_pair(W, H, scene, 700 + b)[1] for b in range(n)])
    out, u8, t = run(jn, jn.Bm.parameters(num_disparities=D, cost_function=1, **kw), Ls, Rs)
    po = bm.params(D, cost_function=1, **kw)
    for b in range(n):
        exp = bm.process(po, Ls[b], Rs[b])
        assert np.array_equal(out[b], exp), (b, int((out[b] != exp).sum()), np.argwhere(out[b] != exp)[:5])
        assert np.array_equal(u8[b], sgm.to_u8(exp, kw.get("subpixel", 0)))
    assert t["match"] > 0 and t["total"] >= t["match"]


@pytest.mark.parametrize("band", ["1", "7", "16", "37", "200"])
def test_bm_ssd_band_heights_give_the_same_map(jn, hooks, bm, oracle, monkeypatch, band):
    """rows per wave (JN_BMQ_BAND overrides the launch's choice): the running sums restart per band, the bits must not change"""
    monkeypatch.setenv("JN_BMQ_BAND", band)
    L, R = oracle.synth_pair(320, 200, 40, 9)
    for sub in (0, 1):
        out, _, _ = run(jn, jn.Bm.parameters(num_disparities=64, subpixel=sub, cost_function=1), L[None], R[None])
        assert np.array_equal(out[0], bm.process(bm.params(64, subpixel=sub, cost_function=1), L, R))


def test_bm_ssd_on_other_scenes_and_random_images(jn, bm):
    from scenes import make_scene
    W, H, D = 320, 240, 64
    pairs = [make_scene(k, W, H, 60, 5 + i) for i, k in enumerate(["strips", "patches", "slanted", "photometric", "blobs"])]
    rng = np.random.default_rng(9)
    pairs.append((rng.integers(0, 256, (H, W)).astype(np.uint8), rng.integers(0, 256, (H, W)).astype(np.uint8)))   # no structure at all
    pairs.append((np.full((H, W), 77, np.uint8), np.full((H, W), 77, np.uint8)))                                    # flat: every cost ties (smallest d must win)
    Ls = np.stack([p[0] for p in pairs]); Rs = np.stack([p[1] for p in pairs])
    for sub in (0, 1):
        out, _, _ = run(jn, jn.Bm.parameters(num_disparities=D, subpixel=sub, cost_function=1), Ls, Rs)
        for b in range(len(pairs)):
            assert np.array_equal(out[b], bm.process(bm.params(D, subpixel=sub, cost_function=1), Ls[b], Rs[b])), (sub, b)


def test_bm_ssd_committed_hashes_incl_1080p_d256(jn, oracle):
    """tests/golden/bm_ssd_hashes.txt (oracle/bm_oracle.cpp with cost_function = 1, generator committed): 640x480 D=64 and config 5's
    frame 1920x1080 D=256 with the sub-pixel option — a size the scalar definition takes too long for inside the GPU suite."""
    rows = [l.split() for l in open(os.path.join(ROOT, "tests", "golden", "bm_ssd_hashes.txt")) if not l.startswith("#")]
    assert len(rows) >= 3
    for W, H, scene, D, r, sub, seed, h in rows:
        L, R = oracle.synth_pair(int(W), int(H), int(scene), int(seed))
        out, _, _ = run(jn, jn.Bm.parameters(num_disparities=int(D), block_radius=int(r), subpixel=int(sub), cost_function=1), L[None], R[None])
        assert "%016x" % oracle.fnv(out[0].view(np.uint32)) == h, (W, H, D, sub)


def test_bm_ssd_refuses_ranges_it_does_not_tile(jn):
    from jackal_navigation_amd import _lib
    with pytest.raises(_lib.JnError) as e:
        jn.Bm(jn.Bm.parameters(num_disparities=72, cost_function=1), 320, 240)
    assert e.value.status == _lib.JN_ERR_UNSUPPORTED
    with pytest.raises(_lib.JnError):
        jn.Bm(jn.Bm.parameters(num_disparities=64, cost_function=7), 320, 240)


@pytest.mark.gpu
@pytest.mark.parametrize("cost", [0, 1])
def test_bm_pipelined_slots_equal_the_synchronous_call(jn, oracle, cost):
    """jn_bm_submit_scan / jn_bm_wait: four batches of different frames in flight on four slots (slots 1-3 allocate their own scratch), with the
    node's tail on the slot's stream — disparities, u8 maps and scans must equal what jn_bm_process_scan gives for the same frames, twice over
    (the slots are reused); both cost functions; a second submit on a busy slot is refused."""
    from jackal_navigation_amd.device import DeviceArray
    from jackal_navigation_amd import node, _lib
    W, H, D, n, S = 320, 180, 64, 2, 4
    sp = node.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    frames = [[oracle.synth_pair(W, H, 48, 700 + 10 * k + t) for t in range(n)] for k in range(2 * S)]
    dL = [DeviceArray.from_numpy(np.stack([f[0] for f in fs])) for fs in frames]
    dR = [DeviceArray.from_numpy(np.stack([f[1] for f in fs])) for fs in frames]
    with jn.Bm(jn.Bm.parameters(num_disparities=D, subpixel=1, cost_function=cost), W, H, max_batch=n) as m:
        want = []
        for k in range(2 * S):
            dd = DeviceArray((n, H, W), np.int16); du = DeviceArray((n, H, W), np.uint8)
            bins = DeviceArray((n, sp.bins), np.float64); meta = DeviceArray((n, 4), np.float64)
            m.process_scan(n, dL[k].ptr, dR[k].ptr, W, H * W, dd.ptr, sp, lut.ptr, du.ptr, bins.ptr, meta.ptr)
            want.append((dd.numpy().copy(), du.numpy().copy(), bins.numpy().copy(), meta.numpy().copy()))
        outs = [dict(dd=DeviceArray((n, H, W), np.int16), du=DeviceArray((n, H, W), np.uint8), bins=DeviceArray((n, sp.bins), np.float64),
                     meta=DeviceArray((n, 4), np.float64)) for _ in range(S)]
        got = [None] * (2 * S)
        for k in range(2 * S):
            s = k % S
            if k >= S:
                m.wait(s)
                got[k - S] = tuple(outs[s][x].numpy().copy() for x in ("dd", "du", "bins", "meta"))
            o = outs[s]
            m.submit_scan(s, n, dL[k].ptr, dR[k].ptr, W, H * W, o["dd"].ptr, sp, lut.ptr, o["du"].ptr, o["bins"].ptr, o["meta"].ptr)
            if k == 0:
                with pytest.raises(_lib.JnError):               # one batch per slot
                    m.submit_scan(s, n, dL[k].ptr, dR[k].ptr, W, H * W, o["dd"].ptr, sp, lut.ptr, o["du"].ptr, o["bins"].ptr, o["meta"].ptr)
        for k in range(S, 2 * S):
            s = k % S
            m.wait(s)
            got[k] = tuple(outs[s][x].numpy().copy() for x in ("dd", "du", "bins", "meta"))
        for k in range(2 * S):
            for a, b in zip(got[k], want[k]):
                assert np.array_equal(a, b), (cost, k)
