"""SGM mode on the GPU: bit-exact against its scalar definition (oracle/sgm_oracle.cpp; self-referential — the
reference has no SGM), through the C-ABI (jn_sgm_*)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sgm():
    from oracle.binding import SgmOracle
    return SgmOracle()


def run(jn, p, Ls, Rs):
    from jackal_navigation_amd.device import DeviceArray
    n, H, W = Ls.shape
    dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
    dD = DeviceArray((n, H, W), np.int16)
    with jn.Sgm(p, W, H, max_batch=n) as s:
        s.process_batch(n, dL.ptr, dR.ptr, W, H * W, dD.ptr)
        t = s.last_times()
        du8 = DeviceArray((n, H, W), np.uint8)
        s.to_u8(dD.ptr, du8.ptr, n * H * W)
    out, u8 = dD.numpy(), du8.numpy()
    for a in (dL, dR, dD, du8):
        a.free()
    return out, u8, t


@pytest.mark.parametrize("W,H,D,scene,n,kw", [
    (640, 480, 64, 64, 2, {}),                      # BASELINE config 2's frame and range
    (1280, 720, 128, 128, 1, {}),                   # BASELINE config 3's frame and range
    (320, 180, 256, 48, 2, {"subpixel": 1}),        # D = 256 (four disparities per lane), 1/16 pixel
    (333, 101, 64, 30, 3, {"subpixel": 1, "P1": 4, "P2": 30, "prefilter_cap": 15}),   # ragged size, other penalties
    (200, 150, 128, 90, 2, {"lr_max_diff": -1}),    # no L/R check; disparities beyond the image width near the left border
    (96, 64, 128, 20, 1, {"lr_max_diff": 0}),       # image narrower than the disparity range
])
def test_sgm_bit_exact_vs_its_definition(jn, sgm, oracle, W, H, D, scene, n, kw):
    Ls = np.stack([oracle.synth_pair(W, H, scene, 700 + b)[0] for b in range(n)])
    Rs = np.stack([oracle.synth_pair(W, H, scene, 700 + b)[1] for b in range(n)])
    out, u8, t = run(jn, jn.Sgm.parameters(num_disparities=D, **kw), Ls, Rs)
    po = sgm.params(D, **{k: v for k, v in kw.items()})
    for b in range(n):
        exp = sgm.process(po, Ls[b], Rs[b])
        assert np.array_equal(out[b], exp), (b, int((out[b] != exp).sum()))
        assert np.array_equal(u8[b], sgm.to_u8(exp, kw.get("subpixel", 0)))
    assert t["paths"] > 0 and t["total"] >= t["paths"]


def test_sgm_on_other_scenes_and_random_images(jn, sgm):
    from scenes import make_scene
    W, H, D = 320, 240, 64
    pairs = [make_scene(k, W, H, 60, 5 + i) for i, k in enumerate(["strips", "patches", "slanted", "photometric", "blobs"])]
    rng = np.random.default_rng(9)
    pairs.append((rng.integers(0, 256, (H, W)).astype(np.uint8), rng.integers(0, 256, (H, W)).astype(np.uint8)))   # no structure at all
    pairs.append((np.full((H, W), 77, np.uint8), np.full((H, W), 77, np.uint8)))                                    # flat: every cost ties
    Ls = np.stack([p[0] for p in pairs]); Rs = np.stack([p[1] for p in pairs])
    for sub in (0, 1):
        out, _, _ = run(jn, jn.Sgm.parameters(num_disparities=D, subpixel=sub), Ls, Rs)
        for b in range(len(pairs)):
            assert np.array_equal(out[b], sgm.process(sgm.params(D, subpixel=sub), Ls[b], Rs[b])), (sub, b)


def test_sgm_feeds_the_node_tail(jn, sgm, oracle):
    """SGM disparity -> u8 depth map -> the same LUT scan the ELAS path uses (jn_obstacle_scan), against the oracle chain."""
    from jackal_navigation_amd.device import DeviceArray
    from jackal_navigation_amd import node
    W, H, D = 320, 180, 64
    L, R = oracle.synth_pair(W, H, 48, 31)
    out, u8, _ = run(jn, jn.Sgm.parameters(num_disparities=D), L[None], R[None])
    sp, spo = node.scan_params(W, H), oracle.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    du8 = DeviceArray.from_numpy(u8)
    bins = DeviceArray((1, sp.bins), np.float64); meta = DeviceArray((1, 4), np.float64)
    node.obstacle_scan(sp, 1, du8.ptr, lut.ptr, W, H, bins.ptr, meta.ptr)
    bo, mo, used = oracle.scan(spo, sgm.to_u8(sgm.process(sgm.params(D), L, R), 0), oracle.valid_lut(spo, W, H))
    assert used > 0 and np.allclose(bins.numpy()[0], bo, rtol=0, atol=1e-4) and np.allclose(meta.numpy()[0], mo, rtol=0, atol=1e-4)


def test_sgm_config3_batch32_720p(jn, sgm, oracle):
    """BASELINE config 3 as it is named: 1280x720, D = 128, SGM 8 paths, batch 32 in one call.  Frames 0 and 17 against the scalar
    definition, frame 0 also against the committed hash, every frame against a second run (the block pipeline must not depend on
    which workgroup started first)."""
    import os
    W, H, D, n = 1280, 720, 128, 32
    pairs = [oracle.synth_pair(W, H, 128, 12345 + b) for b in range(n)]
    Ls = np.stack([p[0] for p in pairs]); Rs = np.stack([p[1] for p in pairs])
    out, u8, t = run(jn, jn.Sgm.parameters(num_disparities=D), Ls, Rs)
    for b in (0, 17):
        exp = sgm.process(sgm.params(D), Ls[b], Rs[b])
        assert np.array_equal(out[b], exp), (b, int((out[b] != exp).sum()))
    rows = [l.split() for l in open(os.path.join(os.path.dirname(__file__), "golden", "sgm_hashes.txt")) if not l.startswith("#")]
    want = [r[6] for r in rows if r[:6] == ["1280", "720", "128", "128", "0", "12345"]][0]
    assert "%016x" % oracle.fnv(np.ascontiguousarray(out[0]).view(np.uint32)) == want
    again, _, _ = run(jn, jn.Sgm.parameters(num_disparities=D), Ls, Rs)
    assert np.array_equal(out, again)
    assert len({out[b].tobytes() for b in range(n)}) == n          # 32 different frames in, 32 different maps out


def test_sgm_config5_share_1080p_d256_subpixel(jn, sgm, oracle):
    """BASELINE config 5's per-GPU share: 1920x1080, D = 256, SGM + 1/16-pixel refinement, 8 pairs per GPU.  Frame 3 against the
    scalar definition; all 8 frames twice (determinism)."""
    W, H, D, n = 1920, 1080, 256, 8
    pairs = [oracle.synth_pair(W, H, 256, 500 + b) for b in range(n)]
    Ls = np.stack([p[0] for p in pairs]); Rs = np.stack([p[1] for p in pairs])
    p = jn.Sgm.parameters(num_disparities=D, subpixel=1)
    out, u8, t = run(jn, p, Ls, Rs)
    exp = sgm.process(sgm.params(D, subpixel=1), Ls[3], Rs[3])
    assert np.array_equal(out[3], exp), int((out[3] != exp).sum())
    assert np.array_equal(u8[3], sgm.to_u8(exp, 1))
    again, _, _ = run(jn, p, Ls, Rs)
    assert np.array_equal(out, again)


@pytest.mark.gpu
def test_sgm_launch_tag_wraps_around(jn, hooks, sgm, oracle, monkeypatch):
    """The columns handed from block to block carry a 16-bit launch tag (sgm_sweep.hip); when it wraps the buffer is zeroed.  Start a
    handle three launches before the wrap, run a larger batch, then smaller ones across the wrap, then the larger one again."""
    from jackal_navigation_amd.device import DeviceArray
    monkeypatch.setenv("JN_SGM_EPOCH_START", str(0xFFFF - 3))
    W, H, D = 300, 90, 64
    pairs = [oracle.synth_pair(W, H, 40, 900 + b) for b in range(3)]
    Ls, Rs = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    exp = [sgm.process(sgm.params(D), Ls[b], Rs[b]) for b in range(3)]
    dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
    dD = DeviceArray((3, H, W), np.int16)
    with jn.Sgm(jn.Sgm.parameters(num_disparities=D), W, H, max_batch=3) as s:
        for n in (3, 1, 1, 1, 2, 3):
            s.process_batch(n, dL.ptr, dR.ptr, W, H * W, dD.ptr)
            out = dD.numpy()
            for b in range(n):
                assert np.array_equal(out[b], exp[b]), (n, b)
    for a in (dL, dR, dD):
        a.free()


@pytest.mark.gpu
def test_sgm_long_chain_of_blocks(jn, sgm, oracle):
    """A frame of 3840 columns: 61 blocks hand their columns from one to the next through memory (tagged dwords), each waiting only for
    its producer; rows few enough for the scalar definition to finish in seconds."""
    W, H, D = 3840, 72, 128
    L, R = oracle.synth_pair(W, H, 100, 31)
    exp = sgm.process(sgm.params(D), L, R)
    out, _, _ = run(jn, jn.Sgm.parameters(num_disparities=D), np.stack([L, L, L]), np.stack([R, R, R]))
    for b in range(3):
        assert np.array_equal(out[b], exp), b


@pytest.mark.parametrize("W,H,sub,lr", [(320, 180, 1, None), (333, 187, 0, -1), (270, 161, 0, 2)])
def test_sgm_pipelined_slots_equal_the_synchronous_call(jn, oracle, W, H, sub, lr):
    """jn_sgm_submit_scan / jn_sgm_wait: four batches of different frames in flight on four slots (slots 1-3 allocate their own volumes), with
    the node's tail on the slot's stream — disparities, u8 maps and scans must equal what the synchronous calls give for the same frames,
    twice over (the slots are reused), and a second submit on a busy slot is refused.  The pipelined form's tail is ONE kernel (L/R check,
    int16 map, mono8 map and scan: k_scan<false, true>), the synchronous route's is k_sw_lr, k_sgm_to_u8 and k_scan one after the other:
    with and without the sub-pixel step, with the L/R check off, widths that are no multiple of anything."""
    from jackal_navigation_amd.device import DeviceArray
    from jackal_navigation_amd import node, _lib
    D, n, S = 64, 2, 4
    sp = node.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    frames = [[oracle.synth_pair(W, H, 48, 300 + 10 * k + t) for t in range(n)] for k in range(2 * S)]
    dL = [DeviceArray.from_numpy(np.stack([f[0] for f in fs])) for fs in frames]
    dR = [DeviceArray.from_numpy(np.stack([f[1] for f in fs])) for fs in frames]
    kw = dict(num_disparities=D, subpixel=sub)
    if lr is not None:
        kw["lr_max_diff"] = lr
    with jn.Sgm(jn.Sgm.parameters(**kw), W, H, max_batch=n) as m:
        want = []
        for k in range(2 * S):                                 # the synchronous route: three calls per batch
            dd = DeviceArray((n, H, W), np.int16); du = DeviceArray((n, H, W), np.uint8)
            bins = DeviceArray((n, sp.bins), np.float64); meta = DeviceArray((n, 4), np.float64)
            m.process_batch(n, dL[k].ptr, dR[k].ptr, W, H * W, dd.ptr)
            m.to_u8(dd.ptr, du.ptr, n * H * W)
            node.obstacle_scan(sp, n, du.ptr, lut.ptr, W, H, bins.ptr, meta.ptr)
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
                    m.submit_scan(0, n, dL[k].ptr, dR[k].ptr, W, H * W, o["dd"].ptr)
        for k in range(S, 2 * S):
            m.wait(k % S)
            got[k] = tuple(outs[k % S][x].numpy().copy() for x in ("dd", "du", "bins", "meta"))
    for k in range(2 * S):
        for a, b in zip(want[k], got[k]):
            assert np.array_equal(a, b), k
        assert (want[k][0] >= 0).mean() > 0.5


def test_sgm_wait_covers_the_scan_tail_at_full_size(jn, oracle):
    """jn_sgm_wait returns when the WHOLE batch is complete — the u8 map and the LUT scan queued behind the sweeps included (ADVICE r04:
    it used to wait for the event behind the L/R kernel only, and the slot streams are non-blocking, so a copy right after the wait could
    read bins that were still being written).  1280x720, batch 8: the tail is long enough to lose that race; the outputs are read straight
    after the wait and must equal the synchronous route's.  Also: the synchronous call refuses to run on slot 0 while it carries a batch."""
    from jackal_navigation_amd.device import DeviceArray
    from jackal_navigation_amd import node, _lib
    W, H, D, n = 1280, 720, 128, 8
    sp = node.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    pairs = [oracle.synth_pair(W, H, 100, 4000 + b) for b in range(n)]
    dL = DeviceArray.from_numpy(np.stack([p[0] for p in pairs])); dR = DeviceArray.from_numpy(np.stack([p[1] for p in pairs]))
    with jn.Sgm(jn.Sgm.parameters(num_disparities=D), W, H, max_batch=n) as m:
        dd = DeviceArray((n, H, W), np.int16); du = DeviceArray((n, H, W), np.uint8)
        bins = DeviceArray((n, sp.bins), np.float64); meta = DeviceArray((n, 4), np.float64)
        m.process_batch(n, dL.ptr, dR.ptr, W, H * W, dd.ptr)
        m.to_u8(dd.ptr, du.ptr, n * H * W)
        node.obstacle_scan(sp, n, du.ptr, lut.ptr, W, H, bins.ptr, meta.ptr)
        want = (du.numpy().copy(), bins.numpy().copy(), meta.numpy().copy())
        for slot in (0, 1, 0, 1, 0):
            du2 = DeviceArray((n, H, W), np.uint8); bins2 = DeviceArray((n, sp.bins), np.float64); meta2 = DeviceArray((n, 4), np.float64)
            bins2.fill(0) if hasattr(bins2, "fill") else None
            m.submit_scan(slot, n, dL.ptr, dR.ptr, W, H * W, dd.ptr, sp, lut.ptr, du2.ptr, bins2.ptr, meta2.ptr)
            if slot == 0:
                with pytest.raises(_lib.JnError):
                    m.process_batch(n, dL.ptr, dR.ptr, W, H * W, dd.ptr)
            m.wait(slot)
            got = (du2.numpy(), bins2.numpy(), meta2.numpy())       # plain hipMemcpy: not ordered behind the slot's non-blocking stream
            for a, b in zip(want, got):
                assert np.array_equal(a, b), slot
            for a in (du2, bins2, meta2):
                a.free()
