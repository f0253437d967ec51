"""Rectification front end: stereoRectify / initUndistortRectifyMap / remap.
The arithmetic lives in OpenCV, which is neither in the reference tree nor installed: parity is
UNPINNED, so these tests check geometry (CPU) and GPU-vs-oracle agreement (gpu)."""
import numpy as np
import pytest


def _mats(r):
    return (np.array(r.R1).reshape(3, 3), np.array(r.R2).reshape(3, 3), np.array(r.P1).reshape(3, 4), np.array(r.P2).reshape(3, 4),
            np.array(r.Q).reshape(4, 4))


def _project(K, D, X):
    x, y = X[0] / X[2], X[1] / X[2]
    r2 = x * x + y * y
    kr = 1 + ((D[4] * r2 + D[1]) * r2 + D[0]) * r2
    xd = x * kr + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x)
    yd = y * kr + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
    return K[0] * xd + K[2], K[4] * yd + K[5]


def test_stereo_rectify_geometry(jn):
    from jackal_navigation_amd import node
    c = node.stereo_calib()
    for (W, H) in ((320, 180), (640, 360), (1280, 720)):
        R1, R2, P1, P2, Q = _mats(node.stereo_rectify(c, W, H))
        for R in (R1, R2):
            assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12
        # zero-disparity flag: identical principal points; horizontal rig: baseline only in P2[0,3]
        assert P1[0, 2] == P2[0, 2] and P1[1, 2] == P2[1, 2] and P1[0, 0] == P1[1, 1] == P2[0, 0]
        assert P2[0, 3] < 0 and P2[1, 3] == 0 and np.all(P1[:, 3] == 0)
        T = np.array(c.T); Rm = np.array(c.R).reshape(3, 3)
        t_rect = R2 @ T
        assert abs(t_rect[1]) < 1e-9 and abs(t_rect[2]) < 1e-9          # baseline lies on the rectified x axis
        assert abs(P2[0, 3] - P2[0, 0] * t_rect[0]) < 1e-9
        assert abs(Q[3, 2] + 1.0 / t_rect[0]) < 1e-12 and Q[2, 3] == P1[0, 0] and Q[3, 3] == 0
        # a 3-D point seen by both cameras lands on the same rectified row, disparity = f*B/Z, and Q inverts it
        for X1 in (np.array([0.3, -0.1, 2.0]), np.array([-0.5, 0.2, 4.0]), np.array([0.0, 0.0, 1.0])):
            X2 = Rm @ X1 + T
            r1, r2 = R1 @ X1, R2 @ X2
            u1, v1 = P1[0, 0] * r1[0] / r1[2] + P1[0, 2], P1[1, 1] * r1[1] / r1[2] + P1[1, 2]
            u2, v2 = P2[0, 0] * r2[0] / r2[2] + P2[0, 2], P2[1, 1] * r2[1] / r2[2] + P2[1, 2]
            assert abs(v1 - v2) < 1e-9
            d = u1 - u2
            assert abs(d - (-P2[0, 3]) / r1[2]) < 1e-9
            p = Q @ np.array([u1, v1, d, 1.0])
            assert np.allclose(p[:3] / p[3], r1, atol=1e-9)
        # alpha = 0: every pixel of the new image maps inside the source image
        K1, D1 = np.array(c.K1), np.array(c.D1)
        iR = np.linalg.inv(P1[:, :3] @ R1)
        for (u, v) in ((0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1), (W // 2, 0), (0, H // 2)):
            ray = iR @ np.array([u, v, 1.0])
            su, sv = _project(K1, D1, ray)
            assert -1.0 <= su <= c.calib_width and -1.0 <= sv <= c.calib_height, (W, H, u, v, su, sv)


def test_map_and_remap_oracle_properties(oracle):
    # identity geometry: maps are the pixel grid, remap copies
    K = [100, 0, 40, 0, 100, 30, 0, 0, 1]; D = [0] * 5; R = np.eye(3).ravel(); P = [100, 0, 40, 0, 0, 100, 30, 0, 0, 0, 1, 0]
    mx, my = oracle.undistort_map(K, D, R, P, 80, 60)
    assert np.allclose(mx, np.arange(80)[None, :], atol=1e-4) and np.allclose(my, np.arange(60)[:, None], atol=1e-4)
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (60, 80)).astype(np.uint8)
    assert np.array_equal(oracle.remap(src, mx, my), src)
    # half-pixel shift = average of neighbours (rounded to nearest), constant 0 border
    out = oracle.remap(src, mx + 0.5, my)
    exp = (src[:, :-1].astype(int) + src[:, 1:].astype(int) + 1) // 2
    assert np.array_equal(out[:, :-1], exp)
    assert np.array_equal(out[:, -1], (src[:, -1].astype(int) + 1) // 2)
    assert (oracle.remap(src, mx + 500, my) == 0).all()


@pytest.mark.gpu
def test_maps_and_remap_gpu_vs_oracle(jn, oracle, same):
    from jackal_navigation_amd import node
    from jackal_navigation_amd.device import DeviceArray
    c = node.stereo_calib()
    W, H = 320, 180                                   # rawimsize of the reference node (point_cloud.cpp:49-50, :540)
    r = node.stereo_rectify(c, W, H)
    rng = np.random.default_rng(5)
    n = 3
    src = rng.integers(0, 256, (n, c.calib_height, c.calib_width)).astype(np.uint8)
    dsrc = DeviceArray.from_numpy(src)
    for K, D, R, P in ((c.K1, c.D1, r.R1, r.P1), (c.K2, c.D2, r.R2, r.P2)):
        mx, my = node.init_undistort_rectify_map(list(K), list(D), list(R), list(P), W, H)
        mxo, myo = oracle.undistort_map(list(K), list(D), list(R), list(P), W, H)
        gx, gy = mx.numpy(), my.numpy()
        assert np.abs(gx - mxo).max() <= 2e-4 and np.abs(gy - myo).max() <= 2e-4     # same doubles up to the last float ulp
        ddst = DeviceArray((n, H, W), np.uint8)
        node.remap(n, dsrc.ptr, c.calib_width, c.calib_height, c.calib_width, c.calib_width * c.calib_height, mx.ptr, my.ptr,
                   ddst.ptr, W, H, W, W * H)
        out = ddst.numpy()
        for b in range(n):
            assert same(out[b], oracle.remap(src[b], gx, gy))           # integer arithmetic: bit-exact given the same maps
        assert out.mean() > 60                                          # alpha=0: (almost) no black border


@pytest.mark.gpu
def test_rectify_then_elas_end_to_end(jn, oracle, same):
    """raw frames -> remap (GPU) -> ELAS (GPU) equals remap (oracle) -> ELAS (oracle)."""
    from jackal_navigation_amd import node
    from jackal_navigation_amd.device import DeviceArray
    W, H = 320, 180
    c = node.stereo_calib()
    r = node.stereo_rectify(c, W, H)
    # synthetic "raw" frames: the textured synthetic pair up-sampled to the 640x360 sensor
    L, R = node.synth_pair(W, H, 40, 77)
    rawL = np.ascontiguousarray(np.kron(L, np.ones((2, 2), np.uint8))); rawR = np.ascontiguousarray(np.kron(R, np.ones((2, 2), np.uint8)))
    outs = []
    for raw, K, D, Rr, P in ((rawL, c.K1, c.D1, r.R1, r.P1), (rawR, c.K2, c.D2, r.R2, r.P2)):
        mx, my = node.init_undistort_rectify_map(list(K), list(D), list(Rr), list(P), W, H)
        d = DeviceArray.from_numpy(raw); o = DeviceArray((H, W), np.uint8)
        node.remap(1, d.ptr, 640, 360, 640, 640 * 360, mx.ptr, my.ptr, o.ptr, W, H, W, W * H)
        img = o.numpy()
        assert same(img, oracle.remap(raw, mx.numpy(), my.numpy()))
        outs.append(img)
    D1 = np.zeros((H, W), np.float32); D2 = np.zeros((H, W), np.float32)
    with jn.Elas(jn.Elas.parameters(0, disp_max=63), W, H) as e:
        st = e.process(outs[0], outs[1], D1, D2, (W, H, W))
    st_o, D1o, D2o = oracle.process(oracle.params(0, disp_max=63), outs[0], outs[1])
    assert st == st_o and same(D1, D1o) and same(D2, D2o)


def test_stereo_rectify_against_an_independent_restatement(jn):
    """a25 / f1 had no second implementation.  oracle/rectify_oracle.py restates Bouguet's rectification with other
    formulas throughout (quaternion half rotation, minimal vector-to-vector rotation, Newton-converged undistortion, float64
    points); jackal_navigation_amd/csrc/rectify.cpp follows cvStereoRectify's own steps (Rodrigues both ways, five fixed-point
    undistortion sweeps, float32 points between its calls).  Bounds: rotations agree to 1e-12 (pure double arithmetic on both
    sides); with the restatement also running cvUndistortPoints' five sweeps, focal length, principal point and Q agree to
    2e-6 relative (what is left: float32 corner points); with the distortion inverted to convergence instead they move by up to
    ~1e-3 relative (0.1 - 0.2 pixel of principal point on this lens) — OpenCV 2.4's five sweeps are part of what the node gets.
    On the reference's calibration (calibration/amrl_jackal_webcam_stereo.yml) and on perturbed rigs, horizontal and vertical."""
    from jackal_navigation_amd import node
    from oracle import rectify_oracle as ro
    rng = np.random.default_rng(11)
    base = node.stereo_calib()

    def variants():
        yield base, (320, 180)
        yield base, (640, 360)
        yield base, (0, 0)
        for k in range(6):
            c = node.stereo_calib()
            om = rng.normal(0, 0.03, 3)
            th = np.linalg.norm(om); kx = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]]) / th
            Rp = np.eye(3) + np.sin(th) * kx + (1 - np.cos(th)) * kx @ kx
            Rn = Rp @ np.array(c.R).reshape(3, 3)
            for i in range(9):
                c.R[i] = float(Rn.ravel()[i])
            T = np.array(c.T) * (1 + rng.normal(0, 0.1)) + rng.normal(0, 0.002, 3)
            if k >= 4:
                T = np.array([T[1], -abs(T[0]), T[2]])            # a vertical rig
            for i in range(3):
                c.T[i] = float(T[i])
            for i in range(5):
                c.D1[i] *= float(1 + rng.normal(0, 0.2)); c.D2[i] *= float(1 + rng.normal(0, 0.2))
            yield c, (320, 180) if k % 2 else (1280, 720)

    n = 0
    for c, size in variants():
        R1, R2, P1, P2, Q = _mats(node.stereo_rectify(c, *size))
        args = (np.array(c.K1), np.array(c.D1), np.array(c.K2), np.array(c.D2), np.array(c.R), np.array(c.T), (c.calib_width, c.calib_height), size)
        o = ro.stereo_rectify(*args, sweeps=5)
        assert np.abs(R1 - o[0]).max() < 1e-12 and np.abs(R2 - o[1]).max() < 1e-12
        for got, exp in ((P1, o[2]), (P2, o[3]), (Q, o[4])):
            scale = np.maximum(np.abs(exp), 1e-3 * np.abs(exp).max())
            assert (np.abs(got - exp) / scale).max() < 2e-6, (size, got, exp)
        oc = ro.stereo_rectify(*args)                              # distortion inverted to convergence: close, not equal
        worst = max((np.abs(got - exp) / np.maximum(np.abs(exp), 1e-3 * np.abs(exp).max())).max() for got, exp in ((P1, oc[2]), (P2, oc[3]), (Q, oc[4])))
        assert worst < 5e-3, (size, worst)
        n += 1
    assert n == 9
