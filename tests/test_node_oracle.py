"""Properties of the node-side restatement (point_cloud.cpp functions after Elas::process).
That file cannot be compiled here (ROS/OpenCV), so these pin the restated semantics."""
import numpy as np


def test_convert_to_u8_round_half_even_and_saturate(oracle):
    D = np.array([-10, -1, -0.5, 0, 0.5, 1.5, 2.5, 2.4999, 3.5, 254.5, 255.5, 300, 1e9], np.float32)
    assert oracle.to_u8(D).tolist() == [0, 0, 0, 0, 0, 2, 2, 2, 4, 254, 255, 255, 255]      # point_cloud.cpp:422


def test_compaction_order_and_threshold(oracle):
    bins = np.full(90, 1e9)
    bins[3] = 2.5; bins[50] = 1.25; bins[89] = 7.0; bins[10] = 1e9 - 0.5                       # last one counts as a return (< INF-1 fails)
    assert oracle.compact(bins).tolist() == [7.0, 1.25, 2.5]                                   # pushed from bin 89 down (:278-282)


def test_lut_is_monotone_in_rows_and_wraps(oracle):
    W, H = 160, 90
    sp = oracle.scan_params(W, H)
    lut = oracle.valid_lut(sp, W, H)
    assert (lut[:, :, 1] == 255).all()
    lo = lut[:, :, 0].astype(int)
    # rows far below the horizon only see ground until very large disparities (or never: 256 wraps to 0)
    assert lo[10, W // 2] == 3 and (lo[H - 1] >= lo[H // 2 + 5]).all() | (lo[H - 1] == 0).any()


def test_scan_of_a_fronto_parallel_wall(oracle):
    W, H = 320, 180
    sp = oracle.scan_params(W, H)
    lut = oracle.valid_lut(sp, W, H)
    disp = np.zeros((H, W), np.uint8)
    disp[40:100, :] = 20                             # wall at Z = f*B/d
    bins, meta, used = oracle.scan(sp, disp, lut)
    f = sp.Q[11]; B = 1.0 / sp.Q[14]
    z = f * B / 20.0
    hit = bins[bins < 1e9 - 1]
    assert used > 0 and len(hit) > 10
    assert np.all(hit >= z * 0.95) and np.all(hit <= z * 1.6)       # range = sqrt(X^2+Y^2) >= forward distance
    assert meta[0] < 0 < meta[1] and abs(meta[2] - hit.min()) < 1e-9


def test_point_cloud_order_and_filter(oracle):
    W, H = 64, 48
    sp = oracle.scan_params(W, H)
    disp = np.zeros((H, W), np.uint8)
    disp[5, 7] = 30; disp[20, 7] = 10; disp[3, 9] = 1; disp[4, 9] = 2
    pc = oracle.point_cloud(sp, disp)
    assert pc.shape == (3, 3)                        # d<2 dropped (point_cloud.cpp:324); order: column 7 rows 5,20 then column 9
    assert pc[0, 0] < pc[1, 0]                       # larger disparity = closer
    bins, meta, used = oracle.scan_points(sp, pc.astype(np.float64))
    assert used <= 3
