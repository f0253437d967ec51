"""the scan consumer (navigate.cpp) — product host code against the oracle restatement,
on sequences of scans, so that a LaserScan from this library provably drives the reference's stop / turn logic
the same way.  CPU only."""
import numpy as np
import pytest


def random_scan(rng, kind):
    n = int(rng.integers(0, 91))
    if kind == "far":
        ranges = rng.uniform(1.5, 6.0, n)
    elif kind == "near":
        ranges = rng.uniform(0.2, 1.2, n)
    else:
        ranges = rng.uniform(0.3, 4.0, n)
    lo = float(rng.uniform(-0.8, -0.1)); hi = float(rng.uniform(0.1, 0.8))
    return {"ranges": ranges.astype(np.float32), "angle_min": np.float32(lo), "angle_max": np.float32(hi)}


def oracle_step(oracle, msg, history, last_dir):
    """obstacleAvoidMode's decision (navigate.cpp:229-256) assembled from the oracle's three functions."""
    xy = oracle.scan_to_points(msg["ranges"], msg["angle_min"], msg["angle_max"])
    obst, count, closest, conf = oracle.check_obstacle(xy, history)
    direction = oracle.choose_direction(xy, last_dir) if obst else 0
    return xy, obst, count, closest, conf, direction


def test_points_bit_identical(jn, oracle, same):
    rng = np.random.default_rng(4)
    nav = jn.navigate.Navigator()
    for _ in range(50):
        msg = random_scan(rng, "mixed")
        assert same(nav.scan_callback(msg), oracle.scan_to_points(msg["ranges"], msg["angle_min"], msg["angle_max"]))


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_decisions_follow_the_reference_sequence(jn, oracle, seed):
    """300 ticks of mixed traffic: spatial vote, 0.5 m stop rule, 20-vote history, turn hysteresis."""
    rng = np.random.default_rng(seed)
    nav = jn.navigate.Navigator()
    history, last_dir = [], 0
    seen = set()
    for tick in range(300):
        kind = ("far", "near", "mixed")[int(rng.integers(0, 3))] if tick % 40 < 25 else "far"
        msg = random_scan(rng, kind)
        nav.scan_callback(msg)
        d = nav.obstacle_avoid_step()
        _, obst, count, closest, conf, direction = oracle_step(oracle, msg, history, last_dir)
        last_dir = direction
        assert (d["obstacle"], d["points_inside"], d["direction"], d["points"]) == (obst, count, direction, len(msg["ranges"])), tick
        assert d["closest"] == closest and d["confidence"] == conf, tick
        seen.add((obst, direction))
    assert {(0, 0), (1, 1), (1, 2)} <= seen        # the sequence exercised stop, left and right turns


def test_empty_scan_and_bad_arguments(jn):
    import ctypes as C
    from jackal_navigation_amd import _lib
    nav = jn.navigate.Navigator()
    nav.scan_callback({"ranges": np.zeros(0, np.float32), "angle_min": np.float32(400.), "angle_max": np.float32(-400.)})
    d = nav.obstacle_avoid_step()
    assert d["obstacle"] == 0 and d["points"] == 0 and d["closest"] == 1e9 and d["confidence"] == 0.0
    p = _lib.NavParams(); L = jn.load(); L.jn_nav_params_default(C.byref(p))
    assert (p.clear_front, p.clear_side, p.laser_pt_thresh, p.history) == (0.24 + 0.8, 0.3, 8, 20)     # navigate.cpp:37-42
    p.history = 1000
    s = _lib.NavState(); out = _lib.NavDecision()
    assert L.jn_nav_vote(C.byref(p), C.byref(s), None, 0, C.byref(out)) == _lib.JN_ERR_INVALID
