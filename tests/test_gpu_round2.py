"""GPU parity cases added in round 2 (VERDICT r01 "close the parity coverage holes" + the N>1 path on hardware):
BASELINE config 5's per-GPU share through the benched entry point, the C++ drop-in shim with the reference's literal
call sequence, non-zero crop offsets, and the HIP path under two ranks with the cross-rig merge."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCAN_TOL = 1e-4


def test_config5_share_1080p_d256_batch8_through_submit_scan(jn, oracle, same):
    """1920x1080, D=256, batch 8 = what one GPU gets of BASELINE config 5 (64 pairs over 8 GPUs), through
    jn_elas_submit_scan (the entry point bench.py times): the batched big-lattice filter route (k_filter_resolve_big +
    streamed redundancy passes with n > 1).  Three frames bit-checked against the oracle incl. u8 map and scan, frame 0
    against the reference's own hash, all eight through size-free properties."""
    from jackal_navigation_amd.device import DeviceArray
    from jackal_navigation_amd import node
    W, H, n, D = 1920, 1080, 8, 256
    Ls = np.zeros((n, H, W), np.uint8); Rs = np.zeros((n, H, W), np.uint8)
    for b in range(n):
        Ls[b], Rs[b] = node.synth_pair(W, H, D, 12345 + b)
    dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
    sp, spo = node.scan_params(W, H), oracle.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    outs = []
    with jn.Elas(jn.Elas.parameters(0, disp_max=D - 1), W, H, max_batch=n, slots=2, host_threads=8) as e:
        for rep in range(2):
            d1 = DeviceArray.from_numpy(np.zeros((n, H, W), np.float32)); d2 = DeviceArray.from_numpy(np.zeros((n, H, W), np.float32))
            u8 = DeviceArray((n, H, W), np.uint8); bins = DeviceArray((n, sp.bins), np.float64); meta = DeviceArray((n, 4), np.float64)
            st = (C.c_int32 * n)()
            e.submit_scan(rep, n, dL.ptr, dR.ptr, W, H * W, d1.ptr, d2.ptr, sp, lut.ptr, u8.ptr, bins.ptr, meta.ptr, st)
            e.wait(rep)
            assert list(st) == [0] * n
            outs.append((d1.numpy(), d2.numpy(), u8.numpy(), bins.numpy(), meta.numpy()))
            for a in (d1, d2, u8, bins, meta):
                a.free()
    for a, b in zip(outs[0], outs[1]):
        assert same(a, b)                                           # two slots, two runs: identical
    D1, D2, U8, B_, M_ = outs[0]
    assert oracle.fnv(D1[0]) == 0xcd2740a7ac6afdf7                  # SURVEY §8c / tests/golden/reference_hashes.txt
    yy, xx = np.mgrid[0:H, 0:W]
    gt = (yy / H * (D * 0.6)).astype(int) + 2
    gt[(xx > W // 3) & (xx < W // 2) & (yy > H // 3) & (yy < 2 * H // 3)] = int(D * 0.7)
    for b in range(n):
        valid = D1[b] >= 0
        assert set(np.unique(D1[b][~valid]).tolist()) <= {-10.0}
        assert valid.mean() > 0.7
        assert (np.abs(D1[b][valid] - gt[valid]) <= 1.0).mean() > 0.99
        r = np.rint(D1[b]); r[r < 0] = 0; r[r > 255] = 255
        assert np.array_equal(U8[b], r.astype(np.uint8))            # convertTo(CV_8U) on every frame
    po = oracle.params(0, disp_max=D - 1)
    luto = oracle.valid_lut(spo, W, H)
    for b in (0, 3, 7):
        _, D1o, D2o = oracle.process(po, Ls[b], Rs[b])
        assert same(D1[b], D1o) and same(D2[b], D2o), b
        bo, mo, _ = oracle.scan(spo, oracle.to_u8(D1o), luto)
        assert np.array_equal(B_[b] < 1e9 - 1, bo < 1e9 - 1)
        assert np.allclose(B_[b], bo, rtol=0, atol=SCAN_TOL) and np.allclose(M_[b], mo, rtol=0, atol=SCAN_TOL)


SHIM_SRC = r'''
#include <vector>
#include <cstdio>
#include "jn_elas_shim.hpp"
int main() {
  const int W = 320, H = 180;
  std::vector<uint8_t> l(W * H), r(W * H);
  jn_synth_pair(W, H, 48, 12345u, l.data(), r.data());          // the 320x180 golden pair (SURVEY Appendix A)
  std::vector<float> leftdpf(W * H, 0.f), rightdpf(W * H, 0.f); // Mat::zeros, point_cloud.cpp:413-414
  const int32_t dims[3] = {W, H, W};                            // :415
  Elas::parameters param;                                       // :416
  param.postprocess_only_left = true;                           // :417
  for (int frame = 0; frame < 3; frame++) {                     // the node makes a fresh Elas per frame
    Elas elas(param);                                           // :418
    elas.process(l.data(), r.data(), leftdpf.data(), rightdpf.data(), dims);   // :419
  }
  std::printf("D1 %016llx D2 %016llx\n",
              (unsigned long long)jn_fnv1a64_u32(reinterpret_cast<const uint32_t*>(leftdpf.data()), W * H),
              (unsigned long long)jn_fnv1a64_u32(reinterpret_cast<const uint32_t*>(rightdpf.data()), W * H));
  return 0;
}
'''


def test_shim_runs_the_reference_call_sequence_on_the_gpu(tmp_path, jn):
    """include/jn_elas_shim.hpp with point_cloud.cpp:413-419 verbatim, as a C++ program linked against the library:
    D1/D2 of the 320x180 golden pair must hash to what the compiled reference produces."""
    src = tmp_path / "shim_gpu.cpp"
    src.write_text(SHIM_SRC)
    exe = tmp_path / "shim_gpu"
    libdir = os.path.join(ROOT, "jackal_navigation_amd")
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    os.path.join(libdir, "libjn_stereo.so"), "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "D1 5ba6eff27ab26196 D2 cf41848f68e08f83" in out.stdout, out.stdout


def test_scan_with_crop_offsets(jn, oracle, same):
    """crop_offset_x / crop_offset_y (point_cloud.cpp:51-52, used at :237-238 and in cacheDisparityValues :118) shift
    the pixel coordinates fed to Q: LUT, scan (both flavours) and point cloud with non-zero offsets."""
    from jackal_navigation_amd.device import DeviceArray
    from jackal_navigation_amd import node
    W, H, n = 320, 180, 2
    for ox, oy in ((17, 9), (-8, 30)):
        sp, spo = node.scan_params(W, H), oracle.scan_params(W, H)
        sp.crop_offset_x = spo.crop_offset_x = ox
        sp.crop_offset_y = spo.crop_offset_y = oy
        lut = node.build_valid_disp_lut(sp, W, H)
        luto = oracle.valid_lut(spo, W, H)
        assert same(lut.numpy(), luto)
        Ds = []
        for b in range(n):
            L, R = node.synth_pair(W, H, 48, 60 + b)
            _, D1, _ = oracle.process(oracle.params(0), L, R)
            Ds.append(D1)
        Ds = np.stack(Ds)
        dD = DeviceArray.from_numpy(Ds)
        du8 = DeviceArray((n, H, W), np.uint8); bins = DeviceArray((n, sp.bins), np.float64); meta = DeviceArray((n, 4), np.float64)
        node.disparity_scan(sp, n, dD.ptr, lut.ptr, W, H, du8.ptr, bins.ptr, meta.ptr)
        bins2 = DeviceArray((n, sp.bins), np.float64); meta2 = DeviceArray((n, 4), np.float64)
        node.obstacle_scan_cloud(sp, n, du8.ptr, W, H, bins2.ptr, meta2.ptr)
        u8 = du8.numpy()
        hits = 0
        for b in range(n):
            u8o = oracle.to_u8(Ds[b])
            assert same(u8[b], u8o)
            bo, mo, used = oracle.scan(spo, u8o, luto)
            hits += used
            assert np.array_equal(bins.numpy()[b] < 1e9 - 1, bo < 1e9 - 1)
            assert np.allclose(bins.numpy()[b], bo, rtol=0, atol=SCAN_TOL) and np.allclose(meta.numpy()[b], mo, rtol=0, atol=SCAN_TOL)
            bc, mc, _ = oracle.scan_cloud(spo, u8o)
            assert np.array_equal(bins2.numpy()[b] < 1e9 - 1, bc < 1e9 - 1)
            assert np.allclose(bins2.numpy()[b], bc, rtol=0, atol=SCAN_TOL) and np.allclose(meta2.numpy()[b], mc, rtol=0, atol=SCAN_TOL)
        assert hits > 0
        pc = node.point_cloud(sp, du8.ptr, W, H)
        pco = oracle.point_cloud(spo, u8[0])
        assert pc.shape == pco.shape and np.allclose(pc, pco, rtol=0, atol=SCAN_TOL)
    # the offsets matter: a scan without them differs
    sp0 = node.scan_params(W, H)
    lut0 = node.build_valid_disp_lut(sp0, W, H)
    assert not same(lut0.numpy(), luto)


def test_scan_allreduce_single_rank_communicator(jn):
    """jn_comm_* / jn_scan_allreduce with a one-rank RCCL communicator: the C-ABI merge loads RCCL, packs, reduces and
    unpacks in place (values unchanged for one rank), and reports what RCCL itself sees."""
    from jackal_navigation_amd.device import DeviceArray
    from jackal_navigation_amd import parallel
    n, nb = 5, 90
    rng = np.random.default_rng(5)
    bins = rng.uniform(0.3, 9.0, (n, nb)); bins[rng.random((n, nb)) < 0.3] = 1e9
    meta = np.stack([rng.uniform(-0.8, 0.0, n), rng.uniform(0.0, 0.8, n), rng.uniform(0.3, 1.0, n), rng.uniform(5.0, 9.0, n)], axis=1)
    dB, dM = DeviceArray.from_numpy(bins), DeviceArray.from_numpy(meta)
    comm = parallel.ScanComm(0, 1, 0, lambda raw: raw)
    assert comm.info() == (0, 1, 0)
    for _ in range(3):
        comm.merge(n, nb, dB.ptr, dM.ptr)
    assert np.array_equal(dB.numpy(), bins) and np.array_equal(dM.numpy(), meta)
    comm.close()


def _rank_worker(rank, world, port, out_dir):
    """One rank of the 2-rank HIP run (both on device 0, gloo for the merge): its rigs through jn_elas_submit_scan,
    then the path's exchange step."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import jackal_navigation_amd as jn
    from jackal_navigation_amd import node, parallel
    from jackal_navigation_amd.device import DeviceArray
    r, w, _ = parallel.init("gloo")
    assert (r, w) == (rank, world)
    W, H, frames, rigs = 320, 180, 3, 4                           # `rigs` cameras, `frames` time steps each
    lo, hi = parallel.shard(rigs, rank, world)
    sp = node.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    local = parallel.ScanBuffer(frames, sp.bins, "cpu")
    local.bins.fill_(1e9); local.meta.copy_(torch.tensor([[400., -400., 1e9, -500.]] * frames, dtype=torch.float64))
    per_rig = {}
    with jn.Elas(jn.Elas.parameters(0), W, H, max_batch=frames, slots=2, host_threads=2) as e:
        for k, rig in enumerate(range(lo, hi)):
            Ls = np.stack([node.synth_pair(W, H, 30 + 6 * rig, 4000 + 10 * rig + t)[0] for t in range(frames)])
            Rs = np.stack([node.synth_pair(W, H, 30 + 6 * rig, 4000 + 10 * rig + t)[1] for t in range(frames)])
            dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
            d1 = DeviceArray.from_numpy(np.zeros((frames, H, W), np.float32)); d2 = DeviceArray.from_numpy(np.zeros((frames, H, W), np.float32))
            u8 = DeviceArray((frames, H, W), np.uint8); bins = DeviceArray((frames, sp.bins), np.float64); meta = DeviceArray((frames, 4), np.float64)
            st = (C.c_int32 * frames)()
            e.submit_scan(k % 2, frames, dL.ptr, dR.ptr, W, H * W, d1.ptr, d2.ptr, sp, lut.ptr, u8.ptr, bins.ptr, meta.ptr, st)
            e.wait(k % 2)
            assert list(st) == [0] * frames
            b, m = bins.numpy(), meta.numpy()
            per_rig[rig] = (b.copy(), m.copy())
            local.bins.copy_(torch.minimum(local.bins, torch.from_numpy(b)))
            mt = torch.from_numpy(m)
            local.meta[:, 0::2] = torch.minimum(local.meta[:, 0::2], mt[:, 0::2])
            local.meta[:, 1::2] = torch.maximum(local.meta[:, 1::2], mt[:, 1::2])
    np.save(os.path.join(out_dir, "local_bins%d.npy" % rank), local.bins.numpy().copy())
    np.save(os.path.join(out_dir, "local_meta%d.npy" % rank), local.meta.numpy().copy())
    for rig, (b, m) in per_rig.items():
        np.save(os.path.join(out_dir, "rig%d_bins.npy" % rig), b)
        np.save(os.path.join(out_dir, "rig%d_meta.npy" % rig), m)
    local.merge()                                                 # ONE MIN all-reduce of the packed buffer
    np.save(os.path.join(out_dir, "merged_bins%d.npy" % rank), local.bins.numpy())
    np.save(os.path.join(out_dir, "merged_meta%d.npy" % rank), local.meta.numpy())
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_hip_scans_merge_to_elementwise_min(tmp_path, oracle):
    """BASELINE config 4 in miniature on one GPU: two rank processes, each runs ITS rigs through the HIP path
    (jn_elas_submit_scan), the robot-level scan is the element-wise MIN over all rigs (point_cloud.cpp:264-266 across
    rigs) with min/max of the extrema, identical on both ranks — and equal to what the oracle gives for the same rigs."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ld = lambda name: np.load(tmp_path / name)
    mb0, mb1, mm0, mm1 = ld("merged_bins0.npy"), ld("merged_bins1.npy"), ld("merged_meta0.npy"), ld("merged_meta1.npy")
    assert np.array_equal(mb0, mb1) and np.array_equal(mm0, mm1)
    lb0, lb1, lm0, lm1 = ld("local_bins0.npy"), ld("local_bins1.npy"), ld("local_meta0.npy"), ld("local_meta1.npy")
    assert np.array_equal(mb0, np.minimum(lb0, lb1))
    assert np.array_equal(mm0[:, 0::2], np.minimum(lm0[:, 0::2], lm1[:, 0::2])) and np.array_equal(mm0[:, 1::2], np.maximum(lm0[:, 1::2], lm1[:, 1::2]))
    assert not np.array_equal(lb0, lb1)                           # the ranks really had different rigs
    # the same robot-level scan from the oracle chain
    W, H, frames, rigs = 320, 180, 3, 4
    spo = oracle.scan_params(W, H)
    luto = oracle.valid_lut(spo, W, H)
    exp_b = np.full((frames, spo.bins), 1e9); exp_m = np.array([[400., -400., 1e9, -500.]] * frames)
    for rig in range(rigs):
        rb, rm = ld("rig%d_bins.npy" % rig), ld("rig%d_meta.npy" % rig)
        for t in range(frames):
            L, R = oracle.synth_pair(W, H, 30 + 6 * rig, 4000 + 10 * rig + t)
            _, D1o, _ = oracle.process(oracle.params(0), L, R)
            bo, mo, _ = oracle.scan(spo, oracle.to_u8(D1o), luto)
            assert np.allclose(rb[t], bo, rtol=0, atol=SCAN_TOL) and np.allclose(rm[t], mo, rtol=0, atol=SCAN_TOL), (rig, t)
            exp_b[t] = np.minimum(exp_b[t], bo)
            exp_m[t] = [min(exp_m[t, 0], mo[0]), max(exp_m[t, 1], mo[1]), min(exp_m[t, 2], mo[2]), max(exp_m[t, 3], mo[3])]
    assert np.allclose(mb0, exp_b, rtol=0, atol=SCAN_TOL) and np.allclose(mm0, exp_m, rtol=0, atol=SCAN_TOL)


@pytest.mark.timeout(900)
def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` without a launcher: the script spawns two rank processes (here both on the box's one GPU,
    gloo for the merge), reports n_gpus = 2 from the process group, and its self-check ties what it timed to the
    reference's golden hash."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-gpu", "--steps", "3",
           "--warmup", "1", "--batch", "4", "--slots", "2", "--min-time", "0", "--no-cpu-baseline", "--no-latency-config"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=800, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and len(j["ranks"]) == 2 and {r["rank"] for r in j["ranks"]} == {0, 1}
    # every frame of both slots against the compiled reference's recorded answers (tests/golden/bench_batch_golden.json)
    assert j["check"]["ok"] is True and j["check"]["frames_checked"]["D1"] == 2 * 4 and j["check"]["n_mismatches"] == 0
    assert j["value"] > 0 and j["config"]["pairs_failed"] == 0
    assert "roofline" in j and j["roofline"]["ms_per_launch"] > 0
    # the two ranks were given disjoint host cores (when there are at least two cores to share)
    from jackal_navigation_amd.parallel import parse_cpulist
    sets = [set(parse_cpulist(r["pin"]["cpulist"])) for r in j["ranks"]]
    assert all(sets)
    if len(os.sched_getaffinity(0)) >= 4:
        assert not (sets[0] & sets[1]), sets


def _elas(jn, p, L, R):
    H, W = L.shape
    D1 = np.zeros((H, W), np.float32); D2 = np.zeros((H, W), np.float32)
    with jn.Elas(p, W, H, host_threads=4) as e:
        st = e.process(np.ascontiguousarray(L), np.ascontiguousarray(R), D1, D2, (W, H, W))
    return st, D1, D2


def test_middlebury_preset_against_reference_hashes_and_oracle(jn, oracle, same):
    """SURVEY row a9 and the whole MIDDLEBURY preset (elas.h:118-145): addCornerSupportPoints (six border points, two of
    them outside the image), plane radius 3, match_texture 0, unbounded gap interpolation with border extrapolation, median
    filter, both sides post-processed.  Against the compiled reference's hashes (uninitialised descriptor bytes zero-filled,
    tests/golden/make_middlebury_golden.py) and bit for bit against the oracle."""
    import os
    from scenes import make_scene
    rows = [l.split() for l in open(os.path.join(ROOT, "tests", "golden", "reference_middlebury_hashes.txt")) if not l.startswith("#")]
    assert len(rows) >= 8
    for kind, W, H, sd, dmax, seed, h1, h2 in rows:
        W, H, sd, dmax, seed = int(W), int(H), int(sd), int(dmax), int(seed)
        L, R = oracle.synth_pair(W, H, sd, seed) if kind == "synth" else make_scene(kind, W, H, dmax, seed)
        st, D1, D2 = _elas(jn, jn.Elas.parameters(1, disp_max=dmax), L, R)
        assert st == 0 and oracle.fnv(D1) == int(h1, 16) and oracle.fnv(D2) == int(h2, 16), (kind, W, H)
        if W * H <= 640 * 480:
            _, D1o, D2o = oracle.process(oracle.params(1, disp_max=dmax), L, R)
            assert same(D1, D1o) and same(D2, D2o)
        assert (D1 >= 0).mean() > 0.97                       # "full size disparity map": almost nothing stays invalid


@pytest.mark.parametrize("kw", [
    {"add_corners": 1}, {"ipol_gap_width": 5000}, {"ipol_gap_width": 100, "postprocess_only_left": 0},
    {"add_corners": 1, "ipol_gap_width": 9}, {"add_corners": 1, "ipol_gap_width": 5000, "filter_adaptive_mean": 0},
])
def test_corner_points_and_wide_gaps_with_the_robotics_preset(jn, oracle, same, kw):
    """add_corners and gap widths beyond 64 (the general gap kernels) switched on one at a time on top of the node's preset,
    on a scene with occlusions (wide invalid runs) and on a ragged image size; a batch through the device-pointer API."""
    from scenes import make_scene
    from jackal_navigation_amd.device import DeviceArray
    for (L, R, dmax) in (make_scene("strips", 320, 240, 79, 33) + (79,), oracle.synth_pair(333, 201, 30, 5) + (95,)):
        st, D1, D2 = _elas(jn, jn.Elas.parameters(0, disp_max=dmax, **kw), L, R)
        st_o, D1o, D2o = oracle.process(oracle.params(0, disp_max=dmax, **kw), L, R)
        assert st == st_o == 0 and same(D1, D1o) and same(D2, D2o), (kw, L.shape)
    W, H, n = 320, 180, 4
    Ls = np.stack([oracle.synth_pair(W, H, 48, 80 + b)[0] for b in range(n)]); Rs = np.stack([oracle.synth_pair(W, H, 48, 80 + b)[1] for b in range(n)])
    dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
    d1 = DeviceArray.from_numpy(np.zeros((n, H, W), np.float32)); d2 = DeviceArray.from_numpy(np.zeros((n, H, W), np.float32))
    with jn.Elas(jn.Elas.parameters(0, **kw), W, H, max_batch=n, host_threads=2) as e:
        assert e.process_batch(n, dL.ptr, dR.ptr, W, H * W, d1.ptr, d2.ptr) == [0] * n
    for b in range(n):
        _, D1o, D2o = oracle.process(oracle.params(0, **kw), Ls[b], Rs[b])
        assert same(d1.numpy()[b], D1o) and same(d2.numpy()[b], D2o), (kw, b)


@pytest.mark.parametrize("min_pixels", ["0", "10000000", "1000000000000"])
def test_both_post_processing_routes(jn, oracle, same, monkeypatch, min_pixels):
    """Gap interpolation + adaptive mean run as one fused pass for large batches and as four short kernels for a lone pair
    or a small batch (JN_POST_FUSED_MIN_PIXELS, read per batch; default 10 M pixels): both give the oracle's bits, for a
    lone pair, a ragged size, and a batch that crosses the default threshold."""
    from jackal_navigation_amd.device import DeviceArray
    monkeypatch.setenv("JN_POST_FUSED_MIN_PIXELS", min_pixels)
    for (W, H, sd, seed) in ((640, 480, 64, 3), (324, 203, 40, 4)):
        L, R = oracle.synth_pair(W, H, sd, seed)
        st, D1, D2 = _elas(jn, jn.Elas.parameters(0, disp_max=sd - 1), L, R)
        _, D1o, D2o = oracle.process(oracle.params(0, disp_max=sd - 1), L, R)
        assert st == 0 and same(D1, D1o) and same(D2, D2o), (W, H, min_pixels)
    W, H, n = 1280, 720, 12                                   # 11 M pixels: fused by default
    Ls = np.zeros((n, H, W), np.uint8); Rs = np.zeros((n, H, W), np.uint8)
    for b in range(n):
        Ls[b], Rs[b] = oracle.synth_pair(W, H, 100, 200 + b % 2)
    dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
    d1 = DeviceArray.from_numpy(np.zeros((n, H, W), np.float32)); d2 = DeviceArray.from_numpy(np.zeros((n, H, W), np.float32))
    with jn.Elas(jn.Elas.parameters(0, disp_max=127), W, H, max_batch=n, host_threads=8) as e:
        assert e.process_batch(n, dL.ptr, dR.ptr, W, H * W, d1.ptr, d2.ptr) == [0] * n
    o1, o2 = d1.numpy(), d2.numpy()
    for b in (0, 1):
        _, D1o, D2o = oracle.process(oracle.params(0, disp_max=127), Ls[b], Rs[b])
        assert same(o1[b], D1o) and same(o2[b], D2o)
    for b in range(2, n):
        assert same(o1[b], o1[b % 2]) and same(o2[b], o2[b % 2])


def test_imdecode_gray_on_the_gpu_equals_libjpeg(jn):
    """cv::imdecode(GRAYSCALE) (point_cloud.cpp:436, :478): entropy decode on the host, dequantisation + IDCT on the GPU,
    against the Pillow / libjpeg-turbo fixtures, and against Pillow itself on fresh random frames when it is importable."""
    import hashlib
    import io
    import os
    from jackal_navigation_amd import node, _lib
    z = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"))
    names = sorted({k.split("__")[0] for k in z.files} - {"progressive"})
    assert len(names) >= 10
    for name in names:
        img = node.imdecode_gray(z[name + "__jpeg"]).numpy()
        assert img.shape == tuple(z[name + "__shape"]), name
        assert hashlib.sha256(img.tobytes()).digest() == z[name + "__sha256"].tobytes(), name
        if name + "__gray" in z.files:
            assert np.array_equal(img, z[name + "__gray"])
    with pytest.raises(_lib.JnError) as e:
        node.imdecode_gray(z["progressive__jpeg"])
    assert e.value.status == _lib.JN_ERR_UNSUPPORTED
    try:
        from PIL import Image
    except ImportError:
        return
    rng = np.random.default_rng(4)
    for k in range(12):
        W, H = int(rng.integers(9, 700)), int(rng.integers(9, 400))
        rgb = (rng.integers(0, 256, (H // 8 + 1, W // 8 + 1, 3)).repeat(8, 0).repeat(8, 1)[:H, :W] + rng.integers(-12, 12, (H, W, 3))).clip(0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, "JPEG", quality=int(rng.integers(20, 98)), subsampling=int(rng.integers(0, 3)), optimize=bool(k & 1))
        im = Image.open(io.BytesIO(buf.getvalue())); im.draft("L", im.size); im.load()
        assert np.array_equal(node.imdecode_gray(buf.getvalue()).numpy(), np.asarray(im)), (k, W, H)


def test_both_eyes_decoded_on_two_threads_equal_two_single_decodes(jn):
    """jn_jpeg_decode_gray_pair: the right eye's entropy decode runs on a helper thread.  Same pixels as two single calls, for the
    webcam pair and for every same-sized couple of the fixtures, many frames in a row (the helper thread is re-used), from two
    calling threads at once; a damaged right eye and eyes of different sizes are refused."""
    import os
    import threading
    from jackal_navigation_amd import node, _lib
    z = np.load(os.path.join(ROOT, "tests", "golden", "stereo_jpeg_pair.npz"))
    jl, jr = z["left__jpeg"], z["right__jpeg"]
    one_l, one_r = node.imdecode_gray(jl).numpy(), node.imdecode_gray(jr).numpy()
    for _ in range(20):
        a, b = node.imdecode_gray_pair(jl, jr)
        assert np.array_equal(a.numpy(), one_l) and np.array_equal(b.numpy(), one_r)
    a, b = node.imdecode_gray_pair(jr, jl)
    assert np.array_equal(a.numpy(), one_r) and np.array_equal(b.numpy(), one_l)
    errs = []

    def worker():
        try:
            for _ in range(15):
                a, b = node.imdecode_gray_pair(jl, jr)
                assert np.array_equal(a.numpy(), one_l) and np.array_equal(b.numpy(), one_r)
        except Exception as e:            # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=worker) for _ in range(3)]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert not errs, errs
    cases = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"))
    with pytest.raises(_lib.JnError) as e:
        node.imdecode_gray_pair(jl, jr[2:])                                  # no SOI: damaged (a merely truncated scan decodes, like libjpeg's warning path)
    assert e.value.status == _lib.JN_ERR_INVALID
    with pytest.raises(_lib.JnError) as e:
        node.imdecode_gray_pair(jl, cases["ragged_35x21_q95_422__jpeg"])
    assert e.value.status == _lib.JN_ERR_INVALID
    a, b = node.imdecode_gray_pair(jl, jr)                                   # and the pair still works afterwards
    assert np.array_equal(a.numpy(), one_l) and np.array_equal(b.numpy(), one_r)


def test_whole_frame_from_jpeg_bytes_to_laser_scan(jn, oracle, same):
    """One frame the way the node sees it (point_cloud.cpp:431-490 -> :406-429 -> :213-296): the two compressed images of
    tests/golden/stereo_jpeg_pair.npz -> imdecode (GPU IDCT, pinned by libjpeg's SHA-256) -> rectification maps of the
    shipped calibration -> remap -> ELAS on device pointers -> u8 map + LUT scan -> LaserScan message and the consumer's
    decision; every stage against the oracle chain fed with the same decoded frames."""
    import hashlib
    from jackal_navigation_amd import node
    from jackal_navigation_amd.device import DeviceArray
    z = np.load(os.path.join(ROOT, "tests", "golden", "stereo_jpeg_pair.npz"))
    W, H = 320, 180
    c = node.stereo_calib()
    r = node.stereo_rectify(c, W, H)
    rect_dev, rect_host = [], []
    for name, K, D, Rr, P in (("left", c.K1, c.D1, r.R1, r.P1), ("right", c.K2, c.D2, r.R2, r.P2)):
        frame = node.imdecode_gray(z[name + "__jpeg"])                                   # device, 360 x 640
        raw = frame.numpy()
        assert raw.shape == (360, 640) and hashlib.sha256(raw.tobytes()).digest() == z[name + "__sha256"].tobytes(), name
        mx, my = node.init_undistort_rectify_map(list(K), list(D), list(Rr), list(P), W, H)
        out = DeviceArray((H, W), np.uint8)
        node.remap(1, frame.ptr, 640, 360, 640, 640 * 360, mx.ptr, my.ptr, out.ptr, W, H, W, W * H)
        assert same(out.numpy(), oracle.remap(raw, mx.numpy(), my.numpy())), name
        rect_dev.append(out); rect_host.append(out.numpy())
    sp, spo = node.scan_params(W, H), oracle.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    d1 = DeviceArray.from_numpy(np.zeros((1, H, W), np.float32)); d2 = DeviceArray.from_numpy(np.zeros((1, H, W), np.float32))
    u8 = DeviceArray((1, H, W), np.uint8); bins = DeviceArray((1, sp.bins), np.float64); meta = DeviceArray((1, 4), np.float64)
    st = (C.c_int32 * 1)()
    with jn.Elas(jn.Elas.parameters(0), W, H, max_batch=1, host_threads=4) as e:         # the node's parameters: ROBOTICS, disp_max 255
        e.submit_scan(0, 1, rect_dev[0].ptr, rect_dev[1].ptr, W, H * W, d1.ptr, d2.ptr, sp, lut.ptr, u8.ptr, bins.ptr, meta.ptr, st)
        e.wait(0)
    st_o, D1o, D2o = oracle.process(oracle.params(0), rect_host[0], rect_host[1])
    assert list(st) == [0] and st_o == 0 and same(d1.numpy()[0], D1o)
    assert (D1o >= 0).mean() > 0.6                                                        # a scene, not a degenerate frame
    u8o = oracle.to_u8(D1o)
    assert same(u8.numpy()[0], u8o)
    bo, mo, used = oracle.scan(spo, u8o, oracle.valid_lut(spo, W, H))
    assert used > 1000 and np.array_equal(bins.numpy()[0] < 1e9 - 1, bo < 1e9 - 1)
    assert np.allclose(bins.numpy()[0], bo, rtol=0, atol=SCAN_TOL) and np.allclose(meta.numpy()[0], mo, rtol=0, atol=SCAN_TOL)
    msg = node.laser_scan_message(bins.numpy()[0], meta.numpy()[0], seq=1)
    assert msg["header"]["frame_id"] == "jackal" and len(msg["ranges"]) == int((bo < 1e9 - 1).sum()) and len(msg["ranges"]) > 30


def test_submit_host_streams_host_buffers_through_the_slots(jn, oracle, same):
    """jn_elas_submit_host: batches of host-resident pairs through three slots at once (copies of one overlap kernels of the
    others), a padded-pitch single pair, and a batch with a pair that has too few support points (its host maps stay untouched)."""
    W, H, n = 320, 180, 3
    rng = np.random.default_rng(2)
    batches = []
    for s in range(3):
        Ls = np.stack([oracle.synth_pair(W, H, 48, 500 + 10 * s + b)[0] for b in range(n)]); Rs = np.stack([oracle.synth_pair(W, H, 48, 500 + 10 * s + b)[1] for b in range(n)])
        batches.append((Ls, Rs, np.full((n, H, W), 3.0, np.float32), np.full((n, H, W), 3.0, np.float32), (C.c_int32 * n)()))
    batches[1][0][1] = rng.integers(0, 255, (H, W)); batches[1][1][1] = rng.integers(0, 255, (H, W))        # noise: no support points
    with jn.Elas(jn.Elas.parameters(0), W, H, max_batch=n, slots=3, host_threads=4) as e:
        for rep in range(2):
            for s, (Ls, Rs, D1, D2, st) in enumerate(batches):
                e.submit_host(s, Ls, Rs, D1, D2, st)
            for s in range(3):
                e.wait(s)
        # a single pair whose rows are padded (pitch > width), through the raw call
        Lp = np.zeros((H, W + 24), np.uint8); Rp = np.zeros((H, W + 24), np.uint8)
        Lp[:, :W], Rp[:, :W] = batches[0][0][0], batches[0][1][0]
        d1 = np.zeros((H, W), np.float32); d2 = np.zeros((H, W), np.float32); st1 = (C.c_int32 * 1)()
        from jackal_navigation_amd import _lib
        _lib.check(jn.load().jn_elas_submit_host(e._h, 0, 1, Lp.ctypes.data, Rp.ctypes.data, W + 24, 0, d1.ctypes.data, d2.ctypes.data, st1), "jn_elas_submit_host")
        e.wait(0)
    po = oracle.params(0)
    for s, (Ls, Rs, D1, D2, st) in enumerate(batches):
        for b in range(n):
            st_o, D1o, D2o = oracle.process(po, Ls[b], Rs[b])
            assert st[b] == st_o, (s, b)
            if st_o == 0:
                assert same(D1[b], D1o) and same(D2[b], D2o), (s, b)
            else:
                assert (D1[b] == 3.0).all() and (D2[b] == 3.0).all()                                        # untouched (elas.cpp:66-71)
    assert list(batches[1][4]) == [0, 1, 0]
    _, D1o, D2o = oracle.process(po, batches[0][0][0], batches[0][1][0])
    assert st1[0] == 0 and same(d1, D1o) and same(d2, D2o)


@pytest.mark.parametrize("sorts", ["0", "1"])
def test_gpu_arrangement_equals_the_hosts(jn, hooks, monkeypatch, sorts):
    """k_arrange (the alternating-cut arrangement of a frame side's support points, computed on the GPU so that the host only
    runs the hull recursion) against Delaunay::arrange + split on the host: lattice points as the support list holds them
    (u-major, v ascending), right image x = u - d; sizes from 3 to the kernel's limit (8192 vertices in LDS with 64-bit keys, 12288 with compact
    keys, 16384 through global scratch); sides with coinciding vertices and sides beyond the limit are handed back (ok = 0).
    Both ways of ordering the vertices: by ranks (bitmaps + prefix counts: what runs where the bitmaps fit the LDS) and by the bitonic
    sorts (JN_ARRANGE_SORTS=1 in the hooks build; what runs for lattices too large for the bitmaps)."""
    monkeypatch.setenv("JN_ARRANGE_SORTS", sorts)
    L = jn.load()
    rng = np.random.default_rng(12)

    def case(n, cw, ch, dmax, step=5, force_dup=False, row_d=False):
        cells = rng.choice(cw * ch, size=n, replace=False)
        cells.sort()                                            # uc-major, vc ascending = the list's order
        uc, vc = cells // ch, cells % ch
        d = rng.integers(0, dmax + 1, n)
        if row_d:
            d = (vc * 7) % (dmax + 1)                           # one disparity per lattice row: (u - d, v) stays distinct, the order changes
        if force_dup and n >= 2:                                # two right-image vertices coincide: (u - d, v) equal, (u, v) distinct
            taken = set(zip(uc.tolist(), vc.tolist()))
            for j in range(1, n):
                if uc[j] > uc[j - 1] and (int(uc[j]), int(vc[j - 1])) not in taken:
                    vc[j] = vc[j - 1]; d[j] = d[j - 1] + step * (uc[j] - uc[j - 1])
                    break
            order = np.lexsort((vc, uc)); uc, vc, d = uc[order], vc[order], d[order]      # keep the list's order
        tri = np.stack([uc, vc, d], axis=1).astype(np.int16).copy()
        left = np.zeros(max(n, 1), np.uint16); right = np.zeros(max(n, 1), np.uint16); ok = (C.c_int32 * 2)()
        assert L.jn_device_arrangement(0, tri.ctypes.data, n, step, left.ctypes.data, right.ctypes.data, ok) == 0
        for side, got in ((0, left), (1, right)):
            x = (uc * step - (d if side else 0)).astype(np.int32).copy(); y = (vc * step).astype(np.int32).copy()
            exp = np.zeros(max(n, 1), np.uint16)
            host_ok = L.jn_host_arrangement(x.ctypes.data, y.ctypes.data, n, exp.ctypes.data)
            distinct = len(set(zip(x.tolist(), y.tolist()))) == n
            if n < 3 or n > 16384 or not distinct:
                assert ok[side] == 0, (n, side)
            else:
                assert ok[side] == 1 and host_ok == 1, (n, side)
                assert np.array_equal(got[:n], exp[:n]), (n, side, int((got[:n] != exp[:n]).sum()))
        return ok[0], ok[1]

    for n in (0, 2, 3, 4, 5, 7, 8, 13, 64, 100, 1023, 1024, 1025, 3232, 5000, 8192):
        case(n, 256, 144, 127)
    for n in (8193, 11200, 16384):
        assert case(n, 384, 216, 255, row_d=True) == (1, 1)     # beyond the LDS: the working arrays live in global scratch
        assert case(n, 384, 216, 255)[0] == 1                   # random disparities this dense: right-image vertices coincide
    assert case(16385, 384, 216, 255) == (0, 0)                 # beyond the kernel's limit: host
    assert case(3000, 2000, 100, 127, row_d=True) == (1, 1)     # a lattice whose bitmaps do not fit the LDS (10 000 columns): the sorts, whatever JN_ARRANGE_SORTS says
    for n in (50, 700, 4000):
        case(n, 384, 216, 255)                                  # 1080p lattice, D = 256: negative right-image columns
    dup = 0
    for n in (10, 300, 3000):
        a, b = case(n, 256, 144, 127, force_dup=True)
        dup += (b == 0)
    assert dup >= 1                                             # the coinciding pair was seen and handed back


def test_merge_as_the_tail_of_scan_batches(jn, oracle):
    """jn_elas_set_comm: with a communicator attached every scan batch ends with pack -> ncclAllReduce(MIN) -> unpack, queued by
    the slot worker in submission order.  One rank here (MIN over one rank is the identity), four slots, twelve batches of
    different frames in flight: the bins must equal the un-merged run's bit for bit, a merge time is reported, and detaching
    restores the plain path."""
    from jackal_navigation_amd.device import DeviceArray
    from jackal_navigation_amd import node, parallel
    W, H, B, S, rounds = 320, 180, 3, 4, 3
    sp = node.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    pairs = [[node.synth_pair(W, H, 30 + 4 * k, 900 + 7 * k + t) for t in range(B)] for k in range(S * rounds)]
    dLs = [DeviceArray.from_numpy(np.stack([p[0] for p in ps])) for ps in pairs]
    dRs = [DeviceArray.from_numpy(np.stack([p[1] for p in ps])) for ps in pairs]

    def run_all(e):
        got = []
        bufs = [dict(d1=DeviceArray.from_numpy(np.zeros((B, H, W), np.float32)), d2=DeviceArray.from_numpy(np.zeros((B, H, W), np.float32)),
                     u8=DeviceArray((B, H, W), np.uint8), bins=DeviceArray((B, sp.bins), np.float64), meta=DeviceArray((B, 4), np.float64),
                     st=(C.c_int32 * B)()) for _ in range(S)]
        inflight = []
        for k in range(S * rounds):
            slot = k % S
            if len(inflight) == S:
                s0, k0 = inflight.pop(0)
                e.wait(s0)
                got.append((bufs[s0]["bins"].numpy().copy(), bufs[s0]["meta"].numpy().copy(), e.merge_time(s0)))
            b = bufs[slot]
            e.submit_scan(slot, B, dLs[k].ptr, dRs[k].ptr, W, H * W, b["d1"].ptr, b["d2"].ptr, sp, lut.ptr, b["u8"].ptr, b["bins"].ptr, b["meta"].ptr, b["st"])
            inflight.append((slot, k))
        for s0, k0 in inflight:
            e.wait(s0)
            got.append((bufs[s0]["bins"].numpy().copy(), bufs[s0]["meta"].numpy().copy(), e.merge_time(s0)))
        return got

    with jn.Elas(jn.Elas.parameters(0), W, H, max_batch=B, slots=S, host_threads=4) as e:
        plain = run_all(e)
        comm = parallel.ScanComm(0, 1, 0, lambda raw: raw)
        e.set_comm(comm)
        merged = run_all(e)
        e.set_comm(None)
        again = run_all(e)
        comm.close()
    assert len(plain) == len(merged) == S * rounds
    for (b0, m0, t0), (b1, m1, t1), (b2, m2, t2) in zip(plain, merged, again):
        assert np.array_equal(b0, b1) and np.array_equal(m0, m1) and np.array_equal(b0, b2) and np.array_equal(m0, m2)
        assert t0 == 0.0 and t1 > 0.0 and t2 == 0.0
    assert (plain[0][0] < 1e9 - 1).any()


def test_support_count_growing_from_batch_to_batch_on_one_slot(jn, oracle, same, monkeypatch):
    """k_arrange's LDS space follows the support counts of the slot's last batches.  A batch of nearly textureless
    frames (few support points) followed by dense ones on the SAME slot hands the larger sides back to the host (ok = 0) — the
    results must not change: every batch equals the oracle and the run with JN_GPU_ARRANGE=0."""
    from jackal_navigation_amd.device import DeviceArray
    W, H, n = 640, 480, 3
    rng = np.random.default_rng(21)
    dense = [oracle.synth_pair(W, H, 64, 3000 + b) for b in range(n)]
    sparse = []
    for b in range(n):                                        # a small textured window on a flat background: a handful of support points
        L, R = oracle.synth_pair(W, H, 64, 4000 + b)
        flat = np.full((H, W), 90, np.uint8)
        Ls, Rs = flat.copy(), flat.copy()
        Ls[200:280, 250:390] = L[200:280, 250:390]; Rs[200:280, 250:390] = R[200:280, 250:390]
        sparse.append((Ls, Rs))
    p = jn.Elas.parameters(0, disp_max=63)
    seq = [sparse, dense, sparse, dense, dense]

    def run_seq():
        outs = []
        with jn.Elas(p, W, H, max_batch=n, slots=1, host_threads=4) as e:
            for batch in seq:
                dL = DeviceArray.from_numpy(np.stack([x[0] for x in batch])); dR = DeviceArray.from_numpy(np.stack([x[1] for x in batch]))
                d1 = DeviceArray.from_numpy(np.zeros((n, H, W), np.float32)); d2 = DeviceArray.from_numpy(np.zeros((n, H, W), np.float32))
                st = (C.c_int32 * n)()
                e.submit(0, n, dL.ptr, dR.ptr, W, H * W, d1.ptr, d2.ptr, st)
                e.wait(0)
                outs.append((list(st), d1.numpy().copy(), d2.numpy().copy()))
        return outs

    monkeypatch.setenv("JN_GPU_ARRANGE", "1")
    on = run_seq()
    monkeypatch.setenv("JN_GPU_ARRANGE", "0")
    off = run_seq()
    po = oracle.params(0, disp_max=63)
    for k, (a, b) in enumerate(zip(on, off)):
        assert a[0] == b[0] and same(a[1], b[1]) and same(a[2], b[2]), k
        for i in (0, n - 1):
            sto, D1o, D2o = oracle.process(po, seq[k][i][0], seq[k][i][1])
            if sto == 0:
                assert a[0][i] == 0 and same(a[1][i], D1o) and same(a[2][i], D2o), (k, i)
            else:
                assert a[0][i] == 1


def test_start_up_pacing_changes_no_result(jn, oracle, monkeypatch):
    """JN_PACE (default on for batch handles with several slots): a batch's stage A waits on the device for the previous submission's two
    heavy kernels.  It only orders work in time: four slots, twelve batches submitted as fast as the slots free up, with and without it,
    and the oracle for one frame of each."""
    from jackal_navigation_amd.device import DeviceArray
    W, H, B, S, rounds = 320, 180, 2, 4, 3
    pairs = [[oracle.synth_pair(W, H, 24 + 3 * k, 400 + 5 * k + t) for t in range(B)] for k in range(S * rounds)]
    dLs = [DeviceArray.from_numpy(np.stack([p[0] for p in ps])) for ps in pairs]
    dRs = [DeviceArray.from_numpy(np.stack([p[1] for p in ps])) for ps in pairs]
    results = {}
    for pace in ("1", "0"):
        monkeypatch.setenv("JN_PACE", pace)
        outs = []
        with jn.Elas(jn.Elas.parameters(0), W, H, max_batch=B, slots=S, host_threads=4) as e:
            bufs = [dict(d1=DeviceArray.from_numpy(np.zeros((B, H, W), np.float32)), d2=DeviceArray.from_numpy(np.zeros((B, H, W), np.float32)),
                         st=(C.c_int32 * B)()) for _ in range(S)]
            inflight = []
            for k in range(S * rounds):
                slot = k % S
                if len(inflight) == S:
                    s0 = inflight.pop(0)
                    e.wait(s0)
                    outs.append(bufs[s0]["d1"].numpy().copy())
                b = bufs[slot]
                e.submit(slot, B, dLs[k].ptr, dRs[k].ptr, W, H * W, b["d1"].ptr, b["d2"].ptr, b["st"])
                inflight.append(slot)
            for s0 in inflight:
                e.wait(s0)
                outs.append(bufs[s0]["d1"].numpy().copy())
        results[pace] = outs
    assert len(results["1"]) == S * rounds
    for a, b in zip(results["1"], results["0"]):
        assert np.array_equal(a, b)
    p = oracle.params(0)
    for k in (0, 5, 11):
        st, d1, _ = oracle.process(p, pairs[k][0][0], pairs[k][0][1])
        assert st == 0 and np.array_equal(results["1"][k][0], d1), k


def _scan_rounds(jn, W, H, B, S, rounds, comm_env, monkeypatch, fail_seq=None):
    """S slots x `rounds` scan batches of different frames with a one-rank communicator attached; returns (bins per batch or the
    JnError of the batch, the order the merges were queued in)."""
    from jackal_navigation_amd.device import DeviceArray
    from jackal_navigation_amd import node, parallel, _lib
    for k, v in comm_env.items():
        monkeypatch.setenv(k, v)
    if fail_seq is not None:
        monkeypatch.setenv("JN_TEST_FAIL_SEQ", str(fail_seq))
    sp = node.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    pairs = [[node.synth_pair(W, H, 30 + 4 * k, 900 + 7 * k + t) for t in range(B)] for k in range(S * rounds)]
    dLs = [DeviceArray.from_numpy(np.stack([p[0] for p in ps])) for ps in pairs]
    dRs = [DeviceArray.from_numpy(np.stack([p[1] for p in ps])) for ps in pairs]
    got = [None] * (S * rounds)
    with jn.Elas(jn.Elas.parameters(0), W, H, max_batch=B, slots=S, host_threads=4) as e:
        comm = parallel.ScanComm(0, 1, 0, lambda raw: raw)
        e.set_comm(comm)
        bufs = [dict(d1=DeviceArray.from_numpy(np.zeros((B, H, W), np.float32)), d2=DeviceArray.from_numpy(np.zeros((B, H, W), np.float32)),
                     u8=DeviceArray((B, H, W), np.uint8), bins=DeviceArray((B, sp.bins), np.float64), meta=DeviceArray((B, 4), np.float64),
                     st=(C.c_int32 * B)()) for _ in range(S)]

        def finish(s0, k0):
            try:
                e.wait(s0)
                got[k0] = bufs[s0]["bins"].numpy().copy()
            except _lib.JnError as err:
                got[k0] = err
        inflight = []
        for k in range(S * rounds):
            slot = k % S
            if len(inflight) == S:
                finish(*inflight.pop(0))
            b = bufs[slot]
            e.submit_scan(slot, B, dLs[k].ptr, dRs[k].ptr, W, H * W, b["d1"].ptr, b["d2"].ptr, sp, lut.ptr, b["u8"].ptr, b["bins"].ptr, b["meta"].ptr, b["st"])
            inflight.append((slot, k))
        for s0, k0 in inflight:
            finish(s0, k0)
        order = e.merge_order()
        e.set_comm(None)
        comm.close()
    return got, order


def test_merges_are_queued_in_submission_order_when_slots_finish_out_of_order(jn, hooks, monkeypatch):
    """the only cross-rank ordering logic (merge_seq / submit_seq) under skew.  Slots 1 and 3 are held up before their
    merge turn (JN_TEST_SLOT_DELAY_US: what a longer host stage does), so batches reach the merge out of submission order; RCCL
    needs every rank to queue a communicator's collectives in ONE order: the queue order must still be 0, 1, 2, ... and the bins
    must equal the unskewed run's."""
    W, H, B, S, rounds = 320, 180, 2, 4, 3
    plain, order0 = _scan_rounds(jn, W, H, B, S, rounds, {}, monkeypatch)
    skew, order1 = _scan_rounds(jn, W, H, B, S, rounds, {"JN_TEST_SLOT_DELAY_US": "0,6000,0,3000"}, monkeypatch)
    assert order0 == list(range(S * rounds)) and order1 == list(range(S * rounds)), (order0, order1)
    for a, b in zip(plain, skew):
        assert isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and np.array_equal(a, b)


def test_a_failing_batch_keeps_its_turn_and_feeds_the_collective_the_identity(jn, hooks, monkeypatch):
    """A batch that dies on this rank before its merge (JN_TEST_FAIL_SEQ) must not strand anybody: it still takes its turn in the
    merge order and contributes the identity of MIN to the all-reduce its peers are waiting in, its jn_elas_wait reports the error,
    and every later batch completes with the right bins."""
    from jackal_navigation_amd import _lib
    W, H, B, S, rounds = 320, 180, 2, 4, 3
    plain, _ = _scan_rounds(jn, W, H, B, S, rounds, {}, monkeypatch)
    got, order = _scan_rounds(jn, W, H, B, S, rounds, {"JN_TEST_SLOT_DELAY_US": "0,2000,0,0"}, monkeypatch, fail_seq=5)
    assert order == list(range(S * rounds)), order
    for k, (a, b) in enumerate(zip(plain, got)):
        if k == 5:
            assert isinstance(b, _lib.JnError) and b.status == _lib.JN_ERR_INTERNAL
        else:
            assert isinstance(b, np.ndarray) and np.array_equal(a, b), k


def _two_gpus():
    import torch
    return torch.cuda.device_count() >= 2


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs: RCCL refuses two ranks on one device (runs on the driver's multi-GPU node only)")
def test_bench_two_ranks_over_rccl_when_two_gpus_exist():
    """Config 4 in small: `bench.py --gpus 2` over RCCL with the merge as the batch's tail through the C-ABI communicator.  Skipped on
    the 1-GPU boxes this build sees; written so that the first machine with two GPUs exercises the N > 1 path end to end."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "4", "--slots", "2", "--min-time", "0",
           "--no-cpu-baseline", "--no-latency-config"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=800, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and {r["rank"] for r in j["ranks"]} == {0, 1} and {r["device"] for r in j["ranks"]} == {0, 1}
    assert all(r["rccl_comm"] and r["rccl_comm"][1] == 2 for r in j["ranks"])          # RCCL itself reports a two-rank communicator on both
    assert j["check"]["ok"] is True and j["check"]["frames_checked"]["D1"] == 2 * 4     # D1 / u8 maps are per rig: unaffected by the merge
    assert j["value"] > 0 and j["config"]["pairs_failed"] == 0


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs")
def test_a_batch_failing_on_one_rank_does_not_strand_the_other(tmp_path):
    """Two ranks over RCCL, rank 1's third scan batch fails before its kernels (JN_TEST_FAIL_SEQ): rank 0 must get ITS OWN scan for that
    batch (rank 1 contributes the identity of MIN) and finish; rank 1 reports the error for that batch only."""
    script = tmp_path / "two_ranks.py"
    script.write_text('''
import ctypes as C, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
rank = int(os.environ["RANK"])
os.environ["JN_STEREO_LIB"] = os.path.join(%r, "jackal_navigation_amd", "libjn_stereo_hooks.so")   # JN_TEST_FAIL_SEQ exists in the hooks build only
if rank == 1:
    os.environ["JN_TEST_FAIL_SEQ"] = "2"
os.environ["JN_COMM_TIMEOUT_MS"] = "20000"
import jackal_navigation_amd as jn
from jackal_navigation_amd import node, parallel, _lib
from jackal_navigation_amd.device import DeviceArray
torch.cuda.set_device(rank)
dist.init_process_group("gloo", rank=rank, world_size=2)
def exchange(raw):
    t = torch.zeros(128, dtype=torch.uint8)
    if raw is not None: t.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    dist.broadcast(t, src=0); return bytes(t.numpy().tobytes())
W, H, B = 320, 180, 2
sp = node.scan_params(W, H); lut = node.build_valid_disp_lut(sp, W, H, device=rank)
comm = parallel.ScanComm(rank, 2, rank, exchange)
res = []
with jn.Elas(jn.Elas.parameters(0), W, H, max_batch=B, device=rank, slots=2, host_threads=4) as e:
    e.set_comm(comm)
    for k in range(4):
        Ls = np.stack([node.synth_pair(W, H, 40, 100 * rank + 10 * k + t)[0] for t in range(B)]); Rs = np.stack([node.synth_pair(W, H, 40, 100 * rank + 10 * k + t)[1] for t in range(B)])
        dL, dR = DeviceArray.from_numpy(Ls, device=rank), DeviceArray.from_numpy(Rs, device=rank)
        d1, d2 = DeviceArray((B, H, W), np.float32, device=rank), DeviceArray((B, H, W), np.float32, device=rank)
        u8, bins, meta = DeviceArray((B, H, W), np.uint8, device=rank), DeviceArray((B, sp.bins), np.float64, device=rank), DeviceArray((B, 4), np.float64, device=rank)
        st = (C.c_int32 * B)()
        e.submit_scan(0, B, dL.ptr, dR.ptr, W, H * W, d1.ptr, d2.ptr, sp, lut.ptr, u8.ptr, bins.ptr, meta.ptr, st)
        try:
            e.wait(0); res.append("ok %%d" %% int((bins.numpy() < 1e9 - 1).sum()))
        except _lib.JnError as err:
            res.append("err %%d" %% err.status)
    e.set_comm(None)
comm.close()
print("RANK", rank, res, flush=True)
''' % (ROOT, ROOT))
    port = 29000 + os.getpid() % 900
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    r0 = [l for l in outs[0].splitlines() if l.startswith("RANK 0")][0]
    r1 = [l for l in outs[1].splitlines() if l.startswith("RANK 1")][0]
    assert r0.count("ok") == 4 and "err" not in r0, r0
    assert r1.count("ok") == 3 and "err %d" % 5 in r1, r1                      # JN_ERR_INTERNAL for batch 2 only


@pytest.mark.gpu
def test_stage_b_queued_behind_the_gate_changes_no_result(jn, oracle, same, monkeypatch):
    """A latency-mode handle (max_batch 1) queues stage B while the GPU still runs stage A, behind a hipStreamWaitValue32 gate that the
    host opens after its stage; launch sizes are then capacities.  With the gate and without it (JN_GATE_STAGE_B=0, read at create time)
    the maps are the oracle's — through device pointers and through host pointers, for a scene with support points and for one with
    none (JN_ERR_FEW_SUPPORT: outputs untouched, the gate must still open), several calls on one handle."""
    import torch
    W, H = 640, 480
    L, R = jn.node.synth_pair(W, H, 64, 4242)
    flat = np.full((H, W), 128, np.uint8)
    _, D1o, D2o = oracle.process(oracle.params(0, disp_max=63, postprocess_only_left=0), L, R)
    dev = torch.device("cuda", 0)
    for gate in ("1", "0"):
        monkeypatch.setenv("JN_GATE_STAGE_B", gate)
        with jn.Elas(jn.Elas.parameters(0, disp_max=63, postprocess_only_left=0), W, H, max_batch=1, host_threads=4) as e:
            for rep in range(3):
                for (a, b, few) in ((L, R, False), (flat, flat, True), (L, R, False)):
                    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
                    o1 = torch.full((H, W), -7.0, dtype=torch.float32, device=dev); o2 = torch.full((H, W), -7.0, dtype=torch.float32, device=dev)
                    torch.cuda.synchronize()
                    st = e.process_batch(1, ta.data_ptr(), tb.data_ptr(), W, H * W, o1.data_ptr(), o2.data_ptr())
                    if few:
                        assert list(st) == [1] and float(o1.min()) == -7.0 and float(o1.max()) == -7.0, (gate, rep)
                    else:
                        assert list(st) == [0] and same(o1.cpu().numpy(), D1o) and same(o2.cpu().numpy(), D2o), (gate, rep)
                D1 = np.zeros((H, W), np.float32); D2 = np.zeros((H, W), np.float32)
                assert e.process(L, R, D1, D2, (W, H, W)) == 0 and same(D1, D1o) and same(D2, D2o), (gate, rep, "host pointers")
