"""The product's cross-rig merge (csrc/comm.cpp, jn_elas_set_comm, jn_scan_allreduce) with TWO real rank processes on ONE GPU.

RCCL refuses two ranks of a communicator on one device, so until a multi-GPU node runs the scaling bench the merge's ordering and failure logic
had only ever met a ONE-rank communicator, where an ordering bug cannot deadlock.  comm.cpp binds librccl at run time and honours JN_RCCL_LIB:
these tests point it at tests/mocks/fake_rccl.cpp — a stand-in that keeps what the logic depends on (stream-ordered asynchronous all-reduce, one
issue order per communicator on every rank, a rank whose peer never joins waits for ever, ncclCommAbort ends the wait) and reduces through
POSIX shared memory.  Everything else is the product: HIP path, slot workers, turnstile, identity on failure, time-out and abort."""
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, B, S, ROUNDS = 320, 180, 2, 4, 3
N = S * ROUNDS
SCAN_TOL = 1e-4
JN_ERR_INTERNAL, JN_ERR_COMM = 5, 6


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    out = tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "mocks", "fake_rccl.cpp"), "-o", str(out), "-lrt"],
                   check=True, capture_output=True, timeout=300)
    yield str(out)
    for f in glob.glob("/dev/shm/jnfake_*"):                       # a killed rank 0 leaves its segment behind
        try:
            os.unlink(f)
        except OSError:
            pass


def run_ranks(fake, out_dir, scenario, env_per_rank, timeout=240):
    procs = []
    world = len(env_per_rank)
    for r in range(world):
        # the ranks load the HOOKS build: JN_TEST_SLOT_DELAY_US / JN_TEST_FAIL_SEQ_MERGED (the skews and the failing batch) exist only there
        env = dict(os.environ, JN_RCCL_LIB=fake, JN_COMM_INIT_TIMEOUT_S="60", JN_STEREO_LIB=os.path.join(ROOT, "jackal_navigation_amd", "libjn_stereo_hooks.so"))
        env.update(env_per_rank[r])
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mocks", "comm_rank_worker.py"), str(r), str(world), str(out_dir), scenario],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    t0 = time.time()
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=max(1.0, timeout - (time.time() - t0)))[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung (scenario %s):\n%s" % (scenario, "\n".join(outs)))
    return [p.returncode for p in procs], outs, time.time() - t0


def load(out_dir, r):
    rep = json.load(open(os.path.join(out_dir, "report%d.json" % r)))
    return rep, np.load(os.path.join(out_dir, "merged_bins%d.npy" % r)), np.load(os.path.join(out_dir, "merged_meta%d.npy" % r)), \
        np.load(os.path.join(out_dir, "local_bins%d.npy" % r)), np.load(os.path.join(out_dir, "local_meta%d.npy" % r))


def merged_of(lb0, lm0, lb1, lm1):
    mm = lm0.copy()
    mm[..., 0::2] = np.minimum(lm0[..., 0::2], lm1[..., 0::2]); mm[..., 1::2] = np.maximum(lm0[..., 1::2], lm1[..., 1::2])
    return np.minimum(lb0, lb1), mm


@pytest.mark.timeout(600)
def test_two_ranks_merge_through_the_product_under_skew(fake_rccl, tmp_path, oracle):
    """12 scan batches over 4 slots on each of two ranks, slots held up differently on the two ranks (JN_TEST_SLOT_DELAY_US): batches reach
    their merge out of submission order and differently on the two ranks — with two ranks a wrong issue order DEADLOCKS (rank 0 inside
    collective 5 while rank 1 sits in collective 6) or pairs up the wrong batches.  Merged bins must be identical on both ranks, equal the
    element-wise MIN of the two ranks' own scans, and those equal the oracle's scans of the same rigs; the queue order must be 0, 1, 2, ...
    on both ranks; jn_scan_allreduce (the one-call form) reduces across the ranks too."""
    rc, outs, _ = run_ranks(fake_rccl, tmp_path, "plain", [{"JN_TEST_SLOT_DELAY_US": "0,6000,0,3000"}, {"JN_TEST_SLOT_DELAY_US": "4000,0,0,1000"}])
    assert rc == [0, 0], outs
    r0, mb0, mm0, lb0, lm0 = load(tmp_path, 0)
    r1, mb1, mm1, lb1, lm1 = load(tmp_path, 1)
    assert r0["info"] == [0, 2, 0] and r1["info"] == [1, 2, 0]                         # the communicator really has two ranks, both on device 0
    assert r0["status"] == [0] * N and r1["status"] == [0] * N
    assert r0["order"] == list(range(N)) and r1["order"] == list(range(N)), (r0["order"], r1["order"])
    assert np.array_equal(mb0, mb1) and np.array_equal(mm0, mm1)                       # every rank holds the robot-level scan
    eb, em = merged_of(lb0, lm0, lb1, lm1)
    assert np.array_equal(mb0, eb) and np.array_equal(mm0, em)
    assert not np.array_equal(lb0, lb1)                                                # the ranks really had different rigs
    # the ranks' own scans against the oracle chain (a sample of the batches: the CPU oracle takes ~0.1 s a frame)
    spo = oracle.scan_params(W, H)
    luto = oracle.valid_lut(spo, W, H)
    for r, lb, lm in ((0, lb0, lm0), (1, lb1, lm1)):
        for k in (0, 5, 11):
            for t in range(B):
                L, R = oracle.synth_pair(W, H, 30 + 4 * k, 900 + 1000 * r + 7 * k + t)
                _, D1o, _ = oracle.process(oracle.params(0), L, R)
                bo, mo, _ = oracle.scan(spo, oracle.to_u8(D1o), luto)
                assert np.allclose(lb[k, t], bo, rtol=0, atol=SCAN_TOL) and np.allclose(lm[k, t], mo, rtol=0, atol=SCAN_TOL), (r, k, t)
    # jn_scan_allreduce across the two ranks: bins 100 + i (rank 0) against 101 - i (rank 1)
    i = np.arange(90)
    for rep in (r0, r1):
        db, dm = np.array(rep["direct"][0]), np.array(rep["direct"][1])
        assert np.array_equal(db, np.tile(np.minimum(100.0 + i, 101.0 - i), (B, 1)))
        assert np.array_equal(dm, np.tile([1.0, 5.0, 0.0, 8.0], (B, 1)))                # min, max, min, max of (1,5,0,7) and (2,4,2,8)


@pytest.mark.timeout(600)
def test_a_batch_failing_on_one_rank_does_not_strand_the_other(fake_rccl, tmp_path):
    """Rank 1's batch 5 dies before its kernels (JN_TEST_FAIL_SEQ).  It must still take its merge turn and feed the all-reduce the identity of
    MIN, or rank 0 — already inside collective 5 — waits for ever and every later collective pairs up wrongly: rank 1 reports the error for
    that batch alone, rank 0's batch 5 is its OWN scan (nothing from the peer), every other batch is the two-rank minimum on both ranks."""
    rc, outs, _ = run_ranks(fake_rccl, tmp_path, "fail", [{"JN_TEST_SLOT_DELAY_US": "0,2000,0,0"}, {"JN_TEST_FAIL_SEQ_MERGED": "5", "JN_TEST_SLOT_DELAY_US": "1000,0,3000,0"}])
    assert rc == [0, 0], outs
    r0, mb0, mm0, lb0, lm0 = load(tmp_path, 0)
    r1, mb1, mm1, lb1, lm1 = load(tmp_path, 1)
    assert r0["status"] == [0] * N
    assert r1["status"] == [0] * 5 + [JN_ERR_INTERNAL] + [0] * (N - 6)
    assert r0["order"] == list(range(N)) and r1["order"] == list(range(N))
    eb, em = merged_of(lb0, lm0, lb1, lm1)
    for k in range(N):
        if k == 5:
            assert np.array_equal(mb0[k], lb0[k]) and np.array_equal(mm0[k], lm0[k])     # the peer contributed +inf / -inf
        else:
            assert np.array_equal(mb0[k], eb[k]) and np.array_equal(mm0[k], em[k]), k
            assert np.array_equal(mb1[k], eb[k]) and np.array_equal(mm1[k], em[k]), k


@pytest.mark.timeout(600)
def test_a_rank_that_dies_mid_run_makes_its_peer_fail_fast_not_hang(fake_rccl, tmp_path):
    """Rank 1 exits (os._exit, batches in flight) once its batch 5 is done.  Rank 0's next merges find no partner: each must give up after
    JN_COMM_TIMEOUT_MS, abort the communicator and return JN_ERR_COMM — and so must every later scan batch, at once — instead of hanging.
    What completed before the death is right."""
    rc, outs, wall = run_ranks(fake_rccl, tmp_path, "kill", [{"JN_COMM_TIMEOUT_MS": "1500"}, {}], timeout=120)
    assert rc[1] == 0 and rc[0] == 0, outs
    rep = json.load(open(os.path.join(tmp_path, "report0.json")))
    st = rep["status"]
    assert all(s in (0, JN_ERR_COMM) for s in st), st
    assert st[:6] == [0] * 6, st                                                       # rank 1 took part in the first six merges for certain
    first_bad = st.index(JN_ERR_COMM)                                                  # (raises if rank 0 never noticed)
    assert all(s == JN_ERR_COMM for s in st[first_bad:]), st                           # dead from there on
    assert rep["elapsed"] < 30 and wall < 100, (rep["elapsed"], wall)                  # one time-out (1.5 s), not one per batch piling up for minutes
    mb0 = np.load(os.path.join(tmp_path, "merged_bins0.npy"))
    lb0 = np.load(os.path.join(tmp_path, "local_bins0.npy"))
    assert (mb0[:6] <= lb0[:6]).all()                                                  # merged = min(own, peer's)


@pytest.mark.timeout(900)
def test_eight_ranks_take_the_gpu_route_and_merge_in_order(fake_rccl, tmp_path):
    """what an 8-GPU job will run, before the driver's node does.  EIGHT rank processes on device 0 through the product: every
    rank a handle with its share of the CPU quota as `host_threads` (2: below the library's bar, so the triangulations run on the GPU and
    there is no host stage), 4 slots x 6 scan batches, the slots held up differently on every rank, the cross-rig merge as every batch's tail
    over an eight-rank communicator.  Every rank must have taken the GPU triangulation route with no batch handed back, issued its 24 merges
    in submission order, and hold the same robot-level bins: the element-wise MIN over the eight ranks' own scans (extrema: MIN / MAX)."""
    world, rounds = 8, 6
    n = S * rounds
    skews = ["0,6000,0,3000", "4000,0,0,1000", "0,0,5000,0", "2000,2000,0,0", "0,0,0,7000", "1000,3000,500,0", "0,0,0,0", "3000,0,3000,0"]
    env = [{"JN_TEST_SLOT_DELAY_US": skews[r], "JN_WORKER_ROUNDS": str(rounds), "JN_WORKER_HOST_THREADS": "2", "JN_COMM_TIMEOUT_MS": "120000", "JN_WORKER_ID_WAIT_S": "300", "JN_COMM_INIT_TIMEOUT_S": "300"} for r in range(world)]
    rc, outs, _ = run_ranks(fake_rccl, tmp_path, "many", env, timeout=600)
    assert rc == [0] * world, outs
    reps = [load(tmp_path, r) for r in range(world)]
    for r, (rep, mb, mm, lb, lm) in enumerate(reps):
        assert rep["info"] == [r, world, 0], rep["info"]
        assert rep["status"] == [0] * n, (r, rep["status"])
        assert rep["order"] == list(range(n)), (r, rep["order"])
        assert all(rt[0] == 1 and rt[1] == 0 for rt in rep["route"]), (r, rep["route"])     # GPU triangulation, nothing handed back to the host
    eb = np.minimum.reduce([x[3] for x in reps])
    em = reps[0][4].copy()
    em[..., 0::2] = np.minimum.reduce([x[4][..., 0::2] for x in reps]); em[..., 1::2] = np.maximum.reduce([x[4][..., 1::2] for x in reps])
    for r, (rep, mb, mm, lb, lm) in enumerate(reps):
        assert np.array_equal(mb, eb) and np.array_equal(mm, em), r
    assert len({x[3].tobytes() for x in reps}) == world                                  # eight different sets of rigs


@pytest.mark.timeout(1200)
def test_bench_dry_run_with_eight_ranks_on_one_gpu(fake_rccl):
    """`bench.py --gpus 8` as the driver will start it — eight rank processes, pinning plan, quota / 8 as `host_threads`, the merge as
    every batch's tail through the C-ABI communicator — on ONE GPU: --share-gpu, gloo for torch.distributed's rendezvous and barriers,
    the library's communicator bound to the stand-in RCCL.  The line must carry eight `ranks` entries that report an eight-rank
    communicator, a merge time, and no failed pair."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--dist-backend", "gloo", "--steps", "6", "--warmup", "2",
           "--width", "640", "--height", "480", "--disp", "64", "--batch", "2", "--slots", "2", "--min-time", "0", "--no-cpu-baseline", "--no-latency-config", "--no-alone-leg"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(JN_RCCL_LIB=fake_rccl, JN_COMM_INIT_TIMEOUT_S="300", JN_BENCH_STARTUP_TIMEOUT_S="600")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1100, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 8 and sorted(r["rank"] for r in j["ranks"]) == list(range(8))
    assert all(r["rccl_comm"] and r["rccl_comm"][1] == 8 for r in j["ranks"]), j["ranks"]
    assert j["merge"]["merge_ms_per_step"] is not None and j["merge"]["merge_ms_per_step"] > 0
    assert "jn_elas_set_comm" in j["merge"]["kind"]
    assert j["value"] > 0 and j["config"]["pairs_failed"] == 0
    assert j["config"]["host_threads"] <= 4                                       # a rank's share of the quota: the GPU triangulation route
