"""One rank of tests/test_gpu_comm_two_ranks.py (TEST INFRASTRUCTURE).  Runs S slots x `rounds` scan batches of ITS OWN rigs through the HIP
path twice — without a communicator (the rank's local scans) and with the product's communicator attached (jn_elas_set_comm: the merge
as every batch's tail, issued by the slot workers in submission order) — and leaves what it saw in out_dir.
    python comm_rank_worker.py <rank> <world> <out_dir> <scenario>      scenario: plain | fail | kill
JN_RCCL_LIB points at tests/mocks/fake_rccl.cpp's library: both ranks share device 0."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rank, world, out_dir, scenario = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
W, H, B, S = 320, 180, 2, 4
ROUNDS = int(os.environ.get("JN_WORKER_ROUNDS", "3"))             # batches per slot
HOST_THREADS = int(os.environ.get("JN_WORKER_HOST_THREADS", "4"))  # what bench.py passes a rank of an N-rank job: its share of the CPU quota
N = S * ROUNDS


def frame_seed(r, k, t):
    return 900 + 1000 * r + 7 * k + t


def main():
    import jackal_navigation_amd as jn
    from jackal_navigation_amd import node, parallel, _lib
    from jackal_navigation_amd.device import DeviceArray
    jn.load()
    sp = node.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H)
    pairs = [[node.synth_pair(W, H, 30 + 4 * k, frame_seed(rank, k, t)) for t in range(B)] for k in range(N)]
    dLs = [DeviceArray.from_numpy(np.stack([p[0] for p in ps])) for ps in pairs]
    dRs = [DeviceArray.from_numpy(np.stack([p[1] for p in ps])) for ps in pairs]

    def id_exchange(raw):
        path = os.path.join(out_dir, "comm_id.bin")
        if raw is not None:
            with open(path + ".tmp", "wb") as f:
                f.write(raw)
            os.replace(path + ".tmp", path)
            return raw
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > float(os.environ.get("JN_WORKER_ID_WAIT_S", "60")):
                raise RuntimeError("rank 0 never wrote the communicator id")
            time.sleep(0.01)
        return open(path, "rb").read()

    with jn.Elas(jn.Elas.parameters(0), W, H, max_batch=B, slots=S, host_threads=HOST_THREADS) as e:
        bufs = [dict(d1=DeviceArray.from_numpy(np.zeros((B, H, W), np.float32)), d2=DeviceArray.from_numpy(np.zeros((B, H, W), np.float32)),
                     u8=DeviceArray((B, H, W), np.uint8), bins=DeviceArray((B, sp.bins), np.float64), meta=DeviceArray((B, 4), np.float64),
                     st=(C.c_int32 * B)()) for _ in range(S)]

        def rounds():
            got = [None] * N

            def finish(s0, k0):
                e.wait(s0)
                got[k0] = (bufs[s0]["bins"].numpy().copy(), bufs[s0]["meta"].numpy().copy())
            inflight = []
            for k in range(N):
                slot = k % S
                if len(inflight) == S:
                    finish(*inflight.pop(0))
                b = bufs[slot]
                e.submit_scan(slot, B, dLs[k].ptr, dRs[k].ptr, W, H * W, b["d1"].ptr, b["d2"].ptr, sp, lut.ptr, b["u8"].ptr, b["bins"].ptr, b["meta"].ptr, b["st"])
                inflight.append((slot, k))
            for s0, k0 in inflight:
                finish(s0, k0)
            return got

        local = rounds()                                         # no communicator: this rank's own scans
        np.save(os.path.join(out_dir, "local_bins%d.npy" % rank), np.stack([g[0] for g in local]))
        np.save(os.path.join(out_dir, "local_meta%d.npy" % rank), np.stack([g[1] for g in local]))
    # the merged pass: a fresh handle (JN_TEST_FAIL_SEQ / JN_TEST_SLOT_DELAY_US are read when it is created), the two-rank communicator attached
    fail_seq = os.environ.get("JN_TEST_FAIL_SEQ_MERGED")
    if fail_seq is not None:
        os.environ["JN_TEST_FAIL_SEQ"] = fail_seq
    with jn.Elas(jn.Elas.parameters(0), W, H, max_batch=B, slots=S, host_threads=HOST_THREADS) as e:
        bufs = [dict(d1=DeviceArray.from_numpy(np.zeros((B, H, W), np.float32)), d2=DeviceArray.from_numpy(np.zeros((B, H, W), np.float32)),
                     u8=DeviceArray((B, H, W), np.uint8), bins=DeviceArray((B, sp.bins), np.float64), meta=DeviceArray((B, 4), np.float64),
                     st=(C.c_int32 * B)()) for _ in range(S)]
        comm = parallel.ScanComm(rank, world, 0, id_exchange)
        info = comm.info()
        e.set_comm(comm)
        t0 = time.time()

        got = [None] * N

        def finish(s0, k0):
            try:
                e.wait(s0)
                got[k0] = (bufs[s0]["bins"].numpy().copy(), bufs[s0]["meta"].numpy().copy())
            except _lib.JnError as err:
                got[k0] = int(err.status)
            if scenario == "kill" and rank == 1 and k0 == 5:
                os._exit(0)                                      # the rank dies here, batches in flight and all: no clean-up of any kind
        inflight = []
        for k in range(N):
            slot = k % S
            if len(inflight) == S:
                finish(*inflight.pop(0))
            b = bufs[slot]
            try:
                e.submit_scan(slot, B, dLs[k].ptr, dRs[k].ptr, W, H * W, b["d1"].ptr, b["d2"].ptr, sp, lut.ptr, b["u8"].ptr, b["bins"].ptr, b["meta"].ptr, b["st"])
                inflight.append((slot, k))
            except _lib.JnError as err:                           # a dead communicator refuses new scan batches
                got[k] = int(err.status)
        for s0, k0 in inflight:
            finish(s0, k0)
        elapsed = time.time() - t0
        order = e.merge_order()
        route = [list(e.route_stats(sl)) for sl in range(S)]      # per slot: [GPU triangulation route chosen, batches handed back to the host, plane flow]
        # the one-call form on both ranks (plain scenario): min over ranks of a buffer that differs per rank
        direct = None
        if scenario == "plain":
            db = DeviceArray.from_numpy(np.full((B, sp.bins), 100.0 + rank) + np.arange(sp.bins)[None, :] * (1 - 2 * rank))
            dm = DeviceArray.from_numpy(np.array([[1.0 + rank, 5.0 - rank, 2.0 * rank, 7.0 + rank]] * B))
            comm.merge(B, sp.bins, db.ptr, dm.ptr)
            direct = [db.numpy().tolist(), dm.numpy().tolist()]
        e.set_comm(None)
        comm.close()
    ok_bins = [g[0] if isinstance(g, tuple) else np.full((B, sp.bins), np.nan) for g in got]
    ok_meta = [g[1] if isinstance(g, tuple) else np.full((B, 4), np.nan) for g in got]
    np.save(os.path.join(out_dir, "merged_bins%d.npy" % rank), np.stack(ok_bins))
    np.save(os.path.join(out_dir, "merged_meta%d.npy" % rank), np.stack(ok_meta))
    json.dump({"status": [0 if isinstance(g, tuple) else g for g in got], "order": order, "elapsed": elapsed, "info": list(info), "direct": direct, "route": route},
              open(os.path.join(out_dir, "report%d.json" % rank), "w"))


main()
