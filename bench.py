#!/usr/bin/env python3
"""bench.py — stereo-pairs/s of the MI355X hot path (rectified pair -> ELAS disparity -> u8 map ->
reprojection -> 90-bin obstacle scan), 1280x720, disp range 128, batch 32 per GPU.

    python bench.py --gpus N --steps K --warmup W

One "step" = one batch of `--batch` synthetic pairs (SURVEY.md Appendix A generator, seed 12345+b)
through the whole path with inputs already resident in HBM.  Steps are pipelined over `--slots`
library slots (GPU stage A / host stage / GPU stage B of different batches overlap).

N > 1: one rank (process) per GPU.  Under torchrun the ranks are already there (RANK / WORLD_SIZE in
the environment); started plainly as `python bench.py --gpus N` this process spawns N fresh rank
processes itself BEFORE anything touches the GPU and relays rank 0's line.  Every rank processes its
own rigs (weak scaling); the per-step exchange is the element-wise MIN all-reduce of the scan bins
(+ extrema), the only cross-rig step the path has, issued through the library's C-ABI
(`jn_scan_allreduce`, RCCL over xGMI).  Each rank is pinned to its own physical cores on its GPU's
NUMA node (jackal_navigation_amd/parallel.py).

The timed region is EXACTLY `--steps` steps between barrier + synchronize pairs; it is repeated until
at least `--min-time` seconds have been timed and the MEDIAN region decides `value` (the spread is
reported).  Rank 0 prints ONE JSON line with `roofline`, `cpu_baseline` and `check` (FNV-1a-64 of D1 of
frame 0 against the reference's golden hash; a mismatch exits non-zero).
"""
import argparse
import ctypes as C
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before HIP initialises: one hardware queue per slot stream (see jackal_navigation_amd/__init__.py)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL between processes)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# algorithmic bytes per pixel per pair of each GPU stage of the ELAS path
STAGE_BYTES_PER_PX = {
    "gpu_descriptor": 6.0,      # sobel 2r+4w (descriptors not counted: an implementation choice)
    "gpu_support": 4.0,         # support match 4r
    "gpu_matching": 16.0,       # dense L+R 8r+8w
    "gpu_lr": 16.0,             # 8r+8w
    "gpu_speckle": 16.0,        # 8r+8w labels included
    "gpu_gap": 16.0,            # rows+cols 8r+8w
    "gpu_adaptive_mean": 16.0,  # H+V 8r+8w
}
def _evidence_round():
    """round of the committed evidence set (profiles/CURRENT, e.g. "r04" or "r03_i" -> "r03")"""
    try:
        return open(os.path.join(ROOT, "profiles", "CURRENT")).read().strip().split("_")[0]
    except OSError:
        return "r00"


PMC_FILE = os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % _evidence_round())
KERNELS_SRC = os.path.join(ROOT, "jackal_navigation_amd", "csrc", "kernels.hip")


def cpu_baseline_worker(args):
    """Times the CPU path on `count` pairs in this process (reference build if present, else the port)."""
    W, H, scene, disp, seed0, count = args
    from oracle.binding import Oracle, LocalReference as Reference   # timing only: heap state is irrelevant
    o = Oracle()
    ref = Reference() if Reference.available() else None
    p = o.params(disp_max=disp - 1)
    sp = o.scan_params(W, H)
    lut = o.valid_lut(sp, W, H)
    pairs = [o.synth_pair(W, H, scene, seed0 + i) for i in range(count)]
    t0 = time.perf_counter()
    for L, R in pairs:
        if ref is not None:
            D1, _ = ref.process(p, L, R)
        else:
            _, D1, _ = o.process(p, L, R)
        o.scan(sp, o.to_u8(D1), lut)
    return time.perf_counter() - t0, ("reference" if ref is not None else "port")


def cpu_baseline(W, H, scene, disp, budget_s=20.0):
    """Reference CPU path on a bounded sample, one process per core (Triangle is not thread-safe)."""
    import multiprocessing as mp
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)   # the CPUs this process may use
    procs = max(1, min(ncpu // 2, 64))
    # single-core probe first (also the figure quoted per core)
    t1, kind = cpu_baseline_worker((W, H, scene, disp, 12345, 2))
    per_pair = t1 / 2
    per_proc = max(1, min(8, int(budget_s / 2 / max(per_pair, 1e-3))))
    ctx = mp.get_context("fork")
    with ctx.Pool(procs) as pool:
        t0 = time.perf_counter()
        pool.map(cpu_baseline_worker, [(W, H, scene, disp, 12345 + 1000 * i, per_proc) for i in range(procs)])
        wall = time.perf_counter() - t0
    # wall includes per-process input generation and LUT build; use it as is (conservative for the CPU)
    total = procs * per_proc
    return {
        "value": round(total / wall, 2), "unit": "pairs/s", "cores": procs, "kind": kind,
        "sample": "%d pairs %dx%d scene disparities <= %d, disp_max=%d in %d processes (%.1f s wall); single core: %.2f pairs/s" %
                  (total, W, H, scene, disp - 1, procs, wall, 1.0 / per_pair),
    }


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mode", default="elas", choices=["elas", "sgm", "bm"],
                    help="elas (default): the reference's matcher, the headline metric; sgm: the 8-path SGM mode of include/jn_sgm.h "
                         "(no reference counterpart), same workload shape, roofline on SURVEY 8d's B_sgm; bm: the block matcher of include/jn_bm.h "
                         "(no reference counterpart; BASELINE config 2 is --mode bm --width 640 --height 480 --disp 64 --batch 1)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--disp", type=int, default=128, help="disparity range D (disp_max = D-1)")
    ap.add_argument("--scene-disp", type=int, default=0, help="largest disparity in the synthetic scene (default: D)")
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--min-time", type=float, default=1.0, help="repeat the timed region until this many seconds have been timed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency-config", action="store_true",
                    help="skip the informational 640x480 batch-1 leg (profiles then hold the headline workload's launches only)")
    ap.add_argument("--no-alone-leg", action="store_true", help="skip the un-pipelined leg that times k_dense_row running alone")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (=RCCL, default) or gloo; gloo + --share-gpu lets two ranks dry-run the N>1 path on one GPU")
    ap.add_argument("--merge", default="cabi", choices=["cabi", "torch"],
                    help="cross-rig merge with the nccl backend: the library's jn_scan_allreduce (default) or torch.distributed")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses device 0")
    ap.add_argument("--force-merge", action="store_true",
                    help="N=1 only: attach a ONE-rank RCCL communicator, so that every batch ends with the cross-rig merge it has on a multi-GPU node "
                         "(pack -> ncclAllReduce(MIN) -> unpack in the slot worker); the line then carries merge_ms_per_step and the rate without it")
    ap.add_argument("--no-pin", action="store_true", help="do not pin ranks to their GPU's NUMA node")
    ap.add_argument("--subpixel", type=int, default=0, help="sgm / bm mode: 1/16-pixel refinement")
    ap.add_argument("--block-radius", type=int, default=4, help="bm mode: block radius r (2, 3, 4)")
    ap.add_argument("--bm-slots", type=int, default=4, help="bm mode: batches in flight (jn_bm_submit_scan / jn_bm_wait), as --sgm-slots for the SGM mode")
    ap.add_argument("--sgm-slots", type=int, default=6,
                    help="sgm mode: batches in flight (six measured best: 4.70 / 4.76 / 4.95 / 4.95 / 4.83 k pairs/s with 4 / 5 / 6 / 7 / 8) (jn_sgm_submit_scan / jn_sgm_wait; 1 = one synchronous batch at a time as in rounds 2-3; each further "
                         "slot holds its own three W*H*D byte volumes per pair of the batch)")
    ap.add_argument("--bm-cost", default="sad", choices=["sad", "ssd"],
                    help="bm mode: sad = absolute differences (v_qsad kernel, default); ssd = squared differences as a banded int8 contraction on the matrix "
                         "cores (v_mfma_i32_32x32x32_i8, csrc/bm_mfma.hip; BASELINE config 5's \"int8 cost volume (CDNA4 MFMA path)\")")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this process has not touched the GPU
    and never will), wait for them, pass rank 0's line through.  Returns the exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    code = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0 and code == 0:
                    code = rc
                    for q in pending:          # one rank failed: the others would wait for it in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return code


def kernels_sha():
    try:
        return hashlib.sha256(open(KERNELS_SRC, "rb").read()).hexdigest()
    except OSError:
        return None


def bench_golden(W, H, scene, disp_max):
    """seed -> recorded reference answers for this configuration: all 32 seeds of the headline batch (bench_batch_golden.json), seed 12345
    only for the other recorded configurations (reference_hashes.txt + scan_golden.json)."""
    out = {}
    g = os.path.join(ROOT, "tests", "golden")
    for name in ("bench_batch_golden.json", "bench_vga_golden.json"):
        try:
            j = json.load(open(os.path.join(g, name)))
            if j.get("config") == [W, H, scene, disp_max]:
                out = {int(k): v for k, v in j.items() if k.isdigit()}
        except (OSError, ValueError):
            pass
    if 12345 not in out:
        h = golden_hash(W, H, scene, disp_max)
        if h:
            out[12345] = {"d1_fnv": h}
            try:
                sg = json.load(open(os.path.join(g, "scan_golden.json"))).get("%d %d %d %d" % (W, H, scene, disp_max))
                if sg:
                    out[12345].update({"u8_fnv": sg["u8_fnv"], "bins": sg["bins"], "meta": sg["meta"]})
            except (OSError, ValueError):
                pass
    return out


def golden_hash(W, H, scene, disp_max):
    """D1 hash of the reference on the Appendix-A pair (seed 12345) for this configuration, if one was recorded."""
    try:
        for line in open(os.path.join(ROOT, "tests", "golden", "reference_hashes.txt")):
            f = line.split()
            if len(f) >= 5 and not line.startswith("#") and (int(f[0]), int(f[1]), int(f[2]), int(f[3])) == (W, H, scene, disp_max):
                return f[4]
    except OSError:
        pass
    return None


def sgm_cpu_worker(args):
    W, H, scene, D, sub, seed0, count, radius = args
    from oracle.binding import Oracle, SgmOracle, BmOracle
    o, s = Oracle(), SgmOracle()
    b = BmOracle() if radius else None
    sp = o.scan_params(W, H)
    lut = o.valid_lut(sp, W, H)
    pairs = [o.synth_pair(W, H, scene, seed0 + i) for i in range(count)]
    t0 = time.perf_counter()
    for L, R in pairs:
        d = b.process(b.params(D, radius, subpixel=sub), L, R) if b else s.process(s.params(D, subpixel=sub), L, R)
        o.scan(sp, s.to_u8(d, sub), lut)
    return time.perf_counter() - t0


def sgm_cpu_baseline(W, H, scene, D, sub, radius=0):
    """The SGM (radius 0) or block-matching mode's scalar definition (oracle/sgm_oracle.cpp / bm_oracle.cpp, kind "port": the
    reference has neither) on a bounded sample."""
    import multiprocessing as mp
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = max(1, min(ncpu // 2, 32))
    t1 = sgm_cpu_worker((W, H, scene, D, sub, 12345, 1, radius))
    per_proc = max(1, min(4, int(10.0 / max(t1, 1e-3))))
    with mp.get_context("fork").Pool(procs) as pool:
        t0 = time.perf_counter()
        pool.map(sgm_cpu_worker, [(W, H, scene, D, sub, 12345 + 1000 * i, per_proc, radius) for i in range(procs)])
        wall = time.perf_counter() - t0
    return {"value": round(procs * per_proc / wall, 2), "unit": "pairs/s", "cores": procs, "kind": "port",
            "sample": "%d pairs %dx%d D=%d in %d processes (%.1f s wall) of the scalar definition oracle/%s_oracle.cpp; single core: %.2f pairs/s"
                      % (procs * per_proc, W, H, D, procs, wall, "bm" if radius else "sgm", 1.0 / t1)}


def bm_ssd_roofline(W, H, D, B, radius, ms):
    """The squared-difference block matcher on the matrix cores: bound "mfma" in the contract's vocabulary.  Algorithmic work = the cross
    terms the definition needs, 2 (2r+1)^2 W H D multiply-adds... per pair and side; what the kernel ISSUES is one 32x32x32 int8 MFMA per 32 x 32
    tile of (candidate, column) pairs and row (both the entering and the leaving row of the window in its K = 32), i.e. 65 536 int8 operations
    for 2 (2r+1) x 2 x 1024 useful ones.  `achieved` is the ISSUED rate (what the matrix cores did) against the dense int8 peak; `useful` next to
    it.  The pass is bound by vector issue (keys and minima), not by the matrix cores: see DESIGN.md 4c."""
    sides = 2
    tiles = ((W + 31) // 32) * (D // 32 + 1) * H * B * sides
    issued = tiles * 2.0 * 32 * 32 * 32
    useful = 2.0 * (2 * radius + 1) ** 2 * W * H * D * B * sides
    t = ms["match"] * 1e-3
    peak = 5000.0                                              # TOP/s dense int8 (MI355X_MICROARCH.md; the 2:1-sparsity figure is not used)
    return {"bound": "mfma", "kernel": "k_bmq_match x2 (v_mfma_i32_32x32x32_i8; + prefilter, patch norms, finish: one batch)", "achieved": round(issued / t / 1e12, 2),
            "peak": peak, "unit": "TOP/s", "frac": round(issued / t / 1e12 / peak, 4), "traffic": None, "useful_TOPs": round(useful / t / 1e12, 2),
            "ms_per_launch": round(ms["match"] / sides, 4), "ms_per_batch_all_kernels": round(ms["total"], 4),
            "note": "issued int8 operations of the matching passes / their GPU time; useful = the definition's 2 (2r+1)^2 W H D per side.  Bound by vector issue "
                    "(one shift-add and half a min3 per candidate pair), the matrix cores are ~1/4 busy"}


def bm_roofline(W, H, D, B, radius, sub, ms):
    """Block matching streams ~13 bytes per pixel (2 in, a 4-byte winner record per side, 2 out, u8 map + LUT + scan) and spends
    W H D block costs of (2r+1)^2 absolute differences on them: the bound is vector issue, not HBM (PMC, DESIGN.md 4c: the VALU
    pipes are busy for the whole duration of the match kernels).  The contract's roofline is the HBM one — algorithmic bytes
    over the GPU time of one batch — and is small by construction; the rate of absolute differences sits next to it."""
    alg = (2.0 + 2.0 + 5.0) * W * H * B
    achieved = alg / (ms["total"] * 1e-3) / 1e9
    sides = 2
    ads = float((2 * radius + 1) ** 2) * W * H * D * B * sides
    return {"bound": "hbm", "kernel": "k_bm x2 (+ k_bm_prefilter, k_bm_finish: one batch)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "ms_per_launch": round(ms["match"] / sides, 4), "ms_per_batch_all_kernels": round(ms["total"], 4),
            "algorithmic_bytes_per_launch": int(alg),
            "note": "bound by vector issue, not HBM (no cost volume exists): %.1f T absolute differences/s in the match kernels" % (ads / (ms["match"] * 1e-3) / 1e12)}


def sgm_pmc_traffic(W, H, D, B):
    """HBM bytes per batch of the sweep kernels from the committed PMC passes (profiles/<round>_sgm_pmc_traffic.json), if they were
    taken on this source (sha256 of sgm_sweep.hip) and this workload."""
    try:
        name = "%s_sgm_pmc_traffic.json" % _evidence_round()
        j = json.load(open(os.path.join(ROOT, "profiles", name)))
        src = os.path.join(ROOT, "jackal_navigation_amd", "csrc", "sgm_sweep.hip")
        if j.get("sgm_sweep_sha256") != hashlib.sha256(open(src, "rb").read()).hexdigest() or j.get("workload") != [W, H, D, B]:
            return None
        return {"bytes": int(j["bytes_per_batch"]), "note": "PMC (FETCH_SIZE x 2 + WRITE_SIZE, separate passes, profiles/%s);" % name}
    except (OSError, ValueError, KeyError):
        return None


def run_sgm(a):
    """--mode sgm: step = one batch through prefilter -> 8 paths -> sum/WTA/check -> u8 map -> 90-bin scan.
    --mode bm: the same with the block matcher (prefilter -> left / right block costs + WTA -> check) in place of the paths."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", str(rank)))
    W, H, B, D = a.width, a.height, a.batch, a.disp
    scene = a.scene_disp or a.disp
    bm = a.mode == "bm"
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = sgm_cpu_baseline(W, H, scene, D, a.subpixel, a.block_radius if bm else 0)
    import torch
    import jackal_navigation_amd as jn
    from jackal_navigation_amd import node
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    on_gpu = a.dist_backend == "nccl"
    Ls = np.empty((B, H, W), np.uint8); Rs = np.empty((B, H, W), np.uint8)
    for b in range(B):
        Ls[b], Rs[b] = node.synth_pair(W, H, scene, 12345 + b + 1000 * rank)
    dL = torch.from_numpy(Ls).to(dev); dR = torch.from_numpy(Rs).to(dev)
    disp = torch.zeros((B, H, W), dtype=torch.int16, device=dev)
    u8 = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
    bins = torch.zeros((B, 90), dtype=torch.float64, device=dev); meta = torch.zeros((B, 4), dtype=torch.float64, device=dev)
    # SGM: batches pipelined over --sgm-slots slots (jn_sgm_submit_scan / jn_sgm_wait), each with its own outputs and its own copy of the inputs
    SS = max(1, min(6 if bm else 8, a.bm_slots if bm else a.sgm_slots))     # (the block matcher pipelines the same way: jn_bm_submit_scan / jn_bm_wait)
    slot_in = [(dL, dR)] + [(dL.clone(), dR.clone()) for _ in range(SS - 1)]
    slot_out = [(disp, u8, bins, meta)] + [(torch.zeros_like(disp), torch.zeros_like(u8), torch.zeros_like(bins), torch.zeros_like(meta)) for _ in range(SS - 1)]
    if bm:
        sgm = jn.Bm(jn.Bm.parameters(num_disparities=D, block_radius=a.block_radius, subpixel=a.subpixel, cost_function=1 if a.bm_cost == "ssd" else 0),
                    W, H, max_batch=B, device=local_rank)
    else:
        sgm = jn.Sgm(jn.Sgm.parameters(num_disparities=D, subpixel=a.subpixel), W, H, max_batch=B, device=local_rank)
    sp = node.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H, device=local_rank)
    torch.cuda.synchronize()
    acc = {}

    def step():
        if bm and SS == 1:                                   # matcher + u8 map + scan on one stream (jn_bm_process_scan), one batch at a time
            sgm.process_scan(B, dL.data_ptr(), dR.data_ptr(), W, H * W, disp.data_ptr(), sp, lut.ptr, u8.data_ptr(), bins.data_ptr(), meta.data_ptr())
            for k, v in sgm.last_times().items():
                acc.setdefault(k, []).append(v)
            return
        if SS == 1:                                          # one batch at a time: three synchronous calls
            sgm.process_batch(B, dL.data_ptr(), dR.data_ptr(), W, H * W, disp.data_ptr())
            for k, v in sgm.last_times().items():
                acc.setdefault(k, []).append(v)
            sgm.to_u8(disp.data_ptr(), u8.data_ptr(), B * H * W)
            node.obstacle_scan(sp, B, u8.data_ptr(), lut.ptr, W, H, bins.data_ptr(), meta.data_ptr(), device=local_rank)
            return
        slot = step.count % SS
        step.count += 1
        if slot in step.inflight:                            # the slot's previous batch first
            sgm.wait(slot); step.inflight.discard(slot)
            for k, v in sgm.last_times().items():
                acc.setdefault(k, []).append(v)
        (iL, iR), (od, ou, ob, om) = slot_in[slot], slot_out[slot]
        sgm.submit_scan(slot, B, iL.data_ptr(), iR.data_ptr(), W, H * W, od.data_ptr(), sp, lut.ptr, ou.data_ptr(), ob.data_ptr(), om.data_ptr())
        step.inflight.add(slot)
    step.count = 0
    step.inflight = set()

    def drain():
        for slot in sorted(step.inflight):
            sgm.wait(slot)
            for k, v in sgm.last_times().items():          # (a region shorter than the slots never comes back to a slot: its batches' times are read here)
                acc.setdefault(k, []).append(v)
        step.inflight.clear()

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def region():
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        drain()                                              # every batch of the region complete inside the region
        sync()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(max(a.warmup, SS)):                       # (every slot once before the timed regions: a slot allocates its volumes when it is first used)
        step()
    drain()
    acc.clear()
    regions = [region()]
    for _ in range(max(1, min(200, int(math.ceil(a.min_time / max(regions[0], 1e-6))))) - 1):
        regions.append(region())
    elapsed = float(np.median(regions))
    value = world * B * a.steps / elapsed
    if bm and SS > 1:                                        # the kernels' own durations for the roofline: a few batches alone, outside the timed regions
        acc.clear()
        for i in range(10):
            sgm.process_scan(B, dL.data_ptr(), dR.data_ptr(), W, H * W, disp.data_ptr(), sp, lut.ptr, u8.data_ptr(), bins.data_ptr(), meta.data_ptr())
            if i >= 2:
                for k, v in sgm.last_times().items():
                    acc.setdefault(k, []).append(v)
    ms = {k: float(np.mean(v)) for k, v in acc.items()}
    check = None
    if rank == 0:
        want = None
        try:
            for line in open(os.path.join(ROOT, "tests", "golden", ("bm_ssd_hashes.txt" if a.bm_cost == "ssd" else "bm_hashes.txt") if bm else "sgm_hashes.txt")):
                f = line.split()
                key = [W, H, scene, D, a.block_radius, a.subpixel, 12345] if bm else [W, H, scene, D, a.subpixel, 12345]
                if not line.startswith("#") and len(f) == len(key) + 1 and [int(x) for x in f[:-1]] == key:
                    want = f[-1]
        except OSError:
            pass
        host = disp[0].cpu().numpy()
        got = "%016x" % jn.load().jn_fnv1a64_u32(host.ctypes.data, host.size // 2)
        check = {"what": "FNV-1a-64 of the int16 disparity map of frame 0 (seed 12345) after the timed region", "got": got, "expected": want,
                 "source": ("tests/golden/%s_hashes.txt (scalar definition oracle/%s_oracle.cpp; self-referential, the reference has no such matcher)"
                            % (("bm", "bm") if bm else ("sgm", "sgm"))) if want else None,
                 "ok": (got == want) if want else None}
    # roofline: SURVEY 8d's algorithmic bytes of the cost-volume mode, B_sgm = 4 W H D + 5 W H per pair, over the GPU time of
    # one batch (HIP events on the library's stream).  `traffic` = the bytes this decomposition really moves (PMC file, or the
    # three byte volumes written and read once).
    b_sgm = (4.0 * W * H * D + 5.0 * W * H) * B
    if bm:
        roofline = bm_ssd_roofline(W, H, D, B, a.block_radius, ms) if a.bm_cost == "ssd" else bm_roofline(W, H, D, B, a.block_radius, a.subpixel, ms)
    if not bm:
        # one batch at a time: the batch's own GPU time (HIP events); pipelined: batches overlap, a batch's events then span the other
        # batches' kernels too, so the time a batch COSTS is the step time
        batch_ms = ms["total"] if SS == 1 else elapsed / a.steps * 1e3
        achieved = b_sgm / (batch_ms * 1e-3) / 1e9
        moved = (6.0 * W * H * D + 2.0 * W * H * 2 + 2.0 * W * H) * B
        pmc = sgm_pmc_traffic(W, H, D, B)
        roofline = {"bound": "hbm", "kernel": "k_sw_h + k_sw_w<down> + k_sw_w<up, winners> (+ k_sw_prefilter, the L/R check: one batch)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": pmc["bytes"] if pmc else int(moved),
                    "traffic_note": (pmc["note"] if pmc else "computed, not PMC: 3 byte volumes of W H D written (two horizontal sweeps, the downward sweep) and read once by the upward sweep; ") +
                                    " moved bytes / time = %.0f GB/s" % ((pmc["bytes"] if pmc else moved) / (batch_ms * 1e-3) / 1e9),
                    "ms_per_launch": round(ms["paths"], 4), "ms_per_batch_all_kernels": round(ms["total"], 4), "ms_per_batch_pipelined": round(batch_ms, 4), "slots": SS,
                    "algorithmic_bytes_per_launch": int(b_sgm),
                    "bound_note": "the sweeps are integer-VALU bound, not HBM bound: ~100 packed 16-bit wave-instructions per pixel against the chip's "
                                  "~540 G wave-instructions/s (scripts/probes/valu_rate_probe.hip); frac is still quoted on SURVEY 8d's byte count"}
    if rank == 0:
        out = {"metric": "stereo_pairs_per_sec", "value": round(value, 1), "unit": "pairs/s", "n_gpus": (dist.get_world_size() if dist is not None else 1),
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "ms_per_frame": round(elapsed / (B * a.steps) * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/i16", "data": "synthetic",
               "config": {"workload": "%dx%d rectified pairs (scene disparities <= %d), %s D=%d%s (include/jn_%s.h; no reference counterpart), batch=%d per GPU -> u8 map -> 90-bin scan"
                                      % (W, H, scene, ("block matching %dx%d" % (2 * a.block_radius + 1, 2 * a.block_radius + 1)) if bm else "SGM 8 paths", D,
                                         " + 1/16 px" if a.subpixel else "", "bm" if bm else "sgm", B), "batch_per_gpu": B, "mode": a.mode,
                          "parallelism": "rigs sharded 1 batch/GPU over %d ranks" % world if world > 1 else "single GPU"},
               "timing": {"regions": len(regions), "region_s_median": round(elapsed, 5), "region_s_min": round(min(regions), 5), "region_s_max": round(max(regions), 5)},
               "stage_ms_per_batch": {k: round(v, 3) for k, v in ms.items()}, "roofline": roofline, "cpu_baseline": cpu, "check": check}
        print(json.dumps(out))
        sys.stdout.flush()
    sgm.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and check and check["ok"] is False:
        raise SystemExit("bench.py: the %s disparity map differs from its golden hash: %s" % (a.mode, check))


def thread_cpu_seconds():
    """user + system seconds of every thread of this process, keyed by the thread's name (/proc/self/task/*/comm; the library names
    its threads jn-pool and jn-slot)."""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                name = open("/proc/self/task/%s/comm" % tid).read().strip()
                f = open("/proc/self/task/%s/stat" % tid).read().rsplit(")", 1)[1].split()
                out.setdefault(name, [0.0, 0])
                out[name][0] += (int(f[11]) + int(f[12])) / tick
                out[name][1] += 1
            except OSError:
                pass
    except OSError:
        pass
    return out


def cgroup_cpu_stat():
    try:
        d = {l.split()[0]: int(l.split()[1]) for l in open("/sys/fs/cgroup/cpu.stat")}
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        d["quota_cpus"] = None if q[0] == "max" else int(q[0]) / int(q[1])
        return d
    except (OSError, ValueError, IndexError):
        return None


def host_cpu_report(cpu0, cg0, wall):
    """Host cores the timed regions used, by thread role, next to the container's CPU quota (cgroup v2 cpu.max) and how often the
    kernel throttled the container meanwhile — the host stage (Delaunay) is the path's CPU consumer."""
    cpu1, cg1 = thread_cpu_seconds(), cgroup_cpu_stat()
    roles = {}
    for name, (sec, cnt) in cpu1.items():
        d = sec - cpu0.get(name, [0.0, 0])[0]
        if d > 0.005 * wall:
            roles[name] = {"threads": cnt, "cores": round(d / wall, 2)}
    rep = {"wall_s": round(wall, 3), "cores_by_thread_name": dict(sorted(roles.items(), key=lambda kv: -kv[1]["cores"])),
           "cores_total": round(sum(v["cores"] for v in roles.values()), 2)}
    if cg0 and cg1:
        rep["cgroup"] = {"quota_cpus": cg1.get("quota_cpus"), "cores_used": round((cg1["usage_usec"] - cg0["usage_usec"]) / 1e6 / wall, 2),
                         "periods": cg1.get("nr_periods", 0) - cg0.get("nr_periods", 0),
                         "periods_throttled": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                         "throttled_thread_seconds": round((cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)) / 1e6, 3)}
    return rep


def run_rank(a):
    if a.mode in ("sgm", "bm"):
        return run_sgm(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; start it as `python bench.py --gpus N` or under "
                         "torch.distributed.run with --nproc-per-node N" % (a.gpus, world))
    W, H, B, S = a.width, a.height, a.batch, a.slots
    scene = a.scene_disp or a.disp
    started = start_watchdog(a)

    # CPU baseline first: it forks worker processes, which must happen before this process touches the GPU.
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(W, H, scene, a.disp)

    import torch
    import jackal_navigation_amd as jn
    from jackal_navigation_amd import node, parallel, _lib

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if a.share_gpu:
        local_rank = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    on_gpu = a.dist_backend == "nccl"      # gloo reduces CPU tensors

    # host cores of this rank: whole physical cores on the GPU's NUMA node, split between the ranks that share the node;
    # set before the library starts its slot workers and Delaunay pool (threads inherit the mask)
    pin = None
    if not a.no_pin and hasattr(os, "sched_setaffinity"):
        pin = parallel.pin_rank(rank, world, [0] * world if a.share_gpu else list(range(world)))
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # Pool threads: 16 per GPU keep the host stage hidden behind the kernels of the other slots (14 of them are busy at
    # 20 k pairs/s).  What a rank may really use is the smaller of its affinity mask and its share of the container's CPU
    # quota (cgroup cpu.max; the GPU boxes give 16 CPUs per GPU): a pool larger than that gets the whole container
    # throttled, slot workers included.
    cg = cgroup_cpu_stat()
    quota = cg.get("quota_cpus") if cg else None
    cpu_share = min(ncpu, quota / world) if quota else ncpu
    host_threads = a.host_threads or (16 if cpu_share >= 16 else max(2, int(cpu_share) - 1))

    # synthetic batch of this rank, resident in HBM: one DISTINCT copy per slot (the same B pairs, rotated by slot * B / S frames), so that
    # no two batches in flight read the same bytes and frame i of one slot is not frame i of another
    Ls = np.empty((B, H, W), np.uint8); Rs = np.empty((B, H, W), np.uint8)
    for b in range(B):
        Ls[b], Rs[b] = node.synth_pair(W, H, scene, 12345 + b + 1000 * rank)
    rot = [(s_ * max(1, B // S)) % B for s_ in range(S)]
    dLs = [torch.from_numpy(np.roll(Ls, -rot[s_], axis=0)).to(dev) for s_ in range(S)]
    dRs = [torch.from_numpy(np.roll(Rs, -rot[s_], axis=0)).to(dev) for s_ in range(S)]

    def seed_of(slot, i):
        """generator seed of frame i of a slot's batch"""
        return 12345 + (i + rot[slot]) % B + 1000 * rank
    D1 = [torch.zeros((B, H, W), dtype=torch.float32, device=dev) for _ in range(S)]
    D2 = [torch.zeros((B, H, W), dtype=torch.float32, device=dev) for _ in range(S)]
    U8 = [torch.zeros((B, H, W), dtype=torch.uint8, device=dev) for _ in range(S)]
    scans = [parallel.ScanBuffer(B, 90, dev) for _ in range(S)]            # bins + extrema of a batch in one buffer
    bins = [s.bins for s in scans]; meta = [s.meta for s in scans]
    status = [(C.c_int32 * B)() for _ in range(S)]

    p = jn.Elas.parameters(jn.Elas.ROBOTICS, disp_max=a.disp - 1)          # point_cloud.cpp:416-417 + D
    elas = jn.Elas(p, W, H, max_batch=B, device=local_rank, host_threads=host_threads, slots=S)
    sp = node.scan_params(W, H)
    lut = node.build_valid_disp_lut(sp, W, H, device=local_rank)
    torch.cuda.synchronize()

    # the cross-rig merge: the library's own RCCL communicator (C-ABI), torch.distributed as the alternative
    comm, merge_kind, comm_info = None, None, None
    if dist is not None:
        # the C-ABI communicator needs device buffers, not an NCCL process group: with --dist-backend gloo it is still used when JN_RCCL_LIB
        # names the library to bind (the dry run of an N-rank job on one GPU: tests/mocks/fake_rccl.cpp, --share-gpu)
        if (on_gpu or os.environ.get("JN_RCCL_LIB")) and a.merge == "cabi":
            def exchange(raw):
                t = torch.zeros(128, dtype=torch.uint8, device=dev if on_gpu else "cpu")
                if raw is not None:
                    t.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
                dist.broadcast(t, src=0)
                return bytes(t.cpu().numpy().tobytes())
            try:
                comm = parallel.ScanComm(rank, world, local_rank, exchange)
                comm_info = comm.info()
                merge_kind = "jn_scan_allreduce (C-ABI, RCCL ncclAllReduce MIN, one packed buffer per batch)"
            except _lib.JnError as e:
                print("bench.py rank %d: %s; falling back to torch.distributed for the merge" % (rank, e), file=sys.stderr)
                comm = None
            ok = torch.tensor([1 if comm is not None else 0], device=dev if on_gpu else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)                      # all ranks take the same route
            if int(ok.item()) == 0 and comm is not None:
                comm.close(); comm = None
        if comm is None:
            merge_kind = "torch.distributed all_reduce(MIN) (%s), one packed buffer per batch" % a.dist_backend
    elif a.force_merge and on_gpu:
        comm = parallel.ScanComm(0, 1, local_rank, lambda raw: raw)
        comm_info = comm.info()
        merge_kind = "jn_elas_set_comm with a ONE-rank communicator (--force-merge): the merge's own cost, no peer"
    if comm is not None:
        # the merge is the batch's tail, queued by the slot's worker (jn_elas_set_comm): jn_elas_wait returns with robot-level bins
        elas.set_comm(comm)
        if dist is not None:
            merge_kind = "jn_elas_set_comm (C-ABI): pack -> RCCL ncclAllReduce(MIN) -> unpack queued by the slot worker behind the scan, one packed buffer per batch"

    stage_acc = {}
    dense_ms = []
    owner_ms = []
    merge_ms = []
    merge_state = {"attached": comm is not None}

    def finish(slot):
        """Tail of a batch: wait for ELAS + u8 map + scan, then the cross-rig MIN reduce."""
        elas.wait(slot)                           # ELAS and the node's tail (u8 map + scan) ran on the slot's stream
        for k, v in elas.last_times(slot).items():
            stage_acc.setdefault(k, []).append(v)
        dense_ms.append(elas.kernel_time(slot)[0])
        owner_ms.append(elas.kernel_time(slot, b"k_owner")[0])
        if merge_state["attached"]:               # merged already: the batch ended with the all-reduce (jn_elas_set_comm)
            merge_ms.append(elas.merge_time(slot))
            return
        if dist is None:
            return
        # the path's one exchange step: robot-level scan = MIN over rigs, one all-reduce per batch; it completes
        # before the slot is handed a new batch (the next batch's kernels write the same bins)
        if on_gpu:
            scans[slot].merge()
            torch.cuda.current_stream().synchronize()
        else:
            host = parallel.ScanBuffer(B, 90, "cpu")
            host.flat.copy_(scans[slot].flat)
            host.merge()
            scans[slot].flat.copy_(host.flat)
            torch.cuda.current_stream().synchronize()

    def run(steps, depth=S):
        inflight = []
        for i in range(steps):
            slot = i % depth
            if len(inflight) == depth:
                finish(inflight.pop(0))
            elas.submit_scan(slot, B, dLs[slot].data_ptr(), dRs[slot].data_ptr(), W, H * W, D1[slot].data_ptr(), D2[slot].data_ptr(), sp, lut.ptr,
                             U8[slot].data_ptr(), bins[slot].data_ptr(), meta[slot].data_ptr(), status[slot])
            inflight.append(slot)
        while inflight:
            finish(inflight.pop(0))

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region():
        """EXACTLY a.steps steps between barrier + synchronize pairs; MAX over ranks."""
        sync()
        t0 = time.perf_counter()
        run(a.steps)
        sync()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    run(a.warmup)
    sync()
    started.set()                                 # through start-up: the watchdog stands down
    stage_acc.clear()
    del dense_ms[:]
    cpu0, cg0, t_cpu0 = thread_cpu_seconds(), cgroup_cpu_stat(), time.perf_counter()
    regions = [timed_region()]                    # identical on every rank (max-reduced), so the repeat count agrees
    repeats = max(1, min(200, int(math.ceil(a.min_time / max(regions[0], 1e-6)))))
    for _ in range(repeats - 1):
        regions.append(timed_region())
    host_cpu = host_cpu_report(cpu0, cg0, time.perf_counter() - t_cpu0)
    elapsed = float(np.median(regions))

    failed = sum(1 for s in status for x in s if x != 0)
    pairs = world * B * a.steps
    value = pairs / elapsed
    stage_ms = {k: float(np.mean(v)) for k, v in stage_acc.items()}
    k_ms_pipelined = float(np.mean(dense_ms))

    # --force-merge: the same timed region once more with the communicator detached = what the merge costs the pipeline
    merge_report = None
    if merge_state["attached"]:
        merge_report = {"kind": merge_kind, "merge_ms_per_step": round(float(np.mean(merge_ms)), 4) if merge_ms else None,
                        "what": "slot worker's clock: scan complete -> merged bins in place (its turn in the submission order, ncclAllReduce MIN in place on the buffer the scan kernel packed, unpack, "
                                "one host wait); runs as the batch's tail in the slot worker, overlapped with the other slots"}
        if a.force_merge and dist is None:
            elas.set_comm(None); merge_state["attached"] = False
            run(a.warmup)
            without = [timed_region() for _ in range(min(5, len(regions)))]
            merge_report["pairs_per_sec_without_merge"] = round(pairs / float(np.median(without)), 1)
            merge_report["pairs_per_sec_with_merge"] = round(value, 1)
            merge_report["cost_frac"] = round(1.0 - value / (pairs / float(np.median(without))), 4)
            stage_acc.clear(); run(a.warmup)           # stage times of the un-merged pipeline again, as in a plain run
            stage_ms = {k: float(np.mean(v)) for k, v in stage_acc.items()}

    # k_dense_row running ALONE (one batch in flight, kernels back to back): what the roofline fraction is computed from
    k_ms_alone = owner_ms_alone = None
    if not a.no_alone_leg:
        del dense_ms[:]; del owner_ms[:]
        run(6, depth=1)
        k_ms_alone = float(np.mean(dense_ms[1:]))
        owner_ms_alone = float(np.mean(owner_ms[1:]))
    sync()

    # what was timed is what the reference computes: EVERY frame of EVERY slot after the timed region against what the compiled reference
    # (+ the oracle's node tail) made of the same pair — tests/golden/bench_batch_golden.json for the headline batch (seeds 12345 .. 12376),
    # frame 0's recorded hashes (tests/golden/reference_hashes.txt, scan_golden.json) for the other configurations
    check = None
    if rank == 0:
        L = jn.load()
        gold = bench_golden(W, H, scene, a.disp - 1)
        merged = dist is not None or merge_state["attached"]          # bins after a cross-rig merge are not one rig's bins
        n_d1 = n_u8 = n_scan = 0
        bad, worst = [], 0.0
        for s_ in range(S):
            d1_host, u8_host = D1[s_].cpu().numpy(), U8[s_].cpu().numpy()
            bins_host, meta_host = bins[s_].cpu().numpy(), meta[s_].cpu().numpy()
            for i in range(B):
                g = gold.get(seed_of(s_, i))
                if g is None:
                    continue
                got = "%016x" % L.jn_fnv1a64_u32(d1_host[i].ctypes.data, d1_host[i].size)
                n_d1 += 1
                if got != g["d1_fnv"]:
                    bad.append({"slot": s_, "frame": i, "seed": seed_of(s_, i), "what": "D1", "got": got, "expected": g["d1_fnv"]})
                if "u8_fnv" in g:
                    got = "%016x" % L.jn_fnv1a64_u32(u8_host[i].ctypes.data, u8_host[i].size // 4)
                    n_u8 += 1
                    if got != g["u8_fnv"]:
                        bad.append({"slot": s_, "frame": i, "seed": seed_of(s_, i), "what": "u8 map", "got": got, "expected": g["u8_fnv"]})
                if "bins" in g and not merged:
                    diff = max(float(np.abs(bins_host[i] - np.array(g["bins"])).max()), float(np.abs(meta_host[i] - np.array(g["meta"])).max()))
                    worst = max(worst, diff); n_scan += 1
                    if not diff <= 1e-4:
                        bad.append({"slot": s_, "frame": i, "seed": seed_of(s_, i), "what": "scan", "max_abs_diff": diff})
        check = {"what": "after the timed region, every frame of every slot that has a recorded reference answer: FNV-1a-64 of D1 (bit-exact), of the u8 map (bit-exact), "
                         "90 bins + 4 extrema within 1e-4 (north star)",
                 "slots": S, "frames_per_slot": B, "distinct_input_batch_per_slot": True, "frames_checked": {"D1": n_d1, "u8_map": n_u8, "scan": n_scan},
                 "scan_max_abs_diff": worst if n_scan else None, "scan_tolerance": 1e-4, "mismatches": bad[:8], "n_mismatches": len(bad),
                 "source": "tests/golden/bench_batch_golden.json, reference_hashes.txt, scan_golden.json: compiled reference src/elas; node tail = oracle/node_oracle.cpp on the reference's D1 "
                           "(OpenCV / ROS side by definition)",
                 "ok": (len(bad) == 0) if n_d1 else None}

    # roofline of the dominant kernel, k_dense_row: algorithmic bytes per launch (dense L+R = 16 B per pixel per
    # pair, one launch = the whole batch, both sides) over its average duration, measured with HIP events the library
    # records around the kernel on the stream it runs on.
    alg_bytes = STAGE_BYTES_PER_PX["gpu_matching"] * W * H * B
    k_ms = k_ms_alone if k_ms_alone else k_ms_pipelined
    if not k_ms or k_ms <= 0:                # JN_STAGE_EVENTS=0: the library recorded no events around the kernel — nothing to price
        raise SystemExit("bench.py: the library recorded no kernel times (JN_STAGE_EVENTS=0?): the roofline object needs them; unset the switch")
    if not k_ms_pipelined or k_ms_pipelined <= 0:
        k_ms_pipelined = k_ms
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic, traffic_note, extra_roof = None, None, {}
    try:   # HBM bytes per launch from the PMC passes (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 runs) of the same workload
        pmc = json.load(open(PMC_FILE))
        if (W, H, B, a.disp) == (1280, 720, 32, 128):
            if pmc.get("kernels_hip_sha256") == kernels_sha():
                traffic = pmc["k_dense"]["traffic_bytes"]
                extra_roof = {"valu_issue_frac_alone": pmc["k_dense"].get("valu_issue_frac_alone"),
                              "SQ_INSTS_VALU": pmc["k_dense"].get("SQ_INSTS_VALU"), "pmc_commit": pmc.get("commit")}
                wp = pmc.get("whole_path")
                if wp:                                       # every kernel of a batch: counted HBM bytes against SURVEY 8d's 97 B per pixel and pair
                    extra_roof.update({"whole_path_traffic": wp["traffic_bytes_per_batch"], "whole_path_traffic_ratio": wp["traffic_ratio"],
                                       "whole_path_traffic_is": "PMC FETCH_SIZE x 2 + WRITE_SIZE summed over the %d kernels of one batch (profiles/%s)" % (len(wp["kernels"]), os.path.basename(PMC_FILE))})
            else:
                traffic_note = "PMC passes in %s were taken with a different kernels.hip (sha256 differs): not reported" % os.path.basename(PMC_FILE)
    except Exception:
        traffic_note = "no PMC file for this round yet"
    roofline = {"bound": "hbm", "kernel": "k_dense_row", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "ms_per_launch": round(k_ms, 4), "ms_per_launch_is": "alone (one batch in flight)" if k_ms_alone else "pipelined",
                "ms_per_launch_pipelined": round(k_ms_pipelined, 4),
                "frac_pipelined": round(alg_bytes / (k_ms_pipelined * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "whole_path_frac": round(value / world * 97.0 * W * H / 1e9 / HBM_PEAK_GBS, 4),
                "ms_owner_pass": round(owner_ms_alone, 4) if owner_ms_alone else None,
                "note": "k_dense_row (the dense matcher; k_owner in front of it resolves which triangle owns a pixel: ms_owner_pass, same stage) is bound by "
                        "vector-instruction issue, not by HBM; pipelined, its launches stretch because kernels of the other slots share the GPU"}
    if traffic_note:
        roofline["traffic_note"] = traffic_note
    roofline.update(extra_roof)

    # BASELINE config 2 beside the headline workload: 640x480, D=64, batch 1, latency mode (one synchronous
    # call per pair on device pointers, nothing pipelined).  Informational; not part of `value`.
    extra = None
    if rank == 0 and world == 1 and not a.no_latency_config:
        w2, h2, d2, reps = 640, 480, 64, 50
        l2, r2 = node.synth_pair(w2, h2, d2, 12345)
        tl, tr = torch.from_numpy(l2).to(dev), torch.from_numpy(r2).to(dev)
        o1 = torch.zeros((h2, w2), dtype=torch.float32, device=dev); o2 = torch.zeros_like(o1)
        e2 = jn.Elas(jn.Elas.parameters(jn.Elas.ROBOTICS, disp_max=d2 - 1), w2, h2, max_batch=1, device=local_rank, host_threads=8, slots=1)
        for _ in range(30):                                  # threads, first-touch pages and clocks settle
            e2.process_batch(1, tl.data_ptr(), tr.data_ptr(), w2, h2 * w2, o1.data_ptr(), o2.data_ptr())
        torch.cuda.synchronize()
        calls = []
        for _ in range(4 * reps):
            t1 = time.perf_counter()
            e2.process_batch(1, tl.data_ptr(), tr.data_ptr(), w2, h2 * w2, o1.data_ptr(), o2.data_ptr())
            calls.append(time.perf_counter() - t1)
        calls.sort()
        lat = calls[len(calls) // 2]                          # median of 200 synchronous calls (p90 beside it)
        lat_p90 = calls[(9 * len(calls)) // 10]
        lone_times = e2.last_times()
        e2.close()
        extra = {"workload": "640x480 D=64 batch=1 latency mode (BASELINE config 2), ELAS", "ms_per_frame": round(lat * 1e3, 3),
                 "ms_per_frame_p90": round(lat_p90 * 1e3, 3), "ms_per_frame_is": "median of 200 synchronous calls",
                 "pairs_per_sec": round(1.0 / lat, 1), "host_stage_ms": round(lone_times["host_stage"], 3), "call_ms_last": round(lone_times["total"], 3)}
        # the same shape through the matcher config 2 names (block matching, include/jn_bm.h — no reference counterpart):
        # one synchronous call per pair, then the u8 map and the scan
        try:
            bm = jn.Bm(jn.Bm.parameters(num_disparities=d2), w2, h2, max_batch=1, device=local_rank)
            d16 = torch.zeros((h2, w2), dtype=torch.int16, device=dev); u8b = torch.zeros((h2, w2), dtype=torch.uint8, device=dev)
            sp2 = node.scan_params(w2, h2); lut2 = node.build_valid_disp_lut(sp2, w2, h2, device=local_rank)
            bins2 = torch.zeros((1, 90), dtype=torch.float64, device=dev); meta2 = torch.zeros((1, 4), dtype=torch.float64, device=dev)

            def bm_call():                                   # jn_bm_process_scan: matcher + u8 map + scan on one stream, one synchronisation
                bm.process_scan(1, tl.data_ptr(), tr.data_ptr(), w2, h2 * w2, d16.data_ptr(), sp2, lut2.ptr, u8b.data_ptr(), bins2.data_ptr(), meta2.data_ptr())
            for _ in range(10):
                bm_call()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(4 * reps):
                bm_call()
            torch.cuda.synchronize()
            lat_bm = (time.perf_counter() - t1) / (4 * reps)
            extra["block_matching"] = {"workload": "640x480 D=64 9x9 block matching batch=1 -> u8 map -> 90-bin scan, one synchronous call (jn_bm_process_scan) per pair",
                                       "ms_per_frame": round(lat_bm * 1e3, 3), "pairs_per_sec": round(1.0 / lat_bm, 1),
                                       "gpu_ms_matcher": round(bm.last_times()["total"], 4)}
            bm.close()
        except Exception as exc:                     # informational leg: never let it take the headline line down
            extra["block_matching"] = {"error": repr(exc)}

    # north_star asks for the rate at 640x480 too: the same pipelined path (ELAS -> u8 map -> scan, inputs resident, four batches of 32 in
    # flight) on 640x480 D=64 Appendix-A pairs, informational, outside `value`; EVERY frame of every slot against the compiled reference (round 6)
    vga = None
    if rank == 0 and world == 1 and not a.no_latency_config and (W, H) != (640, 480):
        try:
            w3, h3, d3, b3 = 640, 480, 64, 32
            L3 = np.empty((b3, h3, w3), np.uint8); R3 = np.empty((b3, h3, w3), np.uint8)
            for b in range(b3):
                L3[b], R3[b] = node.synth_pair(w3, h3, d3, 12345 + b)
            # one DISTINCT copy of the batch per slot (rotated by slot * B / S frames, as the headline's), every frame of every slot checked
            rot3 = [(sl * b3) // S for sl in range(S)]
            tL3 = [torch.from_numpy(np.roll(L3, -rot3[sl], axis=0)).to(dev) for sl in range(S)]
            tR3 = [torch.from_numpy(np.roll(R3, -rot3[sl], axis=0)).to(dev) for sl in range(S)]
            e3 = jn.Elas(jn.Elas.parameters(jn.Elas.ROBOTICS, disp_max=d3 - 1), w3, h3, max_batch=b3, device=local_rank, host_threads=host_threads, slots=S)
            sp3 = node.scan_params(w3, h3); lut3 = node.build_valid_disp_lut(sp3, w3, h3, device=local_rank)
            o1 = [torch.zeros((b3, h3, w3), dtype=torch.float32, device=dev) for _ in range(S)]; o2 = [torch.zeros_like(x) for x in o1]
            u3 = [torch.zeros((b3, h3, w3), dtype=torch.uint8, device=dev) for _ in range(S)]
            sc3 = [parallel.ScanBuffer(b3, 90, dev) for _ in range(S)]
            st3 = [(C.c_int32 * b3)() for _ in range(S)]

            def run3(steps):
                fl = []
                for i in range(steps):
                    sl = i % S
                    if len(fl) == S:
                        e3.wait(fl.pop(0))
                    e3.submit_scan(sl, b3, tL3[sl].data_ptr(), tR3[sl].data_ptr(), w3, h3 * w3, o1[sl].data_ptr(), o2[sl].data_ptr(), sp3, lut3.ptr,
                                   u3[sl].data_ptr(), sc3[sl].bins.data_ptr(), sc3[sl].meta.data_ptr(), st3[sl])
                    fl.append(sl)
                while fl:
                    e3.wait(fl.pop(0))
            run3(12)
            torch.cuda.synchronize()
            regs3 = []
            for _ in range(5):
                t1 = time.perf_counter(); run3(80); torch.cuda.synchronize(); regs3.append(time.perf_counter() - t1)
            el3 = float(np.median(regs3)) / 80
            gold3 = bench_golden(w3, h3, d3, d3 - 1)          # tests/golden/bench_vga_golden.json: the compiled reference on all 32 seeds
            L_ = jn.load()
            n3 = {"D1": 0, "u8_map": 0, "scan": 0}; bad3 = []; worst3 = 0.0
            for sl in range(S):
                d_h, u_h = o1[sl].cpu().numpy(), u3[sl].cpu().numpy()
                b_h, m_h = sc3[sl].bins.cpu().numpy(), sc3[sl].meta.cpu().numpy()
                for i in range(b3):
                    seed = 12345 + (i + rot3[sl]) % b3
                    g3 = gold3.get(seed)
                    if g3 is None:
                        continue
                    got = "%016x" % L_.jn_fnv1a64_u32(d_h[i].ctypes.data, d_h[i].size); n3["D1"] += 1
                    if got != g3["d1_fnv"]:
                        bad3.append({"slot": sl, "frame": i, "seed": seed, "what": "D1", "got": got, "expected": g3["d1_fnv"]})
                    if "u8_fnv" in g3:
                        got = "%016x" % L_.jn_fnv1a64_u32(u_h[i].ctypes.data, u_h[i].size // 4); n3["u8_map"] += 1
                        if got != g3["u8_fnv"]:
                            bad3.append({"slot": sl, "frame": i, "seed": seed, "what": "u8 map", "got": got, "expected": g3["u8_fnv"]})
                    if "bins" in g3:
                        diff = max(float(np.abs(b_h[i] - np.array(g3["bins"])).max()), float(np.abs(m_h[i] - np.array(g3["meta"])).max()))
                        worst3 = max(worst3, diff); n3["scan"] += 1
                        if not diff <= 1e-4:
                            bad3.append({"slot": sl, "frame": i, "seed": seed, "what": "scan", "max_abs_diff": diff})
            vga = {"workload": "640x480 D=64, ELAS -> u8 map -> 90-bin scan, batch=32, %d batches in flight (a distinct copy of the batch per slot), inputs resident in HBM" % S,
                   "pairs_per_sec": round(b3 / el3, 1), "ms_per_step": round(el3 * 1e3, 3), "steps_per_region": 80, "regions": 5,
                   "whole_path_frac": round(b3 / el3 * 97.0 * w3 * h3 / 1e9 / HBM_PEAK_GBS, 4),
                   "check": {"what": "every frame of every slot against the compiled reference (tests/golden/bench_vga_golden.json): FNV-1a-64 of D1 and of the u8 map "
                                     "(bit-exact), 90 bins + 4 extrema within 1e-4",
                             "frames_checked": n3, "scan_max_abs_diff": worst3 if n3["scan"] else None, "mismatches": bad3[:8], "n_mismatches": len(bad3),
                             "ok": (len(bad3) == 0) if n3["D1"] else None}}
            e3.close()
            del o1, o2, u3, tL3, tR3
        except Exception as exc:                             # informational leg: never let it take the headline line down
            vga = {"error": repr(exc)}

    # the non-reference matchers on the SAME batch, after the timed ELAS regions and outside `value`: BASELINE config 3 names
    # "SGM 8-path", the reference has only ELAS; these keys let the driver's line record what include/jn_sgm.h / jn_bm.h run at
    other_modes = None
    if rank == 0 and world == 1 and not a.no_latency_config and a.disp in (64, 128, 256):
        other_modes = {}
        # The ELAS handle is done: it is closed before these legs.  Left open, its four slot streams keep hardware queues (the runtime
        # has GPU_MAX_HW_QUEUES of them, 8 here), the SGM slots' streams then share queues among themselves and batches that should
        # overlap run one after the other: 4.37 k instead of 4.9 k pairs/s (profiles/*_hw_queues_ab.txt).
        elas.close()
        for kind in ("sgm", "bm", "bm_ssd"):
            try:
                disp16 = torch.zeros((B, H, W), dtype=torch.int16, device=dev)
                if kind == "sgm":
                    m = jn.Sgm(jn.Sgm.parameters(num_disparities=a.disp), W, H, max_batch=B, device=local_rank)
                else:
                    m = jn.Bm(jn.Bm.parameters(num_disparities=a.disp, block_radius=4, cost_function=1 if kind == "bm_ssd" else 0), W, H, max_batch=B, device=local_rank)
                reps_m = 120 if kind == "sgm" else 160   # the pipelined legs need enough batches (0.8 / 0.25 s) for their fill and drain not to weigh
                if kind == "sgm":                            # six batches in flight (jn_sgm_submit_scan / jn_sgm_wait), as `--mode sgm --sgm-slots 6` runs it
                    nsl = 6
                    outs_m = [disp16] + [torch.zeros_like(disp16) for _ in range(nsl - 1)]
                    rot_m = [(s_ * max(1, B // nsl)) % B for s_ in range(nsl)]           # a distinct copy of the batch per slot: the same B pairs rotated by rot_m[slot] frames
                    base_l = torch.roll(dLs[0], shifts=rot[0], dims=0); base_r = torch.roll(dRs[0], shifts=rot[0], dims=0)
                    in_l = [torch.roll(base_l, shifts=-r_, dims=0) for r_ in rot_m]; in_r = [torch.roll(base_r, shifts=-r_, dims=0) for r_ in rot_m]
                    seed_m = lambda s_, i: 12345 + (i + rot_m[s_]) % B + 1000 * rank

                    def run_m(k):
                        for i in range(k):
                            if i >= nsl:
                                m.wait(i % nsl)
                            m.submit_scan(i % nsl, B, in_l[i % nsl].data_ptr(), in_r[i % nsl].data_ptr(), W, H * W, outs_m[i % nsl].data_ptr())
                        for sl in range(nsl):
                            m.wait(sl)
                else:                                        # the block matcher the same way (jn_bm_submit_scan / jn_bm_wait)
                    nsl = 4
                    outs_m = [disp16] + [torch.zeros_like(disp16) for _ in range(nsl - 1)]
                    seed_m = lambda s_, i: seed_of(s_ % S, i)

                    def run_m(k):
                        for i in range(k):
                            if i >= nsl:
                                m.wait(i % nsl)
                            m.submit_scan(i % nsl, B, dLs[(i % nsl) % len(dLs)].data_ptr(), dRs[(i % nsl) % len(dRs)].data_ptr(), W, H * W, outs_m[i % nsl].data_ptr())
                        for sl in range(nsl):
                            m.wait(sl)
                run_m(16 if kind == "sgm" else 8)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                run_m(reps_m)
                torch.cuda.synchronize()
                el_m = (time.perf_counter() - t1) / reps_m
                # every frame of every slot against the mode's scalar definition (tests/golden/bench_modes_golden.json: all 32 pairs of the
                # headline batch); other configurations: frame 0 of slot 0 against the recorded hash of seed 12345
                got_m = want_m = None
                n_checked, bad_m = 0, []
                modes_gold = None
                try:
                    mg = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_modes_golden.json")))
                    if (mg["W"], mg["H"], mg["scene_disp"], mg["D"]) == (W, H, scene, a.disp):
                        modes_gold = mg["frames"]
                except (OSError, ValueError, KeyError):
                    pass
                if modes_gold is not None:
                    for s_ in range(nsl):
                        hm = outs_m[s_].cpu().numpy()
                        for i in range(B):
                            g = modes_gold.get(str(seed_m(s_, i)))
                            if g is None:
                                continue
                            gh = "%016x" % jn.load().jn_fnv1a64_u32(hm[i].ctypes.data, hm[i].size // 2)
                            n_checked += 1
                            if gh != g[kind]:
                                bad_m.append({"slot": s_, "frame": i, "seed": seed_m(s_, i), "got": gh, "expected": g[kind]})
                    check_m = {"frames_checked": n_checked, "slots": nsl, "n_mismatches": len(bad_m), "mismatches": bad_m[:4], "ok": (len(bad_m) == 0) if n_checked else None,
                               "source": "tests/golden/bench_modes_golden.json (the mode's scalar definition on every pair of the batch; the reference has no such matcher)"}
                else:
                    host = disp16[0].cpu().numpy()
                    got_m = "%016x" % jn.load().jn_fnv1a64_u32(host.ctypes.data, host.size // 2)
                    for line in open(os.path.join(ROOT, "tests", "golden", "%s_hashes.txt" % kind)):
                        f = line.split()
                        if kind == "sgm" and f[:6] == [str(W), str(H), str(scene), str(a.disp), "0", "12345"]:
                            want_m = f[6]
                        if kind in ("bm", "bm_ssd") and len(f) >= 8 and f[:7] == [str(W), str(H), str(scene), str(a.disp), "4", "0", "12345"]:
                            want_m = f[7]
                    check_m = {"got": got_m, "expected": want_m, "ok": (got_m == want_m) if want_m else None, "frames_checked": 1 if want_m else 0,
                               "source": "tests/golden/%s_hashes.txt (the mode's scalar definition; the reference has no such matcher)" % kind}
                roof_m = None
                if kind == "sgm":                            # SURVEY 8d's B_sgm = one read + one write of the W x H x D byte volume and the images, per batch, against the step time
                    b_sgm = float((4 * W * H * a.disp + 5 * W * H) * B)
                    roof_m = {"bound": "hbm", "achieved": round(b_sgm / el_m / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b_sgm / el_m / 1e9 / HBM_PEAK_GBS, 4),
                              "algorithmic_bytes_per_batch": int(b_sgm), "traffic": None, "is": "B_sgm per batch / step time with six batches in flight"}
                    pm_ = sgm_pmc_traffic(W, H, a.disp, B)
                    if pm_:
                        roof_m["traffic"] = pm_["bytes"]; roof_m["traffic_is"] = pm_["note"]
                other_modes[kind] = {"roofline": roof_m, "workload": "%dx%d D=%d %s batch=%d, disparity maps only (jn_%s_submit_scan / wait), a distinct input batch per slot, same pairs as the ELAS regions" %
                                                 (W, H, a.disp, {"sgm": "SGM 8 paths (six batches in flight)", "bm": "9x9 block matching (SAD, v_qsad; four batches in flight)", "bm_ssd": "9x9 block matching (SSD as an int8 contraction, v_mfma_i32_32x32x32_i8; four batches in flight)"}[kind], B,
                                                  "bm" if kind == "bm_ssd" else kind),
                                     "pairs_per_sec": round(B / el_m, 1), "ms_per_batch": round(el_m * 1e3, 3), "gpu_ms_stages": {k: round(v, 3) for k, v in m.last_times().items()},
                                     "check": check_m}
                m.close()
                del disp16
            except Exception as exc:                 # informational legs: never let them take the headline line down
                other_modes[kind] = {"error": repr(exc)}

    # who took part: gathered over the collective backend, so the line shows what the N ranks really ran on
    ranks_info = None
    if dist is not None:
        mine = {"rank": rank, "device": local_rank, "pin": pin, "rccl_comm": comm_info, "host_threads": host_threads}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks_info = gathered

    if rank == 0:
        out = {
            "metric": "stereo_pairs_per_sec", "value": round(value, 1), "unit": "pairs/s",
            "n_gpus": (dist.get_world_size() if dist is not None else 1),
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "ms_per_frame": round(elapsed / (B * a.steps) * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32",
            "data": "synthetic",
            "config": {"workload": "%dx%d rectified pairs (scene disparities <= %d), ELAS disp_max=%d (D=%d), batch=%d per GPU -> u8 map -> 90-bin scan" %
                                   (W, H, scene, a.disp - 1, a.disp, B),
                       "batch_per_gpu": B, "slots": S, "host_threads": host_threads, "pairs_failed": failed,
                       "parallelism": ("rigs sharded 1 batch/GPU over %d ranks; merge: %s" % (world, merge_kind)) if world > 1 else "single GPU"},
            "timing": {"regions": len(regions), "region_s_median": round(elapsed, 5), "region_s_min": round(min(regions), 5),
                       "region_s_max": round(max(regions), 5), "timed_s_total": round(sum(regions), 4),
                       "value_from": "median region; every region is exactly `steps` steps between barrier+synchronize pairs"},
            "stage_ms_per_batch": {k: round(v, 3) for k, v in stage_ms.items()},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "check": check,
            "latency_config": extra,
            "vga_config": vga,
            "other_modes": other_modes,
            "host_cpu": host_cpu,
        }
        if merge_report is not None:
            out["merge"] = merge_report
        if ranks_info is not None:
            out["ranks"] = ranks_info
            out["distinct_devices"] = sorted({r["device"] for r in ranks_info})
        elif pin is not None:
            out["pin"] = pin
        print(json.dumps(out))
        sys.stdout.flush()
    elas.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and check and check["ok"] is False:
        raise SystemExit("bench.py: D1 of the timed path differs from the reference's golden hash: %s" % check)
    if rank == 0 and vga and vga.get("check", {}).get("ok") is False:
        raise SystemExit("bench.py: the 640x480 leg differs from the reference's recorded answers: %s" % vga["check"])


def enough_gpus(a):
    """--gpus N needs N devices (one rank per GPU; RCCL refuses two ranks on one device).  torch.cuda.device_count() reads the driver's
    list without initialising the GPU, so this is safe before ranks are spawned."""
    if a.share_gpu or a.gpus <= 1:
        return
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus:
        raise SystemExit("bench.py: --gpus %d but this machine shows %d GPU%s (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = %s / %s); nothing was started.  "
                         "For a dry run of the N > 1 path on one GPU: --dist-backend gloo --share-gpu"
                         % (a.gpus, have, "" if have == 1 else "s", os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")))


def start_watchdog(a):
    """A rank that is stuck before its first timed region (rendezvous, communicator creation: a peer that never arrives) must not hang the
    job: after JN_BENCH_STARTUP_TIMEOUT_S (default 300) without the `started` event the process exits non-zero; spawn_ranks / the launcher
    then ends the other ranks.  A thread that only ever calls os._exit — never an exec."""
    import threading
    started = threading.Event()
    limit = float(os.environ.get("JN_BENCH_STARTUP_TIMEOUT_S", "300"))

    def guard():
        if not started.wait(limit):
            sys.stderr.write("bench.py rank %s: not through start-up (process group, communicator, first batch) after %.0f s: exiting 3\n"
                             % (os.environ.get("RANK", "0"), limit))
            sys.stderr.flush()
            os._exit(3)
    if a.gpus > 1 and limit > 0:
        threading.Thread(target=guard, daemon=True).start()
    return started


def main():
    a = parse_args()
    enough_gpus(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    run_rank(a)


if __name__ == "__main__":
    main()
