"""Wide parity sweep of the HIP path against the CPU oracle (run on the GPU box): many seeds, sizes, scene depths and
parameter sets, batches through the pipelined device-pointer API; every D1 / D2 must be bit-identical.
Usage: python3 scripts/parity_sweep.py [pairs_per_config]   (the oracle is run in a process pool)"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_pair(W, H, sd, dmax, seed):
    """sd: largest scene disparity of the survey's generator, or the name of a scene kind of tests/scenes.py"""
    if isinstance(sd, str):
        from scenes import make_scene
        return make_scene(sd, W, H, dmax, seed)
    from oracle.binding import Oracle
    return Oracle().synth_pair(W, H, sd, seed)


def oracle_job(args):
    W, H, sd, dmax, seed, kw = args
    from oracle.binding import Oracle
    o = Oracle()
    L, R = make_pair(W, H, sd, dmax, seed)
    kw = dict(kw)
    setting = kw.pop("setting", 0)
    kw.pop("host_threads", None)
    st, D1, D2 = o.process(o.params(setting, disp_max=dmax, **kw), L, R)
    return st, D1, D2


def sgm_job(args):
    W, H, sd, D, seed, kw = args
    from oracle.binding import SgmOracle
    o = SgmOracle()
    L, R = make_pair(W, H, sd, D - 1, seed)
    return o.process(o.params(D, **kw), L, R)


def bm_job(args):
    W, H, sd, D, seed, kw = args
    from oracle.binding import BmOracle
    o = BmOracle()
    L, R = make_pair(W, H, sd, D - 1, seed)
    return o.process(o.params(D, **kw), L, R)


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    configs = [
        (1280, 720, 128, 127, {}), (1280, 720, 40, 127, {}), (640, 480, 64, 63, {}), (320, 180, 48, 255, {}),
        (333, 201, 30, 95, {}), (1920, 1080, 200, 255, {}), (800, 600, 90, 127, {"postprocess_only_left": 0}),
        (640, 360, 64, 95, {"filter_median": 1, "speckle_size": 100}), (512, 384, 50, 79, {"incon_min_support": 8, "incon_threshold": 3}),
    ]
    for kind in ("strips", "patches", "slanted", "photometric", "blobs", "shallow"):
        configs += [(320, 240, kind, 79, {"postprocess_only_left": 0}), (640, 480, kind, 127, {"postprocess_only_left": 0}),
                    (1280, 720, kind, 127, {}), (448, 333, kind, 255, {"postprocess_only_left": 0, "ipol_gap_width": 7})]
    # round 2: both presets, corner points, wide gaps, other grid sizes (k_dense2 / the k_dense fallback), plane radius 3,
    # widths that cut the support kernel's lattice rows into segments, lone pairs on a large pool (Delaunay cut in parts)
    configs += [
        (640, 480, 64, 63, {"setting": 1}), (1280, 720, 128, 127, {"setting": 1}), (448, 333, "strips", 127, {"setting": 1}),
        (640, 360, "blobs", 95, {"add_corners": 1, "ipol_gap_width": 12}), (800, 600, 90, 127, {"ipol_gap_width": 200, "postprocess_only_left": 0}),
        (640, 480, 64, 63, {"grid_size": 16}), (640, 480, "slanted", 95, {"grid_size": 24, "sradius": 3.0}), (512, 384, 50, 79, {"grid_size": 6}),
        (1280, 720, 128, 255, {}), (1600, 900, 150, 191, {}), (2048, 512, 100, 127, {"postprocess_only_left": 0}),
        (3840, 2160, 220, 255, {}), (4096, 600, 100, 127, {"postprocess_only_left": 0}),
        (1280, 720, 128, 127, {"host_threads": 32, "lone": 1}), (1920, 1080, 200, 255, {"host_threads": 32, "lone": 1}),
        (640, 480, "patches", 63, {"filter_adaptive_mean": 0}), (640, 480, "photometric", 63, {"ipol_gap_width": 2}),
    ]
    # round 4: the two options the node never sets — disp_min, and subsampling (half-size maps)
    configs += [
        (640, 480, 64, 63, {"disp_min": 8}), (1280, 720, 128, 127, {"disp_min": 30, "postprocess_only_left": 0}),
        (640, 480, 64, 63, {"subsampling": 1}), (1280, 720, "strips", 127, {"subsampling": 1, "postprocess_only_left": 0}),
        (800, 600, 90, 127, {"subsampling": 1, "filter_median": 1}), (1920, 1080, 200, 255, {"subsampling": 1}),
    ]
    sgm_configs = [(640, 480, 64, 64, {}), (1280, 720, 128, 128, {"subpixel": 1}), (320, 240, "strips", 64, {"subpixel": 1}),
                   (448, 333, "blobs", 128, {"P1": 5, "P2": 40, "prefilter_cap": 20}), (500, 200, 200, 256, {"lr_max_diff": 2})]
    bm_configs = [(640, 480, 64, 64, {}), (1280, 720, 128, 128, {"subpixel": 1}), (320, 240, "strips", 64, {"subpixel": 1, "block_radius": 3}),
                  (448, 333, "blobs", 120, {"block_radius": 2, "prefilter_cap": 20}), (500, 200, 200, 256, {"lr_max_diff": 2}),
                  (1000, 37, "patches", 40, {"lr_max_diff": -1, "subpixel": 1})]
    with ProcessPoolExecutor(max_workers=min(48, os.cpu_count() or 8)) as pool:
        futures = []
        for ci, (W, H, sd, dmax, kw) in enumerate(configs):
            n = per if W * H <= 1280 * 720 else (max(4, per // 3) if W * H <= 1920 * 1080 else 2)
            if kw.get("lone"):
                n = 3
            futures.append([pool.submit(oracle_job, (W, H, sd, dmax, 31000 + 100 * ci + b, {k: v for k, v in kw.items() if k != "lone"})) for b in range(n)])
        sgm_futures = [[pool.submit(sgm_job, (W, H, sd, D, 52000 + 100 * ci + b, kw)) for b in range(max(2, per // 6))] for ci, (W, H, sd, D, kw) in enumerate(sgm_configs)]
        bm_futures = [[pool.submit(bm_job, (W, H, sd, D, 63000 + 100 * ci + b, kw)) for b in range(max(2, per // 6))] for ci, (W, H, sd, D, kw) in enumerate(bm_configs)]
        import jackal_navigation_amd as jn          # GPU side in the parent only (after the workers were forked)
        from jackal_navigation_amd.device import DeviceArray
        bad = 0
        for ci, (W, H, sd, dmax, kw) in enumerate(configs):
            n = len(futures[ci])
            pairs = [make_pair(W, H, sd, dmax, 31000 + 100 * ci + b) for b in range(n)]
            Ls = np.stack([p[0] for p in pairs]); Rs = np.stack([p[1] for p in pairs])
            dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
            Ho, Wo = (H // 2, W // 2) if kw.get("subsampling") else (H, W)      # elas.h:160-162: half-size maps with subsampling
            d1 = DeviceArray.from_numpy(np.full((n, Ho, Wo), 7.0, np.float32)); d2 = DeviceArray.from_numpy(np.full((n, Ho, Wo), 7.0, np.float32))
            t0 = time.time()
            pkw = {k: v for k, v in kw.items() if k not in ("setting", "host_threads", "lone")}
            with jn.Elas(jn.Elas.parameters(kw.get("setting", 0), disp_max=dmax, **pkw), W, H, max_batch=n, host_threads=kw.get("host_threads", 8)) as e:
                if kw.get("lone"):                                   # one pair per call: the pool has idle threads, Delaunay runs in parts
                    status = []
                    for b in range(n):
                        status += e.process_batch(1, dL.ptr + b * H * W, dR.ptr + b * H * W, W, H * W, d1.ptr + 4 * b * Ho * Wo, d2.ptr + 4 * b * Ho * Wo)
                else:
                    status = e.process_batch(n, dL.ptr, dR.ptr, W, H * W, d1.ptr, d2.ptr)
            t_gpu = time.time() - t0
            D1, D2 = d1.numpy(), d2.numpy()
            wrong = 0
            for b in range(n):
                st, D1o, D2o = futures[ci][b].result()
                if st != status[b]:
                    wrong += 1
                elif st == 0 and not (np.array_equal(D1[b].view(np.uint32), D1o.view(np.uint32)) and
                                      np.array_equal(D2[b].view(np.uint32), D2o.view(np.uint32))):
                    wrong += 1
            bad += wrong
            print("%4dx%-4d scene %-11s disp_max=%-3d %-45s %3d pairs (%d matched, %d refused alike)  %s  (gpu %.2f s)" %
                  (W, H, sd, dmax, kw, n, sum(1 for x in status if x == 0), sum(1 for x in status if x != 0),
                   "all bit-identical" if wrong == 0 else "%d MISMATCH" % wrong, t_gpu), flush=True)
            for a in (dL, dR, d1, d2):
                a.free()
        for ci, (W, H, sd, D, kw) in enumerate(sgm_configs):
            n = len(sgm_futures[ci])
            pairs = [make_pair(W, H, sd, D - 1, 52000 + 100 * ci + b) for b in range(n)]
            Ls = np.stack([p[0] for p in pairs]); Rs = np.stack([p[1] for p in pairs])
            dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
            dd = DeviceArray((n, H, W), np.int16)
            with jn.Sgm(jn.Sgm.parameters(num_disparities=D, **kw), W, H, max_batch=n) as sg:
                sg.process_batch(n, dL.ptr, dR.ptr, W, H * W, dd.ptr)
            out = dd.numpy()
            wrong = sum(0 if np.array_equal(out[b], sgm_futures[ci][b].result()) else 1 for b in range(n))
            bad += wrong
            print("SGM %4dx%-4d scene %-8s D=%-3d %-45s %3d pairs  %s" % (W, H, sd, D, kw, n, "all bit-identical" if wrong == 0 else "%d MISMATCH" % wrong), flush=True)
            for a in (dL, dR, dd):
                a.free()
        for ci, (W, H, sd, D, kw) in enumerate(bm_configs):
            n = len(bm_futures[ci])
            pairs = [make_pair(W, H, sd, D - 1, 63000 + 100 * ci + b) for b in range(n)]
            Ls = np.stack([p[0] for p in pairs]); Rs = np.stack([p[1] for p in pairs])
            dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
            dd = DeviceArray((n, H, W), np.int16)
            with jn.Bm(jn.Bm.parameters(num_disparities=D, **kw), W, H, max_batch=n) as bm:
                bm.process_batch(n, dL.ptr, dR.ptr, W, H * W, dd.ptr)
            out = dd.numpy()
            wrong = sum(0 if np.array_equal(out[b], bm_futures[ci][b].result()) else 1 for b in range(n))
            bad += wrong
            print("BM  %4dx%-4d scene %-8s D=%-3d %-45s %3d pairs  %s" % (W, H, sd, D, kw, n, "all bit-identical" if wrong == 0 else "%d MISMATCH" % wrong), flush=True)
            for a in (dL, dR, dd):
                a.free()
    print("sweep", "PASSED" if bad == 0 else "FAILED (%d)" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
