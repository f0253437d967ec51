#!/usr/bin/env python3
"""Random-configuration stress of the SGM sweep kernels against the scalar definition (GPU box):  python3 scripts/sgm_stress.py [seconds]
Sizes from 8x8 to ~700x300 (narrower than the disparity range, single-block and many-block frames), D in {64,128,256}, random
penalties incl. the 16-bit volume mode (3 P2 > 255), caps, L/R tolerances, sub-pixel on/off, batches of 1-3, both implementations."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jackal_navigation_amd as jn                       # noqa: E402
from jackal_navigation_amd.device import DeviceArray     # noqa: E402
from oracle.binding import Oracle, SgmOracle             # noqa: E402
from scenes import make_scene, KINDS                     # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(os.environ.get("SEED", "7")))
o, so = Oracle(), SgmOracle()
t0 = time.time()
n_cfg = n_pairs = 0
fixed = [(8, 8), (9, 8), (16, 8), (17, 9), (8, 64), (130, 50), (47, 33), (48, 16), (49, 200), (257, 19)]
while time.time() - t0 < budget:
    if n_cfg < len(fixed):
        W, H = fixed[n_cfg]
    else:
        W, H = int(rng.integers(8, 700)), int(rng.integers(8, 300))
    D = int(rng.choice([64, 128, 256]))
    cap = int(rng.integers(1, 32))
    P2 = int(rng.integers(1, 255 - 6 * cap + 1))
    P1 = int(rng.integers(0, P2 + 1))
    kw = dict(P1=P1, P2=P2, prefilter_cap=cap, lr_max_diff=int(rng.integers(-1, 4)), subpixel=int(rng.integers(0, 2)))
    n = int(rng.integers(1, 4))
    pairs = []
    for b in range(n):
        kind = rng.integers(0, len(KINDS) + 2) if min(W, H) > 40 else len(KINDS) + int(rng.integers(0, 2))
        if kind < len(KINDS):
            pairs.append(make_scene(KINDS[kind], W, H, min(D - 1, max(4, W // 3)), int(rng.integers(0, 1 << 30))))
        elif kind == len(KINDS):
            pairs.append((rng.integers(0, 256, (H, W)).astype(np.uint8), rng.integers(0, 256, (H, W)).astype(np.uint8)))
        else:
            pairs.append(o.synth_pair(W, H, min(D, max(8, W // 4)), int(rng.integers(0, 1 << 30))))
    Ls = np.stack([p[0] for p in pairs]); Rs = np.stack([p[1] for p in pairs])
    pitch = W + int(rng.integers(0, 3)) * 8                       # rows of the caller's images may be padded
    Lp = np.zeros((n, H, pitch), np.uint8); Rp = np.zeros((n, H, pitch), np.uint8)
    Lp[:, :, :W] = Ls; Rp[:, :, :W] = Rs
    Lp[:, :, W:] = 199; Rp[:, :, W:] = 7                            # padding bytes must never be read as pixels
    dL, dR = DeviceArray.from_numpy(Lp), DeviceArray.from_numpy(Rp)
    dD = DeviceArray((n, H, W), np.int16)
    with jn.Sgm(jn.Sgm.parameters(num_disparities=D, **kw), W, H, max_batch=n + int(rng.integers(0, 3))) as s:   # handle larger than the batch
        if n > 1:                                                    # a smaller batch first: the handle's buffers are re-used
            s.process_batch(n - 1, dL.ptr + H * pitch, dR.ptr + H * pitch, pitch, H * pitch, dD.ptr)
            first = dD.numpy()[: n - 1].copy()
        s.process_batch(n, dL.ptr, dR.ptr, pitch, H * pitch, dD.ptr)
        out = dD.numpy()
        if n > 1 and not np.array_equal(first, out[1:]):
            print("MISMATCH between a batch of %d and the same frames inside a batch of %d (%dx%d D=%d %s)" % (n - 1, n, W, H, D, kw)); sys.exit(1)
    po = so.params(D, **kw)
    for b in range(n):
        exp = so.process(po, Ls[b], Rs[b])
        if not np.array_equal(out[b], exp):
            bad = np.argwhere(out[b] != exp)
            print("MISMATCH %dx%d D=%d n=%d frame %d %s: %d pixels, first %s got %d exp %d" % (W, H, D, n, b, kw, len(bad), bad[0].tolist(), out[b][tuple(bad[0])], exp[tuple(bad[0])]))
            sys.exit(1)
    for a in (dL, dR, dD):
        a.free()
    n_cfg += 1; n_pairs += n
print("sgm stress PASSED: %d configurations, %d pairs, all bit-identical to oracle/sgm_oracle.cpp (%.0f s)" % (n_cfg, n_pairs, time.time() - t0))
