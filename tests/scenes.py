"""Synthetic stereo scenes beyond the survey's plane-and-box generator, for parity tests: they stress other parts of
the path (occlusions and disparity jumps -> L/R failures, speckles, gaps; textureless patches -> texture tests and
sparse support; photometric differences -> ambiguous matches; slanted surfaces -> non-constant planes)."""
import numpy as np

KINDS = ("strips", "patches", "slanted", "photometric", "blobs", "shallow", "grain", "periodic")


def _texture(rng, H, W, cell=3):
    t = rng.integers(0, 256, ((H + cell - 1) // cell, (W + 512 + cell - 1) // cell)).astype(np.uint8)
    return np.kron(t, np.ones((cell, cell), np.uint8))[:H, :W + 512]


def _binomial_blur(t, passes):
    """Integer 1-2-1 blur in x and y, `passes` times (exact arithmetic: the same bytes on every machine)."""
    t = t.astype(np.int32)
    for _ in range(passes):
        p = np.pad(t, ((0, 0), (1, 1)), mode="edge"); t = (p[:, :-2] + 2 * p[:, 1:-1] + p[:, 2:] + 2) >> 2
        p = np.pad(t, ((1, 1), (0, 0)), mode="edge"); t = (p[:-2] + 2 * p[1:-1] + p[2:] + 2) >> 2
    return t


def make_scene(kind, W, H, dmax, seed):
    """Returns (L, R) uint8 images; R is the texture, L samples it at x - d(x, y) (as the survey's generator does)."""
    rng = np.random.default_rng(seed)
    tex = _texture(rng, H, W)
    yy, xx = np.mgrid[0:H, 0:W]
    if kind == "grain":
        # natural-image statistics instead of block texture: band-limited noise (white noise through a binomial low-pass, stretched back to
        # the full range), a smooth slanted surface with two depth jumps, independent film grain in both images, the right one slightly defocused
        t = _binomial_blur(rng.integers(0, 256, (H, W + 512)), 3)
        tex = ((t - t.min()) * 255 // max(1, int(t.max() - t.min()))).astype(np.uint8)
    if kind == "periodic":
        # repeated structure: vertical bars of period 12 under a slow vertical gradient and weak noise — every bar matches every other bar
        xw = np.arange(W + 512)[None, :]
        tex = np.clip(60 + 120 * ((xw // 6) % 2) + (np.arange(H)[:, None] * 40) // max(1, H) + rng.integers(-10, 11, (H, W + 512)), 0, 255).astype(np.uint8)
    if kind == "strips":                                   # vertical strips at very different depths
        edges = np.sort(rng.integers(0, W, 7))
        d = np.full((H, W), 3, np.int64)
        for i, e in enumerate(edges):
            d[:, e:] = int(rng.integers(2, max(3, int(dmax * 0.8))))
    elif kind == "patches":                                # plane with textureless rectangles painted over both images
        d = (yy * (0.5 * dmax) / H).astype(np.int64) + 2
    elif kind == "slanted":                                # disparity grows along x and y
        d = (2 + xx * (0.35 * dmax) / W + yy * (0.3 * dmax) / H).astype(np.int64)
    elif kind == "photometric":
        d = (yy * (0.6 * dmax) / H).astype(np.int64) + 2
    elif kind == "blobs":                                  # random ellipses at random depths over a far background
        d = np.full((H, W), 2, np.int64)
        for _ in range(25):
            cx, cy = rng.integers(0, W), rng.integers(0, H)
            ax, ay = rng.integers(10, max(11, W // 6)), rng.integers(8, max(9, H // 6))
            d[((xx - cx) / ax) ** 2 + ((yy - cy) / ay) ** 2 < 1] = int(rng.integers(2, max(3, int(dmax * 0.7))))
    elif kind == "grain":
        d = (2 + xx * (0.3 * dmax) / W + yy * (0.2 * dmax) / H).astype(np.int64)
        d[H // 3:2 * H // 3, W // 4:W // 2] += int(0.2 * dmax)
        d[H // 2:, 3 * W // 5:4 * W // 5] = 3
    elif kind == "periodic":
        d = (3 + yy * (0.25 * dmax) / H).astype(np.int64)
    elif kind == "shallow":                                # disparities 0, 1, 2 only: the far field, where d - 1 and u - d hit their limits
        d = ((xx // 37 + yy // 29) % 3).astype(np.int64)
    else:
        raise ValueError(kind)
    R = tex[:, 256:256 + W].copy()
    L = tex[yy, np.clip(xx - d + 256, 0, tex.shape[1] - 1)].copy()
    if kind == "patches":
        for _ in range(12):
            x0, y0 = int(rng.integers(0, W - 20)), int(rng.integers(0, H - 20))
            w, h = int(rng.integers(15, max(16, W // 5))), int(rng.integers(10, max(11, H // 5)))
            g = int(rng.integers(0, 256))
            L[y0:y0 + h, x0:x0 + w] = g; R[y0:y0 + h, max(0, x0 - 10):x0 + w] = g
    if kind == "grain":
        R = _binomial_blur(R, 1)
        L = np.clip(L.astype(np.int32) + rng.integers(-6, 7, L.shape), 0, 255).astype(np.uint8)
        R = np.clip(R + rng.integers(-6, 7, R.shape), 0, 255).astype(np.uint8)
    if kind == "photometric":
        R = np.clip(R.astype(np.float64) * 0.8 + 25 + rng.normal(0, 6, R.shape), 0, 255).astype(np.uint8)
        L = np.clip(L.astype(np.float64) + rng.normal(0, 4, L.shape), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(L), np.ascontiguousarray(R)
