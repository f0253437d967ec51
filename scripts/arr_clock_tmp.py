import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import jackal_navigation_amd as jn
L = jn.load()
rng = np.random.default_rng(5)
def lattice_case(rng, n, cw, ch, dmax):
    cells = rng.choice(cw * ch, size=n, replace=False); cells.sort()
    uc, vc = cells // ch, cells % ch
    d = rng.integers(0, dmax + 1, n)
    return np.stack([uc, vc, d], axis=1).astype(np.int16)
for n, cw, ch in ((3400, 256, 144), (3900, 256, 144), (7000, 384, 216), (11200, 384, 216)):
    t = lattice_case(rng, n, cw, ch, 127)
    left = np.zeros(n, np.uint16); right = np.zeros(n, np.uint16); ok = (C.c_int32 * 2)()
    for rep in range(3):
        assert L.jn_device_arrangement(0, t.ctypes.data, n, 5, left.ctypes.data, right.ctypes.data, ok) == 0
    for side in (0, 1):
        v = ok[side] & 0xFFFFFFFF
        print(n, "side", side, "sorts", ((v >> 16) & 0xFFFF) << 4, "cycles, whole", (v & 0xFFFF) << 4, "cycles (100 MHz counter?)")
