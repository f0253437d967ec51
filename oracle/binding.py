"""ctypes bindings for the two CHECKERS under oracle/ — test infrastructure only.

* ``Oracle``  -> oracle/_build/liboracle.so, the from-scratch scalar restatement (travels everywhere).
* ``Reference`` -> oracle/_ref/libelas_ref.so, the real reference library compiled from
  /root/reference by oracle/Makefile (present when it was built in the dev container; it is a
  prebuilt artefact on the GPU box).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libelas_ref.so")

_PARAM_FIELDS = [
    ("disp_min", C.c_int32), ("disp_max", C.c_int32), ("support_threshold", C.c_float),
    ("support_texture", C.c_int32), ("candidate_stepsize", C.c_int32), ("incon_window_size", C.c_int32),
    ("incon_threshold", C.c_int32), ("incon_min_support", C.c_int32), ("add_corners", C.c_int32),
    ("grid_size", C.c_int32), ("beta", C.c_float), ("gamma", C.c_float), ("sigma", C.c_float),
    ("sradius", C.c_float), ("match_texture", C.c_int32), ("lr_threshold", C.c_int32),
    ("speckle_sim_threshold", C.c_float), ("speckle_size", C.c_int32), ("ipol_gap_width", C.c_int32),
    ("filter_median", C.c_int32), ("filter_adaptive_mean", C.c_int32), ("postprocess_only_left", C.c_int32),
    ("subsampling", C.c_int32),
]


class Params(C.Structure):
    """Mirror of Elas::parameters (reference elas.h:60-82)."""
    _fields_ = _PARAM_FIELDS

    def copy(self):
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        return q


class ScanParams(C.Structure):
    _fields_ = [("Q", C.c_double * 16), ("XR", C.c_double * 9), ("XT", C.c_double * 3),
                ("crop_offset_x", C.c_int32), ("crop_offset_y", C.c_int32),
                ("gp_height_thresh", C.c_double), ("gp_angle_thresh", C.c_double), ("gp_dist_thresh", C.c_double),
                ("fov_deg", C.c_double), ("bins", C.c_int32), ("pi_approx", C.c_double)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build(force=False):
    """Compile liboracle.so (and libelas_ref.so when /root/reference exists)."""
    args = ["make", "-C", HERE, "all"] + (["-B"] if force else [])
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        self.lib = L = C.CDLL(ORACLE_SO)
        L.orc_fnv1a64_u32.restype = C.c_uint64
        for f in ("orc_obstacle_scan", "orc_point_cloud", "orc_obstacle_scan_points"):
            getattr(L, f).restype = C.c_int64

    # ---- params / inputs ----
    def params(self, setting=0, **kw):
        p = Params()
        self.lib.orc_params_default(C.byref(p), setting)
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def scan_params(self, W, H):
        sp = ScanParams()
        self.lib.orc_scan_params_default(C.byref(sp), W, H)
        return sp

    def synth_pair(self, W, H, sceneD, seed=12345):
        L_ = np.zeros((H, W), np.uint8)
        R_ = np.zeros((H, W), np.uint8)
        self.lib.orc_synth_pair(W, H, sceneD, C.c_uint32(seed), _p(L_), _p(R_))
        return L_, R_

    def set_uninit_fill(self, byte):
        """Byte the descriptor border holds (default 0 = the definition); see Reference for what reads it."""
        self.lib.orc_set_uninit_fill(int(byte))

    def fnv(self, a):
        a = np.ascontiguousarray(a)
        assert a.dtype.itemsize == 4
        return int(self.lib.orc_fnv1a64_u32(_p(a), C.c_int64(a.size)))

    # ---- ELAS stages ----
    def sobel(self, I):
        H, bpl = I.shape
        du = np.zeros_like(I)
        dv = np.zeros_like(I)
        self.lib.orc_sobel3x3(_p(np.ascontiguousarray(I)), bpl, H, _p(du), _p(dv))
        return du, dv

    def descriptor(self, I):
        H, W = I.shape
        d = np.zeros((H, W, 16), np.uint8)
        self.lib.orc_descriptor(_p(np.ascontiguousarray(I)), W, H, W, _p(d))
        return d

    def candidates(self, p, d1, d2):
        H, W = d1.shape[:2]
        cw, ch = C.c_int32(), C.c_int32()
        self.lib.orc_candidates(C.byref(p), _p(d1), _p(d2), W, H, None, C.byref(cw), C.byref(ch))
        D = np.zeros((ch.value, cw.value), np.int16)
        self.lib.orc_candidates(C.byref(p), _p(d1), _p(d2), W, H, _p(D), C.byref(cw), C.byref(ch))
        return D

    def remove_inconsistent(self, p, D):
        D = np.ascontiguousarray(D.copy())
        self.lib.orc_remove_inconsistent(C.byref(p), _p(D), D.shape[1], D.shape[0])
        return D

    def remove_redundant(self, D, max_dist, thresh, vertical):
        D = np.ascontiguousarray(D.copy())
        self.lib.orc_remove_redundant(_p(D), D.shape[1], D.shape[0], max_dist, thresh, int(vertical))
        return D

    def support(self, p, d1, d2):
        H, W = d1.shape[:2]
        cap = (W // max(p.candidate_stepsize, 1) + 2) * (H // max(p.candidate_stepsize, 1) + 2) + 8
        out = np.zeros((cap, 3), np.int32)
        n = self.lib.orc_support(C.byref(p), _p(d1), _p(d2), W, H, _p(out), cap)
        return out[:n].copy()

    def triangulate(self, xy):
        xy = np.ascontiguousarray(xy, np.float32)
        n = xy.shape[0]
        cap = 2 * n + 16
        out = np.zeros((cap, 3), np.int32)
        nt = self.lib.orc_triangulate(_p(xy), n, _p(out), cap)
        if nt < 0:
            raise ValueError("orc_triangulate: unsupported input")
        return out[:nt].copy()

    def triangles(self, sup, right):
        sup = np.ascontiguousarray(sup, np.int32)
        n = sup.shape[0]
        cap = 2 * n + 16
        c = np.zeros((cap, 3), np.int32)
        pl = np.zeros((cap, 6), np.float32)
        nt = self.lib.orc_triangles(_p(sup), n, int(right), _p(c), _p(pl), cap)
        if nt < 0:
            raise ValueError("orc_triangles: unsupported input")
        return c[:nt].copy(), pl[:nt].copy()

    def grid(self, p, sup, W, H, right):
        sup = np.ascontiguousarray(sup, np.int32)
        gw = -(-W // p.grid_size)
        gh = -(-H // p.grid_size)
        g = np.zeros((gh, gw, p.disp_max + 2), np.int32)
        dims = (C.c_int32 * 3)()
        self.lib.orc_grid(C.byref(p), _p(sup), sup.shape[0], W, H, int(right), _p(g), dims)
        assert (dims[1], dims[2]) == (gw, gh)
        return g

    def dense(self, p, sup, corners, planes, grid, d1, d2, right):
        H, W = d1.shape[:2]
        sup = np.ascontiguousarray(sup, np.int32)
        corners = np.ascontiguousarray(corners, np.int32)
        planes = np.ascontiguousarray(planes, np.float32)
        gd = (C.c_int32 * 3)(grid.shape[2], grid.shape[1], grid.shape[0])
        D = np.zeros((H, W), np.float32)
        self.lib.orc_dense(C.byref(p), _p(sup), sup.shape[0], _p(corners), _p(planes), corners.shape[0],
                           _p(grid), gd, _p(d1), _p(d2), W, H, int(right), _p(D))
        return D

    def lr_check(self, p, D1, D2):
        D1 = np.ascontiguousarray(D1.copy()); D2 = np.ascontiguousarray(D2.copy())
        self.lib.orc_lr_check(C.byref(p), _p(D1), _p(D2), D1.shape[1], D1.shape[0])
        return D1, D2

    def _inplace(self, fn, D, p=None):
        D = np.ascontiguousarray(D.copy())
        if p is None:
            fn(_p(D), D.shape[1], D.shape[0])
        else:
            fn(C.byref(p), _p(D), D.shape[1], D.shape[0])
        return D

    def speckle(self, p, D): return self._inplace(self.lib.orc_speckle, D, p)
    def gap(self, p, D): return self._inplace(self.lib.orc_gap, D, p)
    def adaptive_mean(self, D): return self._inplace(self.lib.orc_adaptive_mean, D)
    def median(self, D): return self._inplace(self.lib.orc_median, D)

    def process(self, p, I1, I2, fill=0.0):
        H, W = I1.shape
        D1 = np.full((H, W), fill, np.float32); D2 = np.full((H, W), fill, np.float32)
        I1 = np.ascontiguousarray(I1); I2 = np.ascontiguousarray(I2)
        st = self.lib.orc_elas_process(C.byref(p), _p(I1), _p(I2), _p(D1), _p(D2), W, H, W)
        if p.subsampling:                        # elas.h:160-162: the maps are (W/2) x (H/2), at the start of the buffers
            n = (W // 2) * (H // 2)
            return st, D1.reshape(-1)[:n].reshape(H // 2, W // 2).copy(), D2.reshape(-1)[:n].reshape(H // 2, W // 2).copy()
        return st, D1, D2

    # ---- node side ----
    def to_u8(self, D):
        D = np.ascontiguousarray(D, np.float32)
        out = np.zeros(D.shape, np.uint8)
        self.lib.orc_disparity_to_u8(_p(D), _p(out), C.c_int64(D.size))
        return out

    def valid_lut(self, sp, W, H):
        lut = np.zeros((H, W, 2), np.uint8)
        self.lib.orc_build_valid_disp_lut(C.byref(sp), W, H, _p(lut))
        return lut

    def scan(self, sp, disp, lut):
        H, W = disp.shape
        bins = np.zeros(sp.bins, np.float64); meta = np.zeros(4, np.float64)
        used = self.lib.orc_obstacle_scan(C.byref(sp), _p(np.ascontiguousarray(disp)), _p(lut), W, H, _p(bins), _p(meta))
        return bins, meta, used

    def compact(self, bins):
        out = np.zeros(len(bins), np.float32)
        n = self.lib.orc_compact_ranges(_p(np.ascontiguousarray(bins, np.float64)), len(bins), _p(out))
        return out[:n].copy()

    def point_cloud(self, sp, disp):
        H, W = disp.shape
        xyz = np.zeros((H * W, 3), np.float32)
        n = self.lib.orc_point_cloud(C.byref(sp), _p(np.ascontiguousarray(disp)), W, H, _p(xyz))
        return xyz[:n].copy()

    def scan_points(self, sp, xyz):
        xyz = np.ascontiguousarray(xyz, np.float64)
        bins = np.zeros(sp.bins, np.float64); meta = np.zeros(4, np.float64)
        used = self.lib.orc_obstacle_scan_points(C.byref(sp), _p(xyz), C.c_int64(xyz.shape[0]), _p(bins), _p(meta))
        return bins, meta, used


    def scan_to_points(self, ranges, angle_min, angle_max):
        ranges = np.ascontiguousarray(ranges, np.float32)
        xy = np.zeros((len(ranges), 2), np.float64)
        self.lib.orc_scan_to_points(_p(ranges), len(ranges), C.c_float(angle_min), C.c_float(angle_max), _p(xy))
        return xy

    def check_obstacle(self, xy, history, clear_front=0.24 + 0.8, clear_side=0.3, laser_pt_thresh=8):
        """history: python list standing in for navigate.cpp's deque `commands`; updated in place."""
        xy = np.ascontiguousarray(xy, np.float64)
        h = np.zeros(20, np.int32); h[:len(history)] = history
        n = C.c_int32(len(history)); count = C.c_int32(0); closest = C.c_double(0); conf = C.c_double(0)
        self.lib.orc_check_obstacle.restype = C.c_int32
        obst = self.lib.orc_check_obstacle(_p(xy), len(xy), _p(h), C.byref(n), C.c_double(clear_front), C.c_double(clear_side),
                                           laser_pt_thresh, C.byref(count), C.byref(closest), C.byref(conf))
        history[:] = [int(v) for v in h[:n.value]]
        return obst, count.value, closest.value, conf.value

    def choose_direction(self, xy, last_dir, clear_front=0.24 + 0.8):
        xy = np.ascontiguousarray(xy, np.float64)
        self.lib.orc_choose_direction.restype = C.c_int32
        return self.lib.orc_choose_direction(_p(xy), len(xy), C.c_double(clear_front), last_dir)

    def scan_cloud(self, sp, disp):
        H, W = disp.shape
        bins = np.zeros(sp.bins, np.float64); meta = np.zeros(4, np.float64)
        self.lib.orc_obstacle_scan_cloud.restype = C.c_int64
        used = self.lib.orc_obstacle_scan_cloud(C.byref(sp), _p(np.ascontiguousarray(disp)), W, H, _p(bins), _p(meta))
        return bins, meta, used

    def undistort_map(self, K, D, R, P, W, H):
        arr = [np.ascontiguousarray(np.asarray(a, np.float64)) for a in (K, D, R, P)]
        mx = np.zeros((H, W), np.float32); my = np.zeros((H, W), np.float32)
        self.lib.orc_init_undistort_rectify_map(_p(arr[0]), _p(arr[1]), _p(arr[2]), _p(arr[3]), W, H, _p(mx), _p(my))
        return mx, my

    def remap(self, src, mx, my):
        src = np.ascontiguousarray(src)
        H, W = mx.shape
        dst = np.zeros((H, W), np.uint8)
        self.lib.orc_remap_bilinear(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(np.ascontiguousarray(mx)),
                                    _p(np.ascontiguousarray(my)), _p(dst), W, H, W)
        return dst


class SgmParams(C.Structure):
    _fields_ = [("num_disparities", C.c_int32), ("P1", C.c_int32), ("P2", C.c_int32), ("prefilter_cap", C.c_int32),
                ("lr_max_diff", C.c_int32), ("subpixel", C.c_int32)]


class SgmOracle:
    """oracle/sgm_oracle.cpp — the scalar definition of the SGM mode (self-referential: the reference has no SGM)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        self.lib = C.CDLL(ORACLE_SO)
        self.lib.orc_sgm_process.restype = C.c_int32

    @staticmethod
    def params(num_disparities=128, P1=10, P2=60, prefilter_cap=31, lr_max_diff=1, subpixel=0):
        return SgmParams(num_disparities, P1, P2, prefilter_cap, lr_max_diff, subpixel)

    def prefilter(self, I, cap=31):
        I = np.ascontiguousarray(I, np.uint8)
        g = np.zeros_like(I)
        self.lib.orc_sgm_prefilter(_p(I), I.shape[1], I.shape[0], cap, _p(g))
        return g

    def path(self, gL, gR, D, P1, P2, dx, dy):
        H, W = gL.shape
        out = np.zeros((H, W, D), np.uint8)
        self.lib.orc_sgm_path(_p(np.ascontiguousarray(gL)), _p(np.ascontiguousarray(gR)), W, H, D, P1, P2, dx, dy, _p(out))
        return out

    def process(self, p, L, R):
        L = np.ascontiguousarray(L, np.uint8); R = np.ascontiguousarray(R, np.uint8)
        H, W = L.shape
        disp = np.zeros((H, W), np.int16)
        rc = self.lib.orc_sgm_process(C.byref(p), _p(L), _p(R), W, H, _p(disp))
        if rc != 0:
            raise ValueError("orc_sgm_process: parameters outside the definition")
        return disp

    def to_u8(self, disp, subpixel):
        disp = np.ascontiguousarray(disp, np.int16)
        out = np.zeros(disp.shape, np.uint8)
        self.lib.orc_sgm_to_u8(_p(disp), int(subpixel), _p(out), C.c_int64(disp.size))
        return out


class BmParams(C.Structure):
    _fields_ = [("num_disparities", C.c_int32), ("block_radius", C.c_int32), ("prefilter_cap", C.c_int32),
                ("lr_max_diff", C.c_int32), ("subpixel", C.c_int32), ("cost_function", C.c_int32)]


class BmOracle:
    """oracle/bm_oracle.cpp — the scalar definition of the block-matching mode (self-referential: the reference has none)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        self.lib = C.CDLL(ORACLE_SO)
        self.lib.orc_bm_process.restype = C.c_int32
        self.lib.orc_bm_cost.restype = C.c_int32

    @staticmethod
    def params(num_disparities=64, block_radius=4, prefilter_cap=31, lr_max_diff=1, subpixel=0, cost_function=0):
        return BmParams(num_disparities, block_radius, prefilter_cap, lr_max_diff, subpixel, cost_function)

    def cost(self, gL, gR, r, side, x, y, d, squared=False):
        H, W = gL.shape
        f = self.lib.orc_bm_cost_ssd if squared else self.lib.orc_bm_cost
        f.restype = C.c_int32
        return int(f(_p(np.ascontiguousarray(gL)), _p(np.ascontiguousarray(gR)), W, H, r, side, x, y, d))

    def process(self, p, L, R):
        L = np.ascontiguousarray(L, np.uint8); R = np.ascontiguousarray(R, np.uint8)
        H, W = L.shape
        disp = np.zeros((H, W), np.int16)
        rc = self.lib.orc_bm_process(C.byref(p), _p(L), _p(R), W, H, _p(disp))
        if rc != 0:
            raise ValueError("orc_bm_process: parameters outside the definition")
        return disp


class LocalReference:
    """The compiled reference (libelas from /root/reference), per stage, loaded INTO THIS PROCESS.

    libelas reads descriptor bytes it never wrote (see ``Reference`` below), so results obtained through this class depend
    on what the calling process's heap happens to hold.  Use it only where that does not matter (bench.py's cpu_baseline
    timing) or inside the isolated worker; parity checks and golden generators go through ``Reference``."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " (build with `make -C oracle ref` where /root/reference exists)")
        self.lib = L = C.CDLL(REF_SO)
        L.ref_open.restype = C.c_void_p
        L.ref_descriptor.restype = C.c_void_p
        L.ref_grid.restype = C.c_void_p

    def params(self, setting=0, **kw):
        p = Params()
        self.lib.ref_params_default(C.byref(p), setting)
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def process(self, p, I1, I2, fill=0.0):
        H, W = I1.shape
        D1 = np.full((H, W), fill, np.float32); D2 = np.full((H, W), fill, np.float32)
        I1 = np.ascontiguousarray(I1); I2 = np.ascontiguousarray(I2)
        self.lib.ref_elas_process(C.byref(p), _p(I1), _p(I2), _p(D1), _p(D2), W, H, W)
        if p.subsampling:
            n = (W // 2) * (H // 2)
            return D1.reshape(-1)[:n].reshape(H // 2, W // 2).copy(), D2.reshape(-1)[:n].reshape(H // 2, W // 2).copy()
        return D1, D2

    def sobel(self, I):
        H, bpl = I.shape
        du = np.zeros_like(I); dv = np.zeros_like(I)
        self.lib.ref_sobel(_p(np.ascontiguousarray(I)), bpl, H, _p(du), _p(dv))
        return du, dv

    def triangulate(self, xy):
        xy = np.ascontiguousarray(xy, np.float32)
        n = xy.shape[0]
        cap = 2 * n + 16
        out = np.zeros((cap, 3), np.int32)
        nt = self.lib.ref_triangulate(_p(xy), n, _p(out), cap)
        return out[:nt].copy()

    def open(self, p, I1, I2):
        return RefSession(self, p, I1, I2)


class RefSession:
    def __init__(self, ref, p, I1, I2):
        self.lib = ref.lib
        self.p = p
        self.H, self.W = I1.shape
        self._I1 = np.ascontiguousarray(I1); self._I2 = np.ascontiguousarray(I2)
        self.h = C.c_void_p(self.lib.ref_open(C.byref(p), _p(self._I1), _p(self._I2), self.W, self.H, self.W))

    def close(self):
        if self.h:
            self.lib.ref_close(self.h)
            self.h = None

    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    def descriptor(self, right):
        ptr = self.lib.ref_descriptor(self.h, int(right))
        buf = (C.c_uint8 * (self.H * self.W * 16)).from_address(ptr)
        return np.frombuffer(buf, np.uint8).reshape(self.H, self.W, 16).copy()

    def match_candidate(self, u, v, right):
        return self.lib.ref_match_candidate(self.h, int(u), int(v), int(right))

    def remove_inconsistent(self, D):
        D = np.ascontiguousarray(D.copy())
        self.lib.ref_remove_inconsistent(self.h, _p(D), D.shape[1], D.shape[0])
        return D

    def remove_redundant(self, D, max_dist, thresh, vertical):
        D = np.ascontiguousarray(D.copy())
        self.lib.ref_remove_redundant(self.h, _p(D), D.shape[1], D.shape[0], max_dist, thresh, int(vertical))
        return D

    def support(self):
        cap = (self.W // 2 + 2) * (self.H // 2 + 2)
        out = np.zeros((cap, 3), np.int32)
        n = self.lib.ref_support(self.h, _p(out), cap)
        return out[:n].copy()

    def set_support(self, sup):
        sup = np.ascontiguousarray(sup, np.int32)
        self.lib.ref_set_support(self.h, _p(sup), sup.shape[0])

    def triangles(self, right, nsup):
        cap = 2 * nsup + 16
        c = np.zeros((cap, 3), np.int32); pl = np.zeros((cap, 6), np.float32)
        nt = self.lib.ref_triangles(self.h, int(right), _p(c), _p(pl), cap)
        return c[:nt].copy(), pl[:nt].copy()

    def grid(self, right):
        dims = (C.c_int32 * 3)()
        ptr = self.lib.ref_grid(self.h, int(right), dims)
        n = dims[0] * dims[1] * dims[2]
        buf = (C.c_int32 * n).from_address(ptr)
        return np.frombuffer(buf, np.int32).reshape(dims[2], dims[1], dims[0]).copy()

    def dense(self, right):
        D = np.zeros((self.H, self.W), np.float32)
        self.lib.ref_dense(self.h, int(right), _p(D))
        return D

    def lr_check(self, D1, D2):
        D1 = np.ascontiguousarray(D1.copy()); D2 = np.ascontiguousarray(D2.copy())
        self.lib.ref_lr_check(self.h, _p(D1), _p(D2))
        return D1, D2

    def _ip(self, fn, D):
        D = np.ascontiguousarray(D.copy())
        fn(self.h, _p(D))
        return D

    def speckle(self, D): return self._ip(self.lib.ref_speckle, D)
    def gap(self, D): return self._ip(self.lib.ref_gap, D)
    def adaptive_mean(self, D): return self._ip(self.lib.ref_adaptive_mean, D)
    def median(self, D): return self._ip(self.lib.ref_median, D)


class _Remote:
    """Handle of an object living in the reference worker; method calls are forwarded."""

    def __init__(self, owner, oid):
        self._owner, self._oid = owner, oid

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return lambda *a, **kw: self._owner._call(self._oid, name, a, kw)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._owner._call(self._oid, "close", (), {})
        self._owner._call(None, "_drop", (self._oid,), {})


class Reference:
    """The compiled reference run in a WORKER PROCESS whose heap hands out memory filled with one known byte.

    Why: libelas allocates its 16-byte-per-pixel descriptor image with _mm_malloc and never initialises it
    (descriptor.cpp:29); createDescriptor writes only u in [3,W-4], v in [3,H-4] (descriptor.cpp:84-88), yet
      * the right-image support match reads the tap at u+d+2 = W-3 when u+d = W-5 (elas.cpp:323-326 bound, :340-349 loads),
      * findMatch admits warp columns 2 and W-3 (`u_warp<window_size || u_warp>=width-window_size`, elas.cpp:744-746,
        :752-754, :763-765, :770-772).
    So D1/D2 depend on what malloc returned.  Freshly mapped pages are zero, which is what the product and the oracle
    define those bytes to be (include/jn_stereo.h); glibc's MALLOC_PERTURB_=255 fills every allocation
    with 0x00 (byte = 255 ^ 0xff) and pins the reference to exactly that state, whatever the calling process's heap or
    environment looks like.  ``Reference(fill=b)`` (b != 0xff) pins it to another byte, for the tests that show the
    dependence.  The worker is oracle/ref_worker.py; arguments and results travel pickled over its stdin/stdout."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self, fill=0):
        import pickle
        import sys
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " (build with `make -C oracle ref` where /root/reference exists)")
        if not 0 <= fill < 255:
            raise ValueError("fill must be in [0, 254] (MALLOC_PERTURB_=0 switches the fill off)")
        self.fill = fill
        self._pickle = pickle
        env = dict(os.environ, MALLOC_PERTURB_=str(255 - fill), PYTHONPATH=os.path.dirname(HERE))
        self._proc = subprocess.Popen([sys.executable, os.path.join(HERE, "ref_worker.py")], stdin=subprocess.PIPE,
                                      stdout=subprocess.PIPE, env=env)

    def _call(self, oid, name, args, kw):
        self._pickle.dump((oid, name, args, kw), self._proc.stdin, protocol=4)
        self._proc.stdin.flush()
        kind, val = self._pickle.load(self._proc.stdout)
        if kind == "err":
            raise RuntimeError("reference worker: " + val)
        if kind == "obj":
            return _Remote(self, val)
        return val

    def params(self, setting=0, **kw):
        p = self._call(0, "params_bytes", (setting,), {})
        q = Params.from_buffer_copy(p)
        for k, v in kw.items():
            setattr(q, k, v)
        return q

    def process(self, p, I1, I2, fill=0.0):
        return self._call(0, "process", (bytes(p), I1, I2, fill), {})

    def sobel(self, I):
        return self._call(0, "sobel", (I,), {})

    def triangulate(self, xy):
        return self._call(0, "triangulate", (xy,), {})

    def open(self, p, I1, I2):
        return self._call(0, "open", (bytes(p), I1, I2), {})

    def close(self):
        if self._proc and self._proc.poll() is None:
            try:
                self._proc.stdin.close()
                self._proc.wait(timeout=10)
            except Exception:
                self._proc.kill()
        self._proc = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
