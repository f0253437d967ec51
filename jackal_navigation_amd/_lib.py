"""Loader for libjn_stereo.so (the C-ABI declared in include/jn_stereo.h).

There is no CPU fallback: if the shared library is missing or does not load, importing the compute
API raises.  When PyTorch is installed it is imported FIRST so that this library binds to the HIP
runtime torch already loaded (one HIP runtime per process; see csrc/Makefile).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# JN_STEREO_LIB: another build of the same library (A/B runs of kernel variants); there is still no fallback if it does not load
LIB_PATH = os.environ.get("JN_STEREO_LIB") or os.path.join(HERE, "libjn_stereo.so")
# The hooks build (csrc/hooks.h, `make -C csrc hooks`): the same sources with the test hooks, the profiling switches and the A/B knobs
# compiled in.  Only tests and measurement scripts load it (hooks_library() below, or JN_STEREO_LIB for a child process).
HOOKS_LIB_PATH = os.path.join(HERE, "libjn_stereo_hooks.so")

JN_OK, JN_ERR_FEW_SUPPORT, JN_ERR_UNSUPPORTED, JN_ERR_INVALID, JN_ERR_NO_DEVICE, JN_ERR_INTERNAL, JN_ERR_COMM = range(7)
STATUS_NAMES = ["JN_OK", "JN_ERR_FEW_SUPPORT", "JN_ERR_UNSUPPORTED", "JN_ERR_INVALID", "JN_ERR_NO_DEVICE", "JN_ERR_INTERNAL", "JN_ERR_COMM"]


class JnError(RuntimeError):
    def __init__(self, status, what):
        super().__init__("%s: %s" % (what, STATUS_NAMES[status] if 0 <= status < len(STATUS_NAMES) else status))
        self.status = status


class ElasParams(C.Structure):
    """jn_elas_params == Elas::parameters (reference src/elas/elas.h:60-82)."""
    _fields_ = [
        ("disp_min", C.c_int32), ("disp_max", C.c_int32), ("support_threshold", C.c_float),
        ("support_texture", C.c_int32), ("candidate_stepsize", C.c_int32), ("incon_window_size", C.c_int32),
        ("incon_threshold", C.c_int32), ("incon_min_support", C.c_int32), ("add_corners", C.c_int32),
        ("grid_size", C.c_int32), ("beta", C.c_float), ("gamma", C.c_float), ("sigma", C.c_float),
        ("sradius", C.c_float), ("match_texture", C.c_int32), ("lr_threshold", C.c_int32),
        ("speckle_sim_threshold", C.c_float), ("speckle_size", C.c_int32), ("ipol_gap_width", C.c_int32),
        ("filter_median", C.c_int32), ("filter_adaptive_mean", C.c_int32), ("postprocess_only_left", C.c_int32),
        ("subsampling", C.c_int32),
    ]


class ScanParams(C.Structure):
    """jn_scan_params: file-scope state of point_cloud.cpp read by the scan functions (:28-69, :217-218)."""
    _fields_ = [("Q", C.c_double * 16), ("XR", C.c_double * 9), ("XT", C.c_double * 3),
                ("crop_offset_x", C.c_int32), ("crop_offset_y", C.c_int32),
                ("gp_height_thresh", C.c_double), ("gp_angle_thresh", C.c_double), ("gp_dist_thresh", C.c_double),
                ("fov_deg", C.c_double), ("bins", C.c_int32), ("pi_approx", C.c_double)]


class StereoCalib(C.Structure):
    """jn_stereo_calib: what main() reads from the calibration YAML (point_cloud.cpp:530-536)."""
    _fields_ = [("K1", C.c_double * 9), ("D1", C.c_double * 5), ("K2", C.c_double * 9), ("D2", C.c_double * 5),
                ("R", C.c_double * 9), ("T", C.c_double * 3), ("calib_width", C.c_int32), ("calib_height", C.c_int32)]


class Rectification(C.Structure):
    _fields_ = [("R1", C.c_double * 9), ("R2", C.c_double * 9), ("P1", C.c_double * 12), ("P2", C.c_double * 12), ("Q", C.c_double * 16)]


NAV_MAX_HISTORY = 64


class NavParams(C.Structure):
    """jn_nav_params: the constants checkObstacle / chooseDirection read (navigate.cpp:37-42, :125-146)."""
    _fields_ = [("clear_front", C.c_double), ("clear_side", C.c_double), ("stop_dist", C.c_double),
                ("laser_pt_thresh", C.c_int32), ("history", C.c_int32), ("history_votes", C.c_int32), ("reserved", C.c_int32)]


class NavState(C.Structure):
    _fields_ = [("votes", C.c_int32 * NAV_MAX_HISTORY), ("head", C.c_int32), ("filled", C.c_int32), ("positives", C.c_int32),
                ("last_dir", C.c_int32)]


class NavDecision(C.Structure):
    _fields_ = [("points_inside", C.c_int32), ("points", C.c_int32), ("obstacle", C.c_int32), ("direction", C.c_int32),
                ("closest", C.c_double), ("confidence", C.c_double)]


class StageTimes(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("gpu_descriptor", "gpu_support", "d2h", "host_stage", "h2d", "gpu_matching",
                                         "gpu_lr", "gpu_speckle", "gpu_gap", "gpu_adaptive_mean", "total")]

    def as_dict(self):
        return {n: float(getattr(self, n)) for n, _ in self._fields_}


# every symbol include/jn_stereo.h declares (tests check the library exports all of them)
EXPORTS = [
    "jn_elas_params_default", "jn_elas_create", "jn_elas_destroy", "jn_elas_process", "jn_elas_process_batch",
    "jn_elas_submit", "jn_elas_wait", "jn_elas_last_times", "jn_scan_params_default", "jn_disparity_to_u8",
    "jn_build_valid_disp_lut", "jn_obstacle_scan", "jn_obstacle_scan_cloud", "jn_disparity_scan", "jn_compact_ranges", "jn_point_cloud",
    "jn_synth_pair", "jn_device_count", "jn_device_malloc", "jn_device_free", "jn_memcpy_h2d", "jn_memcpy_d2h",
    "jn_device_synchronize", "jn_elas_kernel_time", "jn_version", "jn_host_triangulate", "jn_host_stage",
    "jn_stereo_calib_default", "jn_stereo_rectify", "jn_init_undistort_rectify_map", "jn_remap_bilinear",
    "jn_nav_params_default", "jn_nav_state_reset", "jn_scan_to_points", "jn_nav_vote", "jn_device_support_filters", "jn_host_arrangement", "jn_device_arrangement", "jn_elas_submit_scan", "jn_elas_submit_host",
    "jn_host_triangulate_parts", "jn_jpeg_info", "jn_jpeg_decode_gray", "jn_host_jpeg_coefficients", "jn_comm_unique_id", "jn_comm_create", "jn_comm_info", "jn_scan_allreduce", "jn_comm_destroy", "jn_fnv1a64_u32", "jn_elas_set_comm", "jn_elas_merge_time", "jn_elas_merge_order", "jn_jpeg_decode_gray_pair", "jn_elas_bin_stats", "jn_device_triangulate", "jn_elas_route_stats",
]

_lib = None
_bound = {}


def load():
    """Return the ctypes handle of libjn_stereo.so; raises if it is not built."""
    global _lib
    if _lib is None:
        _lib = _bind(LIB_PATH)
    return _lib


class hooks_library:
    """Tests and measurement scripts only: inside the `with` block every call of this package goes to the hooks build
    (libjn_stereo_hooks.so: csrc/hooks.h), where the JN_TEST_* hooks, the *_DBG profiling switches and the A/B knobs exist.  Handles made
    inside the block must be closed inside it.  The release library carries none of those switches."""

    def __enter__(self):
        global _lib
        self._prev = _lib
        _lib = _bind(HOOKS_LIB_PATH)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev
        return False


def _bind(path):
    if path in _bound:
        return _bound[path]
    if not os.path.exists(path):
        raise ImportError("%s is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C jackal_navigation_amd/csrc [hooks]`); there is no CPU fallback" % os.path.basename(path))
    try:  # share torch's HIP runtime when torch is present
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C-ABI itself
        pass
    L = C.CDLL(path)
    L.jn_version.restype = C.c_char_p
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.jn_elas_create.argtypes = [C.POINTER(ElasParams), i32, i32, i32, i32, i32, i32, C.POINTER(vp)]
    L.jn_elas_destroy.argtypes = [vp]
    L.jn_elas_process.argtypes = [vp, vp, vp, vp, vp, C.POINTER(i32 * 3)]
    L.jn_elas_process_batch.argtypes = [vp, i32, vp, vp, i32, i64, vp, vp, vp]
    L.jn_elas_submit.argtypes = [vp, i32, i32, vp, vp, i32, i64, vp, vp, vp]
    L.jn_host_arrangement.argtypes = [vp, vp, i32, vp]
    L.jn_host_arrangement.restype = i32
    L.jn_device_arrangement.argtypes = [i32, vp, i32, i32, vp, vp, vp]
    L.jn_device_triangulate.argtypes = [i32, vp, i32, i32, vp, vp, vp, vp]
    L.jn_elas_submit_host.argtypes = [vp, i32, i32, vp, vp, i32, i64, vp, vp, vp]
    L.jn_elas_submit_scan.argtypes = [vp, i32, i32, vp, vp, i32, i64, vp, vp, C.POINTER(ScanParams), vp, vp, vp, vp, vp]
    L.jn_elas_wait.argtypes = [vp, i32]
    L.jn_elas_last_times.argtypes = [vp, i32, C.POINTER(StageTimes)]
    L.jn_elas_bin_stats.argtypes = [vp, i32, C.POINTER(i32 * 3)]
    L.jn_elas_route_stats.argtypes = [vp, i32, C.POINTER(i32 * 3)]
    L.jn_elas_kernel_time.argtypes = [vp, i32, C.c_char_p, C.POINTER(C.c_float), C.POINTER(i32)]
    L.jn_disparity_to_u8.argtypes = [i32, vp, vp, i64]
    L.jn_build_valid_disp_lut.argtypes = [i32, C.POINTER(ScanParams), i32, i32, vp]
    L.jn_obstacle_scan.argtypes = [i32, C.POINTER(ScanParams), i32, vp, vp, i32, i32, vp, vp]
    L.jn_obstacle_scan_cloud.argtypes = [i32, C.POINTER(ScanParams), i32, vp, i32, i32, vp, vp]
    L.jn_disparity_scan.argtypes = [i32, C.POINTER(ScanParams), i32, vp, vp, i32, i32, vp, vp, vp]
    L.jn_compact_ranges.argtypes = [vp, i32, vp]
    L.jn_point_cloud.argtypes = [i32, C.POINTER(ScanParams), vp, i32, i32, vp, C.POINTER(i64)]
    L.jn_synth_pair.argtypes = [i32, i32, i32, C.c_uint32, vp, vp]
    L.jn_device_count.argtypes = [C.POINTER(i32)]
    L.jn_device_malloc.argtypes = [i32, i64, C.POINTER(vp)]
    L.jn_device_free.argtypes = [i32, vp]
    L.jn_memcpy_h2d.argtypes = [i32, vp, vp, i64]
    L.jn_memcpy_d2h.argtypes = [i32, vp, vp, i64]
    L.jn_device_synchronize.argtypes = [i32]
    L.jn_stereo_rectify.argtypes = [C.POINTER(StereoCalib), i32, i32, C.POINTER(Rectification)]
    L.jn_init_undistort_rectify_map.argtypes = [i32, vp, vp, vp, vp, i32, i32, vp, vp]
    L.jn_remap_bilinear.argtypes = [i32, i32, vp, i32, i32, i32, i64, vp, vp, vp, i32, i32, i32, i64]
    L.jn_host_triangulate.argtypes = [vp, vp, i32, vp]
    L.jn_host_triangulate_parts.argtypes = [vp, vp, i32, vp, i32]
    L.jn_host_stage.argtypes = [C.POINTER(ElasParams), i32, i32, vp, vp, i64, vp]
    L.jn_host_stage.restype = i64
    L.jn_device_support_filters.argtypes = [i32, C.POINTER(ElasParams), i32, i32, i32, vp, i32]
    L.jn_nav_params_default.argtypes = [C.POINTER(NavParams)]
    L.jn_nav_params_default.restype = None
    L.jn_nav_state_reset.argtypes = [C.POINTER(NavState)]
    L.jn_nav_state_reset.restype = None
    L.jn_scan_to_points.argtypes = [vp, i32, C.c_float, C.c_float, vp]
    L.jn_nav_vote.argtypes = [C.POINTER(NavParams), C.POINTER(NavState), vp, i32, C.POINTER(NavDecision)]
    L.jn_jpeg_info.argtypes = [vp, i64, C.POINTER(i32), C.POINTER(i32)]
    L.jn_jpeg_decode_gray.argtypes = [i32, vp, i64, vp, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.jn_jpeg_decode_gray_pair.argtypes = [i32, vp, i64, vp, i64, vp, vp, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.jn_host_jpeg_coefficients.argtypes = [vp, i64, vp, i64, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.jn_host_jpeg_coefficients.restype = i64
    L.jn_comm_unique_id.argtypes = [vp]
    L.jn_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.jn_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.jn_scan_allreduce.argtypes = [vp, i32, i32, vp, vp]
    L.jn_comm_destroy.argtypes = [vp]
    L.jn_elas_set_comm.argtypes = [vp, vp]
    L.jn_elas_merge_time.argtypes = [vp, i32, C.POINTER(C.c_float)]
    L.jn_elas_merge_order.argtypes = [vp, C.POINTER(C.c_uint64), i32]
    L.jn_elas_merge_order.restype = i32
    L.jn_comm_destroy.restype = None
    L.jn_fnv1a64_u32.argtypes = [vp, i64]
    L.jn_fnv1a64_u32.restype = C.c_uint64
    _bound[path] = L
    return L


def check(status, what):
    if status != JN_OK:
        raise JnError(status, what)
