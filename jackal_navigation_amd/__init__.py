"""jackal_navigation_amd — MI355X (gfx950) implementation of jackal_nav's `point_cloud` hot path.

Rectified stereo pair -> ELAS disparity -> u8 depth map -> Q reprojection -> ground-plane filter ->
90-bin obstacle scan, as hand-written HIP kernels behind a C ABI (include/jn_stereo.h,
libjn_stereo.so).  This package is the thin host-side mirror of the reference interfaces; all
compute lives in csrc/.  Importing the compute API requires the built library; there is no CPU path.
"""
import os as _os

# One hardware queue per slot stream: the HIP runtime multiplexes streams onto 4 hardware queues by default, so with
# 4 slots + the caller's stream two slots share a queue and serialise (measured: -10 % pairs/s).  16 since round 4: a process
# that holds an ELAS handle AND the SGM or block-matching mode's slots needs more than 8 (profiles/r04_hw_queues_ab.txt).  Must be
# in the environment before the HIP runtime initialises; an explicit setting by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from ._lib import load, hooks_library, JnError, ElasParams, ScanParams, EXPORTS, LIB_PATH, HOOKS_LIB_PATH  # noqa: F401
from .elas import Elas  # noqa: F401
from . import node, device, parallel, navigate  # noqa: F401
from .sgm import Sgm, SGM_EXPORTS  # noqa: F401
from .bm import Bm, BM_EXPORTS  # noqa: F401
