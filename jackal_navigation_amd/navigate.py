"""Host-side mirror of the scan consumer in the reference's `navigate` node
(src/obstacle_avoidance/navigate.cpp), over libjn_stereo.so.

    laserScanCallback   (:344-363)  -> Navigator.scan_callback
    checkObstacle       (:101-153)  \\
    chooseDirection     (:155-197)   > Navigator.obstacle_avoid_step  (the decision part of obstacleAvoidMode :229-256)

Only the decision is mirrored: it is what makes a scan "the same" to the robot.
Velocity ramps, joystick modes and waypoint following are control logic outside the path.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import NavDecision, NavParams, NavState


class Navigator:
    def __init__(self, **overrides):
        L = _lib.load()
        self.params = NavParams()
        L.jn_nav_params_default(C.byref(self.params))
        for k, v in overrides.items():
            if not hasattr(self.params, k):
                raise AttributeError("jn_nav_params has no field %r" % k)
            setattr(self.params, k, v)
        self.state = NavState()
        L.jn_nav_state_reset(C.byref(self.state))
        self.laser_points = np.zeros((0, 2), np.float64)

    def scan_callback(self, msg):
        """msg: the dict node.laser_scan_message builds (ranges float32, angle_min/max float32)."""
        ranges = np.ascontiguousarray(msg["ranges"], np.float32)
        xy = np.zeros((len(ranges), 2), np.float64)
        k = _lib.load().jn_scan_to_points(ranges.ctypes.data, len(ranges), float(msg["angle_min"]), float(msg["angle_max"]), xy.ctypes.data)
        if k < 0:
            raise _lib.JnError(_lib.JN_ERR_INVALID, "jn_scan_to_points")
        self.laser_points = xy
        return xy

    def obstacle_avoid_step(self):
        """One control tick on the latest scan: returns the jn_nav_decision as a dict."""
        d = NavDecision()
        xy = self.laser_points
        _lib.check(_lib.load().jn_nav_vote(C.byref(self.params), C.byref(self.state), xy.ctypes.data if len(xy) else None, len(xy),
                                           C.byref(d)), "jn_nav_vote")
        return {n: getattr(d, n) for n, _ in d._fields_}
