"""ros/point_cloud_node.cpp through a compiler front end (seam B3).

The image has no ROS, OpenCV or popt, so the node cannot be built here.  tests/mocks/ holds declaration-only stand-ins for the
headers it includes (signatures of the real APIs, nothing linked); include/jn_stereo.h is the REAL header, so a wrong C-ABI call,
a typo or a type error in the node fails this test.  A catkin build on a robot image is still what B3 needs."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = os.path.join(ROOT, "ros", "point_cloud_node.cpp")


def _compilers():
    out = []
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("g++")):
        if c and os.path.exists(c):
            out.append(c)
    return out


def _syntax(cxx, path):
    return subprocess.run([cxx, "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tests", "mocks"),
                           "-I" + os.path.join(ROOT, "include"), path], capture_output=True, text=True)


@pytest.mark.parametrize("cxx", _compilers())
def test_node_source_compiles_against_the_real_c_abi_header(cxx):
    r = _syntax(cxx, NODE)
    assert r.returncode == 0, r.stderr[-4000:]


def test_the_check_has_teeth(tmp_path):
    """the same source with one C-ABI call given a wrong argument list must be rejected"""
    cxx = _compilers()[0]
    src = open(NODE).read()
    assert "jn_disparity_scan(0, &sp_, 1, d_D1_, d_lut_, W, H, d_u8_, d_bins_, d_meta_)" in src
    bad = tmp_path / "bad_node.cpp"
    bad.write_text(src.replace("jn_disparity_scan(0, &sp_, 1, d_D1_, d_lut_, W, H, d_u8_, d_bins_, d_meta_)",
                               "jn_disparity_scan(0, &sp_, 1, d_D1_, d_lut_, W, H, d_u8_, d_bins_)").replace('#include "jn_stereo.h"', '#include "jn_stereo.h"\n'))
    r = _syntax(cxx, str(bad))
    assert r.returncode != 0 and "jn_disparity_scan" in r.stderr


def test_node_keeps_the_reference_surface():
    """node name, topics and flags of point_cloud.cpp:499-528, 567-568 (launch files and `navigate` depend on them)"""
    src = open(NODE).read()
    for needle in ('"jackal_obstacle_avoidance"', '"/webcam/left/image_raw/compressed"', '"/webcam/right/image_raw/compressed"',
                   '"/webcam/left/depth_map"', '"/webcam/left/obstacle_scan"', '"/webcam/left/point_cloud"', '"/jackal/time_log"',
                   '"img-height", \'h\'', '"calib-file", \'c\'', '"gen-pcl", \'g\'', '"logging", \'l\'', '"calib-extrinsic", \'m\''):
        assert needle in src, needle
