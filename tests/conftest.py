import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The library sends batches below ~10 M pixels through the four separate gap / adaptive-mean kernels and larger ones
    # through the fused pass (csrc/kernels.hip gap_mean_fusable).  The GPU tests mostly use small frames: run them through the
    # fused pass (the one the bench uses) unless a test chooses; test_both_post_processing_routes covers the other one.
    os.environ.setdefault("JN_POST_FUSED_MIN_PIXELS", "0")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement under oracle/ (checker only)."""
    from oracle.binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The compiled reference (oracle/_ref); tests that need it are skipped where it was not built."""
    from oracle.binding import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libelas_ref.so not built (needs /root/reference)")
    return Reference()        # zero-filled worker process, see oracle.binding.Reference


@pytest.fixture(scope="session")
def jn():
    """The product package; fails loudly (no skip) if libjn_stereo.so is missing."""
    import jackal_navigation_amd as jn
    jn.load()
    return jn


@pytest.fixture
def hooks(jn):
    """For the duration of the test every call of the package goes to the HOOKS build (libjn_stereo_hooks.so, csrc/hooks.h): the
    JN_TEST_* hooks, the route-forcing knobs (JN_OWNER_FAST_MAX, JN_FILTER_LDS_KB, JN_BM_BAND ...) exist only there.  A test that takes this
    fixture says by that that it does NOT exercise the release library; everything else does."""
    with jn.hooks_library() as L:
        yield L


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    return np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.fixture(scope="session")
def same():
    return bits_equal
