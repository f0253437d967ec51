"""The C-ABI library loads and exports every symbol include/jn_stereo.h declares; argument
checking that needs no GPU.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "jn_stereo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(jn_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_symbols_are_exported(jn):
    lib = jn.load()
    declared = _declared_functions()
    assert len(declared) >= 24
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(jn.EXPORTS) == declared, "python EXPORTS list out of sync with the header"
    with jn.hooks_library() as hooks:                         # the hooks build is the same ABI (tests and scripts swap it in)
        assert hooks is not lib and jn.load() is hooks
        assert not [n for n in declared if not hasattr(hooks, n)]
        assert hooks.jn_version() == lib.jn_version()
    assert jn.load() is lib


def test_params_default_mirror_reference_presets(jn, oracle):
    for setting in (0, 1):
        a = jn.Elas.parameters(setting)
        b = oracle.params(setting)
        for name, _ in a._fields_:
            assert getattr(a, name) == getattr(b, name), (setting, name)
    p = jn.Elas.parameters(0)
    assert (p.disp_max, p.candidate_stepsize, p.grid_size, p.speckle_size, p.ipol_gap_width) == (255, 5, 20, 200, 3)   # elas.h:92-115


def test_struct_sizes(jn):
    assert C.sizeof(jn.ElasParams) == 23 * 4
    from jackal_navigation_amd._lib import StageTimes
    assert C.sizeof(StageTimes) == 11 * 4


def test_unsupported_parameter_combinations_are_refused(jn):
    from jackal_navigation_amd import _lib
    L = jn.load()
    # (disp_min and subsampling are honoured since round 4: disp_min beyond disp_max and subsampling of odd-sized images are refused)
    for kw, (W, H) in (({"subsampling": 1}, (321, 180)), ({"subsampling": 1}, (320, 181)), ({"disp_max": 300}, (320, 180)), ({"disp_min": 300}, (320, 180)),
                       ({"ipol_gap_width": -1}, (320, 180))):
        p = jn.Elas.parameters(0, **kw)
        h = C.c_void_p()
        st = L.jn_elas_create(C.byref(p), W, H, 1, 0, 1, 1, C.byref(h))
        assert st == _lib.JN_ERR_UNSUPPORTED and not h.value, (kw, W, H)
    p = jn.Elas.parameters(0)
    h = C.c_void_p()
    assert L.jn_elas_create(C.byref(p), 8, 8, 1, 0, 1, 1, C.byref(h)) == _lib.JN_ERR_INVALID


def test_no_device_fails_loudly(jn):
    """Without a GPU the product must refuse to compute (there is no CPU fallback)."""
    from jackal_navigation_amd import _lib
    from jackal_navigation_amd.device import device_count
    if device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.JnError) as e:
        jn.Elas(jn.Elas.parameters(0), 320, 180)
    assert e.value.status == _lib.JN_ERR_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    """Nothing under the product package may import, link or dlopen anything under oracle/."""
    pkg = os.path.join(ROOT, "jackal_navigation_amd")
    for dirpath, _, files in os.walk(pkg):
        if "_build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(dirpath, f)


def _jn_strings(path):
    import re
    data = open(path, "rb").read()
    return sorted(set(m.decode() for m in re.findall(rb"JN_[A-Z][A-Z0-9_]{2,}", data)))


def test_release_library_carries_no_test_or_debug_switches(jn):
    """a product .so must not compute wrong disparities (or fail batches, or stall slots) because an environment variable
    leaked in.  The *_DBG profiling switches, the JN_TEST_* hooks and the A/B knobs exist only in the hooks build (csrc/hooks.h); what the
    release library reads is exactly what INTEGRATION.md section 9's first table lists."""
    import re
    rel = _jn_strings(jn.LIB_PATH)
    assert len(rel) <= 25, rel
    assert not [v for v in rel if v.endswith("_DBG") or v.startswith("JN_TEST_") or v in ("JN_SGM_EXP", "JN_DT_CLOCKS")], rel
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 9. Environment switches"):]
    first, second = sec.split("### 9b.")[0], sec.split("### 9b.")[1]
    listed = set(re.findall(r"`(JN_[A-Z0-9_]+)`", "\n".join(l.split("|")[1] for l in first.split("\n") if l.startswith("| `"))))
    assert set(rel) <= listed, sorted(set(rel) - listed)          # every switch that ships is documented
    hooks = _jn_strings(jn.HOOKS_LIB_PATH)
    assert set(rel) < set(hooks)                                   # the hooks build = the release switches + the hooks
    listed_hooks = set(re.findall(r"`(JN_[A-Z0-9_]+)`", "\n".join(l.split("|")[1] for l in second.split("\n") if l.startswith("| `"))))
    assert set(hooks) - set(rel) <= listed_hooks | {"JN_HOOKS"}, sorted(set(hooks) - set(rel) - listed_hooks)
    for v in ("JN_TEST_FAIL_SEQ", "JN_DENSE_DBG", "JN_OWNER_FAST_MAX"):
        assert v in hooks, v


def test_synth_generator_matches_appendix_a(jn, oracle):
    import numpy as np
    a = jn.node.synth_pair(320, 180, 48, 12345)
    b = oracle.synth_pair(320, 180, 48, 12345)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    c = jn.node.synth_pair(333, 77, 20, 99)
    d = oracle.synth_pair(333, 77, 20, 99)
    assert np.array_equal(c[0], d[0]) and np.array_equal(c[1], d[1])
