"""The evidence tooling under scripts/ runs on the GPU box only; here: every shell script parses, every Python script compiles, and the
scripts a round's evidence run calls exist (a typo there costs a GPU session, not a test run)."""
import glob
import os
import py_compile
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shell_scripts_parse():
    for p in sorted(glob.glob(os.path.join(ROOT, "scripts", "*.sh"))):
        r = subprocess.run(["bash", "-n", p], capture_output=True, text=True)
        assert r.returncode == 0, (p, r.stderr)


def test_python_scripts_compile(tmp_path):
    for p in sorted(glob.glob(os.path.join(ROOT, "scripts", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        py_compile.compile(p, cfile=str(tmp_path / (os.path.basename(p) + "c")), doraise=True)


def test_the_evidence_run_calls_scripts_that_exist():
    tag = open(os.path.join(ROOT, "profiles", "CURRENT")).read().strip()
    runner = os.path.join(ROOT, "scripts", "round%d_profiles.sh" % int(tag[1:]))
    assert os.path.exists(runner), runner
    text = open(runner).read()
    for name in set(re.findall(r"scripts/([A-Za-z0-9_/]+\.(?:sh|py|hip))", text)):
        assert os.path.exists(os.path.join(ROOT, "scripts", name)), name
