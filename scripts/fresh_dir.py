#!/usr/bin/env python3
"""Empty a scratch directory under gpurun_out/ before a profiler run fills it (so that a summary never picks up an older run's
files):  python3 scripts/fresh_dir.py gpurun_out/<name> [...]      Refuses anything that is not a sub-directory of gpurun_out/."""
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scratch = os.path.join(root, "gpurun_out")
for arg in sys.argv[1:]:
    path = os.path.abspath(arg if os.path.isabs(arg) else os.path.join(root, arg))
    if os.path.dirname(path) != scratch or not os.path.basename(path):
        sys.exit("fresh_dir.py: %s is not a directory directly under gpurun_out/" % arg)
    shutil.rmtree(path, ignore_errors=True)
