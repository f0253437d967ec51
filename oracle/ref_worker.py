"""Worker process behind oracle.binding.Reference — TEST INFRASTRUCTURE.

Started with MALLOC_PERTURB_ set, so that every allocation libelas makes is pre-filled with one known byte (see the
docstring of oracle.binding.Reference for the reference lines that read uninitialised descriptor bytes).  Protocol:
pickled (object id, method, args, kwargs) on stdin -> pickled ("val", result) | ("obj", id) | ("err", text) on stdout."""
import os
import pickle
import sys
import traceback

import numpy as np  # noqa: F401  (results are numpy arrays)

from oracle.binding import LocalReference, Params, RefSession


class _Root:
    def __init__(self):
        self.ref = LocalReference()

    def params_bytes(self, setting):
        return bytes(self.ref.params(setting))

    def process(self, pbytes, I1, I2, fill):
        return self.ref.process(Params.from_buffer_copy(pbytes), I1, I2, fill)

    def sobel(self, I):
        return self.ref.sobel(I)

    def triangulate(self, xy):
        return self.ref.triangulate(xy)

    def open(self, pbytes, I1, I2):
        return self.ref.open(Params.from_buffer_copy(pbytes), I1, I2)


def main():
    inp, out = sys.stdin.buffer, os.fdopen(os.dup(1), "wb")
    os.dup2(2, 1)                                   # libelas prints to stdout ("ERROR: Need at least 3 support points!")
    objs = {0: _Root()}
    next_id = 1
    while True:
        try:
            oid, name, args, kw = pickle.load(inp)
        except EOFError:
            return
        try:
            if oid is None and name == "_drop":
                objs.pop(args[0], None)
                res = ("val", None)
            else:
                val = getattr(objs[oid], name)(*args, **kw)
                if isinstance(val, RefSession):
                    objs[next_id] = val
                    res = ("obj", next_id)
                    next_id += 1
                else:
                    res = ("val", val)
        except Exception:
            res = ("err", traceback.format_exc())
        pickle.dump(res, out, protocol=4)
        out.flush()


if __name__ == "__main__":
    main()
