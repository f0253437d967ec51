"""Create / use / destroy handles repeatedly and from several threads; device memory must come back."""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jackal_navigation_amd as jn
import torch

def free_mb():
    f, t = torch.cuda.mem_get_info(0)
    return f / 2**20

W, H = 640, 360
L, R = jn.node.synth_pair(W, H, 64, 5)
base = None
for it in range(25):
    D1 = np.zeros((H, W), np.float32); D2 = np.zeros((H, W), np.float32)
    with jn.Elas(jn.Elas.parameters(0, disp_max=95), W, H, max_batch=4, slots=3, host_threads=4) as e:
        assert e.process(L, R, D1, D2, (W, H, W)) == 0
    if it == 2:
        base = free_mb()
print("free MB after 3 cycles %.0f, after 25 cycles %.0f" % (base, free_mb()))
assert abs(free_mb() - base) < 64, "device memory leak"
ref = D1.copy()

def worker(k, out):
    Da = np.zeros((H, W), np.float32); Db = np.zeros((H, W), np.float32)
    with jn.Elas(jn.Elas.parameters(0, disp_max=95), W, H, max_batch=2, slots=2, host_threads=2) as e:
        for _ in range(10):
            assert e.process(L, R, Da, Db, (W, H, W)) == 0
    out[k] = np.array_equal(Da, ref)

res = {}
ths = [threading.Thread(target=worker, args=(k, res)) for k in range(4)]
[t.start() for t in ths]; [t.join() for t in ths]
print("4 handles used from 4 threads concurrently:", res)
assert all(res.values())
print("lifecycle OK")
