"""The pipelined SGM mode with and without an open (idle) four-slot ELAS handle in the same process: the ELAS handle's streams hold
hardware queues, the SGM slots' streams then share queues (scripts/hwq_ab.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import jackal_navigation_amd as jn
from jackal_navigation_amd import node
W, H, D, B, NSL = 1280, 720, 128, 32, 4
dev = torch.device("cuda", 0)
Ls = np.empty((B, H, W), np.uint8); Rs = np.empty((B, H, W), np.uint8)
for b in range(B):
    Ls[b], Rs[b] = node.synth_pair(W, H, D, 12345 + b)
dL, dR = torch.from_numpy(Ls).to(dev), torch.from_numpy(Rs).to(dev)
outs = [torch.zeros((B, H, W), dtype=torch.int16, device=dev) for _ in range(NSL)]


def rate(m, reps=80):
    def run(k):
        for i in range(k):
            if i >= NSL:
                m.wait(i % NSL)
            m.submit_scan(i % NSL, B, dL.data_ptr(), dR.data_ptr(), W, H * W, outs[i % NSL].data_ptr())
        for s in range(NSL):
            m.wait(s)
    run(12); torch.cuda.synchronize()
    t = time.perf_counter(); run(reps); torch.cuda.synchronize()
    return B * reps / (time.perf_counter() - t)


res = {}
for with_elas in (True, False):
    e = jn.Elas(jn.Elas.parameters(jn.Elas.ROBOTICS, disp_max=D - 1), W, H, max_batch=B, device=0, host_threads=4, slots=4) if with_elas else None
    if e is not None:                                        # its slots' streams exist once they have run a batch
        D1 = torch.zeros((B, H, W), dtype=torch.float32, device=dev); D2 = torch.zeros_like(D1)
        for s in range(4):
            e.submit(s, B, dL.data_ptr(), dR.data_ptr(), W, H * W, D1.data_ptr(), D2.data_ptr())
        for s in range(4):
            e.wait(s)
    m = jn.Sgm(jn.Sgm.parameters(num_disparities=D), W, H, max_batch=B, device=0)
    res["next to an open ELAS handle" if with_elas else "alone"] = round(rate(m), 1)
    m.close()
    if e is not None:
        e.close()
print("SGM 1280x720 D=128 batch 32, four batches in flight, pairs/s:", res)
