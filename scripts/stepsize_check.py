"""GPU path against the oracle for candidate step sizes other than the presets' 5 (lattice geometry, support kernel segments and the
GPU arrangement all depend on it): python3 scripts/stepsize_check.py  (on the GPU box)"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jackal_navigation_amd as jn
from oracle.binding import Oracle
from jackal_navigation_amd.device import DeviceArray
o = Oracle(); bad = 0
for step in (3, 4, 6, 7, 9):
    for (W, H, sd) in ((640, 480, 60), (333, 201, 30)):
        n = 6
        Ls = np.stack([o.synth_pair(W, H, sd, 900 + b)[0] for b in range(n)]); Rs = np.stack([o.synth_pair(W, H, sd, 900 + b)[1] for b in range(n)])
        dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
        d1 = DeviceArray.from_numpy(np.zeros((n, H, W), np.float32)); d2 = DeviceArray.from_numpy(np.zeros((n, H, W), np.float32))
        kw = dict(disp_max=95, candidate_stepsize=step)
        with jn.Elas(jn.Elas.parameters(0, **kw), W, H, max_batch=n, host_threads=2) as e:      # 2 threads < 4n: one part, GPU arrangement
            for rep in range(2):                                                               # second batch: LDS sized by the hint
                st = e.process_batch(n, dL.ptr, dR.ptr, W, H * W, d1.ptr, d2.ptr)
        for b in range(n):
            so, D1o, D2o = o.process(o.params(0, **kw), Ls[b], Rs[b])
            okb = so == st[b] and (so != 0 or (np.array_equal(d1.numpy()[b].view(np.uint32), D1o.view(np.uint32)) and np.array_equal(d2.numpy()[b].view(np.uint32), D2o.view(np.uint32))))
            bad += 0 if okb else 1
        print("step", step, W, H, "status", st, "mismatches so far", bad, flush=True)
print("PASSED" if bad == 0 else "FAILED")
