"""PCIe-inclusive rates through HOST pointers (never bench.py's `value`):
  * jn_elas_process: the literal drop-in seam, one synchronous call per pair;
  * jn_elas_submit_host: the streaming form, batches of host-resident pairs through the slots (copies of one slot overlap
    kernels of the others), pageable numpy buffers and pinned ones (torch pin_memory).
python3 scripts/host_pointer_rate.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jackal_navigation_amd as jn  # noqa: E402

for (W, H, D) in ((1280, 720, 128), (640, 480, 64)):
    L, R = jn.node.synth_pair(W, H, D, 12345)
    D1 = np.zeros((H, W), np.float32); D2 = np.zeros((H, W), np.float32)
    with jn.Elas(jn.Elas.parameters(0, disp_max=D - 1), W, H, host_threads=8) as e:
        for _ in range(5):
            e.process(L, R, D1, D2, (W, H, W))
        t = time.perf_counter(); N = 50
        for _ in range(N):
            e.process(L, R, D1, D2, (W, H, W))
        dt = (time.perf_counter() - t) / N
    print("%dx%d jn_elas_process (one synchronous call per pair): %.2f ms per call = %.0f pairs/s" % (W, H, dt * 1e3, 1 / dt))
    for (B, S, pinned) in ((1, 4, False), (4, 4, False), (8, 4, False), (8, 4, True), (16, 4, True)):
        def buf(shape, dtype):
            if not pinned:
                return np.zeros(shape, dtype)
            import torch
            return torch.zeros(shape, dtype=torch.uint8 if dtype == np.uint8 else torch.float32).pin_memory().numpy()
        slots = []
        for s in range(S):
            Ls, Rs = buf((B, H, W), np.uint8), buf((B, H, W), np.uint8)
            for b in range(B):
                Ls[b], Rs[b] = jn.node.synth_pair(W, H, D, 12345 + b)
            slots.append((Ls, Rs, buf((B, H, W), np.float32), buf((B, H, W), np.float32), (C.c_int32 * B)()))
        with jn.Elas(jn.Elas.parameters(0, disp_max=D - 1), W, H, max_batch=B, slots=S, host_threads=16) as e:
            def rounds(k):
                for s in range(S):
                    e.submit_host(s, *slots[s])
                for _ in range(k - 1):
                    for s in range(S):
                        e.wait(s); e.submit_host(s, *slots[s])
                for s in range(S):
                    e.wait(s)
            rounds(3)
            K = 12
            t = time.perf_counter(); rounds(K); dt = time.perf_counter() - t
        ok = all(list(x[4]) == [0] * B for x in slots)
        print("%dx%d jn_elas_submit_host batch %2d x %d slots, %s host buffers: %.0f pairs/s (%.2f ms per pair)%s" %
              (W, H, B, S, "pinned" if pinned else "pageable", K * S * B / dt, dt / (K * S * B) * 1e3, "" if ok else "  STATUS != 0"))
