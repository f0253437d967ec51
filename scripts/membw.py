"""Achievable HBM bandwidth on this GPU for plain streaming kernels (torch fill / copy), as a yardstick for the
bandwidth-bound kernels of the path.  Usage (inside gpurun): python3 scripts/membw.py"""
import time
import torch

dev = torch.device("cuda", 0)
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    for name, fn, bytes_moved in (("fill (write)", lambda: a.fill_(1.0), 4 * n), ("copy (read+write)", lambda: b.copy_(a), 8 * n),
                                  ("sum (read)", lambda: a.sum(), 4 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("%5d MB  %-18s %7.1f us  %6.2f TB/s" % (mb, name, dt * 1e6, bytes_moved / dt / 1e12))
