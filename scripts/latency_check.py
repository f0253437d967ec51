"""640x480 D=64 batch-1 latency (BASELINE config 2) and the host-pointer drop-in rate, for A/B runs inside gpurun."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jackal_navigation_amd as jn
from jackal_navigation_amd import node
dev = torch.device("cuda", 0)
if os.environ.get("PIN"):
    from jackal_navigation_amd import parallel
    print("pin:", parallel.pin_rank(0, 1, [0]))
big = None
if os.environ.get("BIG"):                      # an idle 4-slot / 16-thread handle next to the lone one, as in bench.py
    big = jn.Elas(jn.Elas.parameters(jn.Elas.ROBOTICS, disp_max=127), 1280, 720, max_batch=32, device=0, host_threads=16, slots=4)
configs = ((640, 480, 64),) if os.environ.get("LONE_ONLY") else ((640, 480, 64), (1280, 720, 128))
for (w2, h2, d2) in configs:
    l2, r2 = node.synth_pair(w2, h2, d2, 12345)
    tl, tr = torch.from_numpy(l2).to(dev), torch.from_numpy(r2).to(dev)
    o1 = torch.zeros((h2, w2), dtype=torch.float32, device=dev); o2 = torch.zeros_like(o1)
    e2 = jn.Elas(jn.Elas.parameters(jn.Elas.ROBOTICS, disp_max=d2 - 1), w2, h2, max_batch=1, device=0, host_threads=int(os.environ.get("HT", "2")), slots=1)
    for _ in range(10):
        e2.process_batch(1, tl.data_ptr(), tr.data_ptr(), w2, h2 * w2, o1.data_ptr(), o2.data_ptr())
    torch.cuda.synchronize()
    each, host = [], []
    t1 = time.perf_counter()
    for _ in range(300):
        t2 = time.perf_counter()
        e2.process_batch(1, tl.data_ptr(), tr.data_ptr(), w2, h2 * w2, o1.data_ptr(), o2.data_ptr())
        each.append(time.perf_counter() - t2); host.append(e2.last_times()["host_stage"])
    torch.cuda.synchronize()
    lat = (time.perf_counter() - t1) / 300
    each.sort(); host.sort()
    print("%dx%d device pointers: %.3f ms/pair mean, p10 %.3f median %.3f p90 %.3f; host stage median %.3f   stages %s" %
          (w2, h2, lat * 1e3, each[30] * 1e3, each[150] * 1e3, each[270] * 1e3, host[150], {k: round(v, 3) for k, v in e2.last_times().items()}))
    if os.environ.get("LONE_ONLY"):
        e2.close(); continue
    D1 = np.zeros((h2, w2), np.float32); D2 = np.zeros((h2, w2), np.float32)
    for _ in range(5):
        e2.process(l2, r2, D1, D2, (w2, h2, w2))
    t1 = time.perf_counter()
    for _ in range(50):
        e2.process(l2, r2, D1, D2, (w2, h2, w2))
    lat = (time.perf_counter() - t1) / 50
    print("%dx%d host pointers (jn_elas_process): %.3f ms/pair = %.0f pairs/s" % (w2, h2, lat * 1e3, 1 / lat))
    e2.close()
