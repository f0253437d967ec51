import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jackal_navigation_amd as jn
from jackal_navigation_amd.device import DeviceArray
from oracle.binding import Oracle
o = Oracle()
W, H, n, D = 1280, 720, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 128
Ls = np.zeros((n, H, W), np.uint8); Rs = np.zeros((n, H, W), np.uint8)
for b in range(n):
    Ls[b], Rs[b] = jn.node.synth_pair(W, H, D, 12345 + b)
dL, dR = DeviceArray.from_numpy(Ls), DeviceArray.from_numpy(Rs)
outs = []
with jn.Elas(jn.Elas.parameters(0, disp_max=D - 1), W, H, max_batch=n, slots=1) as e:
    for rep in range(4):
        dD1 = DeviceArray.from_numpy(np.zeros((n, H, W), np.float32)); dD2 = DeviceArray.from_numpy(np.zeros((n, H, W), np.float32))
        st = e.process_batch(n, dL.ptr, dR.ptr, W, H * W, dD1.ptr, dD2.ptr)
        outs.append((dD1.numpy(), dD2.numpy()))
        dD1.free(); dD2.free()
for rep in range(1, 4):
    for k, name in ((0, "D1"), (1, "D2")):
        diff = outs[0][k] != outs[rep][k]
        if diff.any():
            frames = np.unique(np.argwhere(diff)[:, 0])
            idx = np.argwhere(diff)
            print("rep", rep, name, "differs:", int(diff.sum()), "px in frames", frames.tolist()[:10], "first", idx[:5].tolist(),
                  [float(outs[0][k][tuple(i)]) for i in idx[:5]], [float(outs[rep][k][tuple(i)]) for i in idx[:5]])
        else:
            print("rep", rep, name, "identical")
po = o.params(0, disp_max=D - 1)
for b in range(min(n, 6)):
    _, D1o, D2o = o.process(po, Ls[b], Rs[b])
    for rep in range(4):
        d1 = (outs[rep][0][b] != D1o).sum(); d2 = (outs[rep][1][b] != D2o).sum()
        if d1 or d2:
            idx = np.argwhere(outs[rep][0][b] != D1o)
            print("frame", b, "rep", rep, "vs oracle: D1 diff", int(d1), "D2 diff", int(d2), idx[:4].tolist())
print("done")
