import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import jackal_navigation_amd as jn
for (W,H,D) in ((1280,720,128),(640,480,64)):
    L,R = jn.node.synth_pair(W,H,D,12345)
    D1=np.zeros((H,W),np.float32); D2=np.zeros((H,W),np.float32)
    with jn.Elas(jn.Elas.parameters(0,disp_max=D-1),W,H,host_threads=2) as e:
        for _ in range(5): e.process(L,R,D1,D2,(W,H,W))
        t=time.perf_counter(); N=50
        for _ in range(N): e.process(L,R,D1,D2,(W,H,W))
        dt=(time.perf_counter()-t)/N
    print(W,H,"host-pointer process: %.2f ms per call = %.0f pairs/s"%(dt*1e3,1/dt))
