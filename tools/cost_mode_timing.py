import sys, time
sys.path.insert(0,'.')
import torch
from stereomatching_amd import pipeline
from stereomatching_amd.synth import CONFIGS, make_pair
for cfg in ("C2","C3"):
    w,h,d,sw,mode = CONFIGS[cfg]
    l,r = make_pair(w,h,d,seed=1)
    L,R = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    plan = pipeline.StereoPlan(w,h,d,sw,mode)
    for cost in ("sad","ssd"):
        for _ in range(3): plan.cost_wta(L,R,cost,want_best=False)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(10): plan.cost_wta(L,R,cost,want_best=False)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
        print(f"{cfg} {cost}: {dt*1e3:.3f} ms  {w*h*d/dt/1e6:.0f} Mpixel-disparities/s")
