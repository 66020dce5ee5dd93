import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from stereomatching_amd import pipeline
from stereomatching_amd.synth import make_pair
from tests import oracle
for (w,h,d,sw,mode) in [(7680,4320,30,21,"toroidal"),(7680,4320,64,9,"ghost"),(5003,1201,100,7,"toroidal"),(4098,2161,128,11,"ghost")]:
    left,right = make_pair(w,h,d,seed=5)
    plan = pipeline.StereoPlan(w,h,d,sw,mode)
    L,R = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    el,er = plan.find_all_edges(L,R,0.15)
    web,best = plan.match_wta(1)
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(5): plan.run(L,R,0.15,web=web)
    torch.cuda.synchronize(); dt=(time.time()-t0)/5
    half=sw//2
    y0,y1 = h//2-5, h//2+11
    a,b = y0-half, y1+half
    elh,erh = el.cpu().numpy()[0], er.cpu().numpy()[0]
    assert np.array_equal(elh[a+1:b-1], oracle.find_all_edges(left[a:b],0.15,mode)[1:-1])
    ob,ow = oracle.hot_path(elh[a:b], erh[a:b], d, sw, mode)
    assert np.array_equal(web.cpu().numpy()[0][y0:y1], ow[half:half+16]), (w,h,d,sw)
    assert np.array_equal(best.cpu().numpy()[0][y0:y1], ob[half:half+16])
    print(w,h,d,sw,mode,"OK  step %.3f ms  %.2f T px-d/s |" % (dt*1e3, w*h*d/dt/1e12), plan.describe()[:60])
    plan.close()
