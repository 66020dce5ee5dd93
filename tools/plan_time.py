import sys, time
sys.path.insert(0,'.')
import torch
from stereomatching_amd import pipeline
torch.cuda.init()
for cfg in [(3840,2160,128,9,"toroidal"),(240,135,30,21,"toroidal"),(1920,1080,64,7,"ghost")]:
    t0=time.perf_counter(); p=pipeline.StereoPlan(*cfg); t1=time.perf_counter(); p.close()
    print(cfg, f"plan create {1e3*(t1-t0):.2f} ms")
