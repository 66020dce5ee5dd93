"""SAD / SSD cost mode (parity unpinned): time per launch at the BASELINE configurations.
    python tools/cost_mode_timing.py [C2 C3 C5 ...] [WxHxDxS[g] ...] [sad] [ssd]     (g: ghost border)"""
import sys
sys.path.insert(0, '.')
import torch
from stereomatching_amd import pipeline
from stereomatching_amd.synth import CONFIGS, make_pair

from tools._options import from_env          # SM_COST_PX / SM_COST_TILE_H / SM_COST_KERNEL in this tool's environment
import re
for a in sys.argv[1:]:
    m = re.fullmatch(r"(\d+)x(\d+)x(\d+)x(\d+)(g?)", a)
    if m:
        CONFIGS[a] = (int(m[1]), int(m[2]), int(m[3]), int(m[4]), "ghost" if m[5] else "toroidal")
cfgs = [a for a in sys.argv[1:] if a in CONFIGS] or ["C2", "C3", "C5"]
costs = [a for a in sys.argv[1:] if a in ("sad", "ssd")] or ["sad", "ssd"]
for cfg in cfgs:
    w, h, d, sw, mode = CONFIGS[cfg]
    l, r = make_pair(w, h, d, seed=1)
    L, R = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    plan = pipeline.StereoPlan(w, h, d, sw, mode, options=from_env() or None)
    for cost in costs:
        for _ in range(5):
            plan.cost_wta(L, R, cost, want_best=False)
        torch.cuda.synchronize()
        n = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            plan.cost_wta(L, R, cost, want_best=False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"{cfg} ({w}x{h}, {d} shifts, {sw}x{sw}, {mode}) {cost}: {ms:.3f} ms per launch, "
              f"{w * h * d / ms / 1e3:.0f} Mpixel-disparities/s", flush=True)
    plan.close()
