#!/usr/bin/env python3
"""Does the step time drift under sustained load?  The C3 step (edges + match) back to back for ~40 s, the mean
time per step and per match launch printed for every chunk.    python3 tools/sustained.py [C3] [chunks] [steps per chunk]"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stereomatching_amd import pipeline  # noqa: E402
from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402
from tools._options import from_env  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
w, h, d, sw, mode = CONFIGS[cfg]
pairs = int(os.environ.get("PAIRS", "1"))            # pairs per step (C4: PAIRS=8)
l, r = make_pair(w, h, d, seed=1)
L = torch.from_numpy(l).cuda().unsqueeze(0).repeat(pairs, 1, 1).contiguous()
R = torch.from_numpy(r).cuda().unsqueeze(0).repeat(pairs, 1, 1).contiguous()
plan = pipeline.StereoPlan(w, h, d, sw, mode, max_pairs=pairs, options=from_env() or None)
web = torch.empty((pairs, h, w), dtype=torch.int32, device="cuda")
print("#", plan.describe(), flush=True)
t_start = time.perf_counter()
for c in range(chunks):
    plan.time_kernels(64, every=max(1, steps // 64))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        plan.run(L, R, 0.15, web=web)
        if i % 64 == 63:
            torch.cuda.synchronize()          # keep the launch queue short
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, n = plan.kernel_ms()
    print(f"t = {time.perf_counter() - t_start:6.1f} s: {dt / steps * 1e6:7.2f} us per step, match launch {ms * 1e3:7.2f} us ({n} timed)", flush=True)
