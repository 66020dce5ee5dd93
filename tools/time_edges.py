#!/usr/bin/env python3
"""Time the edge-detection launch alone (both images of a pair).   python3 tools/time_edges.py C5 C3 REF4K:ghost ..."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stereomatching_amd import pipeline  # noqa: E402
from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402

for spec in sys.argv[1:] or ["C5"]:
    cfg, _, pairs = spec.partition("x")
    pairs = int(pairs or 1)
    w, h, d, sw, mode = CONFIGS[cfg]
    l, r = make_pair(w, h, d, seed=1)
    L = torch.from_numpy(l).cuda().unsqueeze(0).repeat(pairs, 1, 1).contiguous()
    R = torch.from_numpy(r).cuda().unsqueeze(0).repeat(pairs, 1, 1).contiguous()
    plan = pipeline.StereoPlan(w, h, d, sw, mode, max_pairs=pairs)
    for _ in range(50):
        plan.find_all_edges(L, R, 0.15, want_edges=False)
    torch.cuda.synchronize()
    best = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            plan.find_all_edges(L, R, 0.15, want_edges=False)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 10)
    print(f"{cfg} x{pairs} ({w}x{h}, {mode}): edges {min(best):7.2f} us per launch (median {sorted(best)[2]:.2f})", flush=True)
    plan.close()
