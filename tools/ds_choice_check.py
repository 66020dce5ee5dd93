#!/usr/bin/env python3
"""Does the plan's choice of shifts per lane (16 / 8 / 4) hold up beyond the benchmark configurations?  A set of
shapes, each timed with the plan's own choice, with 4 shifts per lane forbidden and with 4 / 8 / 16 forced
(match launch, back to back, same device):   gpurun -- 'python3 tools/ds_choice_check.py > gpurun_out/ds_choice.txt'"""
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stereomatching_amd import pipeline  # noqa: E402
from stereomatching_amd.synth import make_pair  # noqa: E402

SHAPES = [(640, 480, 64, 9, 1), (1280, 720, 128, 9, 1), (1920, 1080, 128, 11, 1), (3840, 2160, 64, 7, 1),
          (2560, 1440, 96, 9, 1), (1920, 1080, 30, 21, 1), (800, 600, 32, 5, 1), (1920, 1080, 64, 7, 2),
          (1920, 1080, 64, 7, 4), (3840, 2160, 128, 13, 1), (3840, 2160, 32, 15, 1), (1024, 768, 100, 17, 1),
          (3840, 2160, 200, 9, 1), (960, 540, 48, 7, 3)]
SETTINGS = [("plan", None), ("no4", dict(no_four_shift_lanes=1)), ("ds4", dict(shifts_per_lane=4)),
            ("ds8", dict(shifts_per_lane=8)), ("ds16", dict(shifts_per_lane=16))]


def time_plan(plan, pairs, reps=7):
    web = None
    ts = []
    for r in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            web, _ = plan.match_wta(pairs, want_best=False, web=web)
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    return statistics.median(ts)


for w, h, d, sw, pairs in SHAPES:
    l, r = make_pair(w, h, d, seed=3)
    L = torch.from_numpy(l).cuda().unsqueeze(0).repeat(pairs, 1, 1).contiguous()
    R = torch.from_numpy(r).cuda().unsqueeze(0).repeat(pairs, 1, 1).contiguous()
    row, seen = [], {}
    for name, opts in SETTINGS:
        try:
            plan = pipeline.StereoPlan(w, h, d, sw, "toroidal", max_pairs=pairs, options=opts)
        except Exception:
            row.append(f"{name} n/a")
            continue
        desc = plan.describe().split("grid")[0]
        if desc in seen:
            row.append(f"{name} = {seen[desc]}")
            plan.close()
            continue
        plan.find_all_edges(L, R, want_edges=False)
        t = time_plan(plan, pairs)
        seen[desc] = name
        lanes = desc.split("shift-lanes of ")[1].split(")")[0] if "shift-lanes of " in desc else "?"
        row.append(f"{name} {t:7.1f} us (ds {lanes})")
        plan.close()
    print(f"{w}x{h} D={d} S={sw} x{pairs}: " + " | ".join(row), flush=True)
