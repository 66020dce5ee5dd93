#!/usr/bin/env python3
"""Sweep the tile height of the tiled match kernel (SM_TILE_H override) and
time the launch with HIP events.  GPU tuning aid, not part of the product path.

    python tools/tune_tile_h.py C3 24 32 43 48 64 86 128
"""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stereomatching_amd import pipeline  # noqa: E402
from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402


def time_plan(cfg, th, pairs=1, iters=20):
    w, h, d, sw, mode = CONFIGS[cfg]
    from tools._options import from_env
    opts = from_env()
    if th:
        opts["tile_h"] = th
    plan = pipeline.StereoPlan(w, h, d, sw, mode, max_pairs=pairs, options=opts or None)
    left, right = make_pair(w, h, d, seed=1)
    L = torch.from_numpy(left).cuda().repeat(pairs, 1, 1).contiguous()
    R = torch.from_numpy(right).cuda().repeat(pairs, 1, 1).contiguous()
    plan.find_all_edges(L, R, 0.15, want_edges=False)
    web = torch.empty((pairs, h, w), dtype=torch.int32, device="cuda")
    for _ in range(3):
        plan.match_wta(pairs, want_best=False, web=web)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        plan.match_wta(pairs, want_best=False, web=web)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    desc = plan.describe()
    plan.close()
    return ts[len(ts) // 2], ts[0], desc


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    ths = [int(a) for a in sys.argv[2:]] or [0]
    pairs = int(os.environ.get("PAIRS", "1"))
    for th in [0] + ths:
        med, mn, desc = time_plan(cfg, th, pairs)
        print(f"{cfg} SM_TILE_H={th or 'model':>5} median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us  | {desc}", flush=True)
