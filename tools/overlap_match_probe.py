#!/usr/bin/env python3
"""Does the TAIL of one match launch (the younger wave of every SIMD pair finishing alone, DESIGN
5.1) fill with the head of the next one when consecutive launches sit on different streams?
Timing experiment only (no ordering between the streams; every launch reads the same ext image and
writes its own map).    python3 tools/overlap_match_probe.py [C3] [pairs]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stereomatching_amd import pipeline  # noqa: E402
from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w, h, d, sw, mode = CONFIGS[cfg]
l, r = make_pair(w, h, d, seed=1)
L = torch.from_numpy(l).cuda().unsqueeze(0).repeat(pairs, 1, 1).contiguous()
R = torch.from_numpy(r).cuda().unsqueeze(0).repeat(pairs, 1, 1).contiguous()
from tools._options import from_env  # noqa: E402

plan = pipeline.StereoPlan(w, h, d, sw, mode, max_pairs=pairs, options=from_env() or None)   # SM_TILE_H=8 ... as the other tools
plan.find_all_edges(L, R, 0.15, want_edges=False)
print("#", plan.describe())
NS = 4
webs = [torch.empty((pairs, h, w), dtype=torch.int32, device="cuda") for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
se = torch.cuda.Stream()
N = 240


def run(n_streams, with_edges):
    for _ in range(20):
        plan.match_wta(pairs, want_best=False, web=webs[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        if with_edges:
            with torch.cuda.stream(se if n_streams > 1 else streams[0]):
                plan.find_all_edges(L, R, 0.15, want_edges=False)
        with torch.cuda.stream(streams[i % n_streams]):
            plan.match_wta(pairs, want_best=False, web=webs[i % n_streams])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


for rep in range(3):
    print(f"{cfg} x{pairs} match only : " + "  ".join(f"{n} stream(s) {run(n, False):7.1f} us" for n in (1, 2, 3, 4)), flush=True)
    print(f"{cfg} x{pairs} edges+match: " + "  ".join(f"{n} stream(s) {run(n, True):7.1f} us" for n in (1, 2, 3, 4)), flush=True)
