#!/usr/bin/env python3
"""Can the edge kernel of the NEXT pair hide beside the match kernel of the current one?
Timing experiment only (no ordering between the two streams: the maps are garbage): N match launches
on one stream and N edge launches on another, issued alternately by one host thread, against the
same launches on one stream.    python3 tools/overlap_probe.py [C3]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stereomatching_amd import pipeline  # noqa: E402
from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
w, h, d, sw, mode = CONFIGS[cfg]
l, r = make_pair(w, h, d, seed=1)
L, R = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
plan = pipeline.StereoPlan(w, h, d, sw, mode)
plan.find_all_edges(L, R, 0.15, want_edges=False)
web = torch.empty((1, h, w), dtype=torch.int32, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
N = 200


def run(two_streams):
    for _ in range(30):
        plan.find_all_edges(L, R, 0.15, want_edges=False)
        plan.match_wta(1, want_best=False, web=web)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        with torch.cuda.stream(sb if two_streams else sa):
            plan.find_all_edges(L, R, 0.15, want_edges=False)
        with torch.cuda.stream(sa):
            plan.match_wta(1, want_best=False, web=web)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


for rep in range(3):
    print(f"{cfg}: one stream {run(False):7.1f} us per (edges + match);  two streams, unordered {run(True):7.1f} us", flush=True)
