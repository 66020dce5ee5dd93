#!/usr/bin/env python3
"""Where the step time goes beyond the two kernels: event-record overhead of
sm_plan_time_kernels, and launch gaps with / without a HIP graph (one device)."""
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stereomatching_amd import pipeline as hip  # noqa: E402
from stereomatching_amd.capi import check, lib  # noqa: E402
from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w, h, d, sw, mode = CONFIGS[cfg]
if len(sys.argv) > 3 and sys.argv[3] == "distinct":
    prs = [make_pair(w, h, d, seed=1 + i) for i in range(P)]
    import numpy as np
    L = torch.from_numpy(np.stack([p[0] for p in prs])).cuda().contiguous()
    R = torch.from_numpy(np.stack([p[1] for p in prs])).cuda().contiguous()
else:
    left, right = make_pair(w, h, d, seed=1)
    L = torch.from_numpy(left).cuda().repeat(P, 1, 1).contiguous()
    R = torch.from_numpy(right).cuda().repeat(P, 1, 1).contiguous()
web = torch.empty((P, h, w), dtype=torch.int32, device="cuda")
plan = hip.StereoPlan(w, h, d, sw, mode, max_pairs=P, options=__import__('tools._options', fromlist=['from_env']).from_env() or None)
print(plan.describe())
s = torch.cuda.Stream()
N = 200


def run(n, stream):
    for _ in range(n):
        check(lib.sm_run(plan._h, C.c_void_p(L.data_ptr()), C.c_void_p(R.data_ptr()), 0.15, P,
                         C.c_void_p(web.data_ptr()), None, C.c_void_p(stream.cuda_stream)))


def timed(label, fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / N * 1e6)
    print(f"{label:40s} {best:8.2f} us/step")


with torch.cuda.stream(s):
    plan.time_kernels(0)
    timed("stream launches, no events", lambda: run(N, s))
    plan.time_kernels(N)
    def with_events():
        plan.time_kernels(N); run(N, s)
    timed("stream launches, events around match", with_events)
    print("   kernel_ms", plan.kernel_ms())
    # match launches only, back to back
    def only_match():
        plan.time_kernels(N)
        for _ in range(N):
            check(lib.sm_match_wta(plan._h, P, C.c_void_p(web.data_ptr()), None, C.c_void_p(s.cuda_stream)))
    timed("match launches only, events", only_match)
    print("   kernel_ms", plan.kernel_ms())
    plan.time_kernels(0)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        run(20, s)
    timed("graph of 20 steps x 10 replays", lambda: [g.replay() for _ in range(N // 20)])
