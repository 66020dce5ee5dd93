#!/usr/bin/env python3
"""Why is the match launch slower right after the edge launch?  Times the match
kernel (events around it) with different things in front of it on the stream."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stereomatching_amd import pipeline as hip  # noqa: E402
from stereomatching_amd.capi import check, lib  # noqa: E402
from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h, d, sw, mode = CONFIGS[cfg]
left, right = make_pair(w, h, d, seed=1)
L = torch.from_numpy(left).cuda().repeat(P, 1, 1).contiguous()
R = torch.from_numpy(right).cuda().repeat(P, 1, 1).contiguous()
web = torch.empty((P, h, w), dtype=torch.int32, device="cuda")
junk = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
small = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
plan = hip.StereoPlan(w, h, d, sw, mode, max_pairs=P, options=__import__('tools._options', fromlist=['from_env']).from_env() or None)
print(plan.describe())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = C.c_void_p


def edges(p):
    check(lib.sm_find_edges(plan._h, vp(L.data_ptr()), vp(R.data_ptr()), 0.15, p, None, None, st))


def match():
    check(lib.sm_match_wta(plan._h, P, vp(web.data_ptr()), None, st))


edges(P)
fronts = {
    "nothing": lambda: None,
    "edges of all pairs": lambda: edges(P),
    "64 MB fill": lambda: junk.fill_(1),
    "1 MB fill": lambda: small.fill_(1),
    "64 MB read (sum)": lambda: junk.sum(),
    "sleep 200 us (host)": lambda: torch.cuda._sleep(400000),
}
N = 100
for name, front in fronts.items():
    for rep in range(2):
        plan.time_kernels(N)
        for _ in range(N):
            front()
            match()
        torch.cuda.synchronize()
        ms, n = plan.kernel_ms()
    print(f"{name:24s} match kernel {ms*1e3:8.1f} us")
edges(P)
