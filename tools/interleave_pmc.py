#!/usr/bin/env python3
"""Workload for a rocprofv3 --pmc pass: N match launches back to back, then N
match launches each behind a 64 MB fill (tools/interleave_probe.py found the
second kind ~25 us slower at 8 x 1080p).  Compare the per-dispatch counters."""
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
plan = hip.StereoPlan(w, h, d, sw, mode, max_pairs=P, options=__import__('tools._options', fromlist=['from_env']).from_env() or None)
print(plan.describe())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = C.c_void_p
check(lib.sm_find_edges(plan._h, vp(L.data_ptr()), vp(R.data_ptr()), 0.15, P, None, None, st))
N = 12
for _ in range(N):
    check(lib.sm_match_wta(plan._h, P, vp(web.data_ptr()), None, st))
torch.cuda.synchronize()
for _ in range(N):
    junk.fill_(1)
    check(lib.sm_match_wta(plan._h, P, vp(web.data_ptr()), None, st))
torch.cuda.synchronize()
