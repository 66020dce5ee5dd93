#!/usr/bin/env python3
"""Same-device A/B of library builds: every variant .so under
stereomatching_amd/variants/ is loaded into THIS process and the match launch is
timed in interleaved rounds (timings from different gpurun boxes differ by up
to ~10 %, so variants must never be compared across calls).

    python tools/ab_variants.py [C3] [pairs] [rounds]

The default loop times the match launch after ITSELF.  That ranks kernel builds correctly
(instruction counts, staging, tiling) but NOT wave-priority schedules: the match launch that
follows the edge kernel finds its workgroups in other slots (round 3: a priority class that won
1-2 % here lost 5 % in the real step).  For those use AB_STEP=1 (edges, then match, per
iteration) or tools/sustained_ab.sh.
"""
import ctypes as C
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
w, h, d, sw, mode = CONFIGS[cfg]
left, right = make_pair(w, h, d, seed=1)
L = torch.from_numpy(left).cuda().repeat(pairs, 1, 1).contiguous()
R = torch.from_numpy(right).cuda().repeat(pairs, 1, 1).contiguous()
web = torch.empty((pairs, h, w), dtype=torch.int32, device="cuda")
vp = C.c_void_p

import os  # noqa: E402

libs = {}
entries = []
for so in sorted((ROOT / "stereomatching_amd" / "variants").glob("*.so")):
    entries.append((so.stem, so, {}))
    if os.environ.get("AB_ENVS"):          # e.g. AB_ENVS="SM_DS=8;SM_DUO=1,SM_TILE_H=32"
        for spec in os.environ["AB_ENVS"].split(";"):
            envs_of = dict(kv.split("=") for kv in spec.split(","))     # "A=1,B=2": both at once
            entries.append((f"{so.stem}@{spec}", so, envs_of))
from tools._options import struct_from_spec  # noqa: E402

for name_, so, envs in entries:
    lib = C.CDLL(str(so))
    lib.sm_last_error.restype = C.c_char_p
    lib.sm_find_edges.argtypes = [vp, vp, vp, C.c_double, C.c_int, vp, vp, vp]
    lib.sm_match_wta.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.sm_cost_wta.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    plan = vp()
    opts = struct_from_spec(envs)          # (variants are chosen through sm_plan_create_ex, not the environment)
    assert lib.sm_plan_create_ex(0, w, h, d, sw, 1 if mode == "ghost" else 0, pairs, C.byref(opts), C.byref(plan)) == 0
    assert lib.sm_find_edges(plan, L.data_ptr(), R.data_ptr(), 0.15, pairs, None, None, None) == 0
    libs[name_] = (lib, plan, dict(envs))
    if os.environ.get("AB_DESCRIBE"):
        lib.sm_plan_describe.restype = C.c_char_p
        lib.sm_plan_describe.argtypes = [vp]
        print(f"# {name_}: {lib.sm_plan_describe(plan).decode()}")
torch.cuda.synchronize()

ref = None
times = {k: [] for k in libs}
for r in range(rounds + 1):
    # alternate the order: a launch is up to ~2 % faster behind launches of the SAME kernel
    # shape than behind another one's, and a fixed order would credit that to one entry
    order = list(libs.items())
    if r % 2 == 0:
        order.reverse()
    for name, (lib, plan, envs_) in order:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            if os.environ.get("AB_COST"):      # the cost mode (sad / ssd) on the gray images
                assert lib.sm_cost_wta(plan, L.data_ptr(), R.data_ptr(), 1 if os.environ["AB_COST"] == "sad" else 2,
                                       pairs, web.data_ptr(), None, None) == 0
            elif os.environ.get("AB_STEP"):    # the whole step: edges, then match
                assert lib.sm_find_edges(plan, L.data_ptr(), R.data_ptr(), 0.15, pairs, None, None, None) == 0
                assert lib.sm_match_wta(plan, pairs, web.data_ptr(), None, None) == 0
            elif os.environ.get("AB_EDGES"):
                assert lib.sm_find_edges(plan, L.data_ptr(), R.data_ptr(), 0.15, pairs, None, None, None) == 0
            else:
                assert lib.sm_match_wta(plan, pairs, web.data_ptr(), None, None) == 0
        e1.record()
        torch.cuda.synchronize()
        if r:
            times[name].append(e0.elapsed_time(e1) / 5 * 1e3)
        if os.environ.get("AB_EDGES"):
            assert lib.sm_match_wta(plan, pairs, web.data_ptr(), None, None) == 0
            torch.cuda.synchronize()
        if ref is None:
            ref = web.clone()
        else:
            assert os.environ.get("AB_NOCHECK") or torch.equal(web, ref), f"{name} differs"
for name, t in times.items():
    print(f"{cfg} x{pairs} {name:16s} median {statistics.median(t):8.1f} us  min {min(t):8.1f} us")
