#!/usr/bin/env python3
"""A few sm_find_edges launches of one configuration through ONE library build (a variant .so or the product), for
rocprofv3 passes:  python3 tools/edge_launch.py C3 [--lib stereomatching_amd/variants/e8.so] [--launches 6]"""
import argparse
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--lib", default=str(ROOT / "stereomatching_amd" / "libstereo_hip.so"))
ap.add_argument("--launches", type=int, default=6)
a = ap.parse_args()
w, h, d, sw, mode = CONFIGS[a.config]
l, r = make_pair(w, h, d, seed=1)
L, R = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
vp = C.c_void_p
lib = C.CDLL(a.lib)
lib.sm_find_edges.argtypes = [vp, vp, vp, C.c_double, C.c_int, vp, vp, vp]
plan = vp()
assert lib.sm_plan_create(0, w, h, d, sw, 1 if mode == "ghost" else 0, 1, C.byref(plan)) == 0
for _ in range(a.launches):
    assert lib.sm_find_edges(plan, L.data_ptr(), R.data_ptr(), 0.15, 1, None, None, None) == 0
torch.cuda.synchronize()
print(a.config, Path(a.lib).name, "done")
