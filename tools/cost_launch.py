#!/usr/bin/env python3
"""A few cost-mode (SAD / SSD, parity unpinned) launches of one configuration, for rocprofv3 passes.
    python3 tools/cost_launch.py C3 sad [--launches 5] [--px N --tile-h N]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from stereomatching_amd import pipeline  # noqa: E402
from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("cost", nargs="?", default="sad")
ap.add_argument("--launches", type=int, default=5)
ap.add_argument("--px", type=int, default=0)
ap.add_argument("--tile-h", type=int, default=0)
ap.add_argument("--kernel", type=int, default=0, help="1 = the general masked kernel")
ap.add_argument("--waves", type=int, default=0, help="waves per workgroup (k_sad_pc, k_ssd_mfma): 1, 2 or 4")
a = ap.parse_args()
w, h, d, sw, mode = CONFIGS[a.config]
l, r = make_pair(w, h, d, seed=1)
L, R = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
plan = pipeline.StereoPlan(w, h, d, sw, mode, options=dict(cost_pixels_per_lane=a.px, cost_tile_h=a.tile_h, cost_kernel=a.kernel,
                                                         cost_workgroup_waves=a.waves))
for _ in range(a.launches):
    plan.cost_wta(L, R, a.cost, want_best=False)
torch.cuda.synchronize()
print(a.config, a.cost, "done")
