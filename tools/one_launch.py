#!/usr/bin/env python3
"""A few match launches of one configuration, for rocprofv3 passes.

    python3 tools/one_launch.py C3 1 [--best] [--launches 3] [--meta out.json]

Tile height etc. follow the plan (SM_TILE_H / SM_DS in this tool's environment override,
as for every plan).  Writes the plan geometry + the model's variant key to --meta.
"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from stereomatching_amd import pipeline, valu_model  # noqa: E402
from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("pairs", type=int, nargs="?", default=1)
ap.add_argument("--best", action="store_true")
ap.add_argument("--launches", type=int, default=3)
ap.add_argument("--meta")
a = ap.parse_args()

w, h, d, sw, mode = CONFIGS[a.config]
from tools._options import from_env  # noqa: E402
plan = pipeline.StereoPlan(w, h, d, sw, mode, max_pairs=a.pairs, options=from_env() or None)
ls, rs = zip(*[make_pair(w, h, d, seed=j) for j in range(a.pairs)])
left = torch.from_numpy(np.stack(ls)).cuda()
right = torch.from_numpy(np.stack(rs)).cuda()
plan.find_all_edges(left, right, want_edges=False)
web = best = None
for _ in range(a.launches):
    web, best = plan.match_wta(a.pairs, want_best=a.best, web=web, best=best)
torch.cuda.synchronize()
if a.meta:
    g = plan.geometry()
    waves, rows = valu_model.waves_and_rows(g, h, a.pairs)
    Path(a.meta).write_text(json.dumps({
        "config": a.config, "pairs": a.pairs, "best": a.best, "geometry": g, "waves": waves,
        "wave_rows": rows, "variant": valu_model.variant_key(g, d, plan.border, a.best),
        "describe": plan.describe()}) + "\n")
print(plan.describe())
