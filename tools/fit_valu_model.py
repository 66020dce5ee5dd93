#!/usr/bin/env python3
"""Fit the two coefficients of stereomatching_amd/valu_model.py per kernel variant to
the SQ_INSTS_VALU values collected by tools/fit_valu_model.sh.

    python tools/fit_valu_model.py gpurun_out/valu_fit profiles/r02/valu_fit.json

Writes the raw points to the given profile file and the coefficients to
stereomatching_amd/valu_counts.json.
"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

import numpy as np

src, prof = Path(sys.argv[1]), Path(sys.argv[2])
ROOT = Path(__file__).resolve().parent.parent
points = collections.defaultdict(list)
for meta_file in sorted(src.glob("*/meta.json")):
    meta = json.loads(meta_file.read_text())
    vals, waves = [], []
    # (gpurun MERGES a call's output into the local gpurun_out/: files of an earlier round's passes may
    # lie next to this one's -- only the newest counter file of a case counts)
    files = sorted(glob.glob(str(meta_file.parent / "prof" / "*" / "*_counter_collection.csv")),
                   key=lambda f: Path(f).stat().st_mtime)
    for f in files[-1:]:
        rows = [r for r in csv.DictReader(open(f)) if "k_match" in r["Kernel_Name"]]
        full = max((int(r["Grid_Size"]) for r in rows), default=0)
        for r in rows:
            if int(r["Grid_Size"]) != full:
                continue                    # the plan's one-workgroup set-up launch
            if r["Counter_Name"] == "SQ_INSTS_VALU":
                vals.append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "SQ_WAVES":
                waves.append(float(r["Counter_Value"]))
    if not vals:
        print("no counters in", meta_file.parent)
        continue
    assert max(vals) - min(vals) <= 1e-6 * max(vals), (meta_file, vals)   # deterministic code
    assert waves and abs(waves[0] - meta["waves"]) < 0.5, (meta_file, waves, meta["waves"])
    points[meta["variant"]].append({"tile_h": meta["geometry"]["tile_h"], "waves": meta["waves"],
                                    "wave_rows": meta["wave_rows"], "sq_insts_valu": vals[0],
                                    "config": f"{meta['config']}:{meta['pairs']}"})

variants = {}
for key, pts in points.items():
    a = np.array([[p["waves"], p["wave_rows"]] for p in pts], float)
    y = np.array([p["sq_insts_valu"] for p in pts], float)
    if len(pts) < 2:
        continue
    coef, *_ = np.linalg.lstsq(a, y, rcond=None)
    res = np.abs(a @ coef - y) / y
    variants[key] = {"per_wave": float(coef[0]), "per_wave_row": float(coef[1]),
                     "max_rel_residual": float(res.max()),
                     "fit_points": [[p["tile_h"], p["waves"], p["sq_insts_valu"]] for p in pts]}
    print(f"{key}: per wave {coef[0]:.1f}, per wave-row {coef[1]:.1f}, residual {res.max() * 100:.3f} %")

prof.parent.mkdir(parents=True, exist_ok=True)
prof.write_text(json.dumps({"points": points}, indent=1, sort_keys=True) + "\n")
out = ROOT / "stereomatching_amd" / "valu_counts.json"
# variants this set of passes did not touch keep their earlier fit (each entry says where it was fitted)
merged = json.loads(out.read_text()).get("variants", {}) if out.exists() else {}
for v in variants.values():
    v["fitted_in"] = str(prof)
merged.update(variants)
out.write_text(json.dumps({"source": f"SQ_INSTS_VALU of rocprofv3 --pmc passes ({prof}; variants these passes did not "
                                     "touch keep their earlier fit, see fitted_in)",
                           "variants": merged}, indent=1, sort_keys=True) + "\n")
print("wrote", prof, "and", out)
