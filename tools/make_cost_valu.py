#!/usr/bin/env python3
"""profiles/cost_valu.json (what bench.py's `sad` / `ssd` objects price their launches with) from the
summaries tools/cost_pmc.sh left under profiles/<round>/cost_<cost>_<cfg>.json.
    python tools/make_cost_valu.py profiles/r03"""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from stereomatching_amd.synth import CONFIGS  # noqa: E402

src = Path(sys.argv[1])
out = {}
for f in sorted(src.glob("cost_*_C*.json")):
    m = re.fullmatch(r"cost_(sad|ssd)_(C\d)\.json", f.name)
    if not m:
        continue                # (A/B records such as cost_sad_C3_round4_kernel.json)
    cost, cfg = m.groups()
    d = json.loads(f.read_text())
    w, h, shifts, _, _ = CONFIGS[cfg]
    valu = sum(v for k, v in d["counters_per_launch"].items() if k.endswith("|SQ_INSTS_VALU"))
    active = sum(v for k, v in d["counters_per_launch"].items() if k.endswith("|SQ_ACTIVE_INST_VALU"))
    mfma = sum(v for k, v in d["counters_per_launch"].items() if k.endswith("|SQ_INSTS_VALU_MFMA_I8"))
    names = [re.sub(r"^void ", "", k["Name"]).split("(")[0] for k in d["kernel_stats"]]
    out[f"{cfg}:{cost}"] = {
        "kernel": names[0] + (f" + ghost strip {names[1].split('<')[0]}" if len(names) > 1 else ""),
        "valu_wave_instructions": int(round(valu)),
        "kernel_trace_avg_ns": {n: round(float(k["AverageNs"]), 2) for n, k in zip(names, d["kernel_stats"])},
        "lane_instructions_per_pixel_shift": round(valu * 64 / (float(w) * h * shifts), 2),
        "source": f"{f.relative_to(ROOT) if f.is_absolute() else f} (rocprofv3 --pmc SQ_INSTS_VALU, separate passes)",
    }
    if cost == "sad" and active > valu:
        # the quad-SAD instructions issue over four passes, every other VALU instruction over one (tools/ubench_sad.hip), and
        # SQ_ACTIVE_INST_VALU counts passes: (passes - instructions) / 3 of them are v_qsad / v_mqsad
        q = (active - valu) / 3.0
        out[f"{cfg}:{cost}"]["qsad_wave_instructions"] = int(round(q))
        out[f"{cfg}:{cost}"]["v_qsad_lane_instructions_per_pixel_shift"] = round(q * 64 / (float(w) * h * shifts), 3)
        out[f"{cfg}:{cost}"]["valu_issue_passes"] = int(round(active))
    if mfma:            # v_mfma_i32_32x32x32_i8: 2 x 32 x 32 x 32 operations each
        out[f"{cfg}:{cost}"]["mfma_i8_instructions"] = int(round(mfma))
order = ["C3:sad", "C5:sad", "C3:ssd", "C5:ssd"]
out = {k: out[k] for k in order if k in out} | {k: v for k, v in out.items() if k not in order}
(ROOT / "profiles" / "cost_valu.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
