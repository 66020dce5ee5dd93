#!/usr/bin/env python3
"""Condense raw rocprofv3 output (tools/collect_profiles.sh) into the small
files kept under profiles/<tag>/ and refresh profiles/hbm_traffic.json.

    python tools/summarise_profiles.py gpurun_out/prof_r01 profiles/r01 [C3:1]
"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

src, dst = Path(sys.argv[1]), Path(sys.argv[2])
key = sys.argv[3] if len(sys.argv) > 3 else "C3:1"
dst.mkdir(parents=True, exist_ok=True)

def newest(pattern):
    fs = sorted(glob.glob(pattern), key=lambda f: Path(f).stat().st_mtime)
    return fs[-1:] if fs else []


# rocprofv3 --stats summary as written, plus the same statistics recomputed from the kernel
# trace WITHOUT the plan's set-up launch of the match kernel (one workgroup, ~2 us: it would
# pull the average down)
for f in newest(str(src / "kt" / "*" / "*_kernel_stats.csv")):
    shutil.copy(f, dst / "kernel_stats_rocprof.csv")
for f in newest(str(src / "kt" / "*" / "*_kernel_trace.csv")):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        gs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        by[r["Kernel_Name"]].append((gs, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    tot = sum(d for v in by.values() for _, d in v)
    with open(dst / "kernel_stats.csv", "w") as out:
        out.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","GridSize"\n')
        for name, v in sorted(by.items(), key=lambda kv: -sum(d for _, d in kv[1])):
            g = max(gs for gs, _ in v)
            ds = [d for gs, d in v if gs == g]           # full-size dispatches only
            out.write(f'"{name}",{len(ds)},{sum(ds)},{sum(ds) / len(ds):.1f},{100 * sum(ds) / tot:.2f},{min(ds)},{max(ds)},{g}\n')
for name in ("bench", "bench_driver_flags", "bench_nograph", "bench_nograph_driver_flags", "bench_graph", "bench_traced", "bench_after"):
    f = src / f"{name}.json"
    if f.exists():
        bench = [l for l in f.read_text().splitlines() if l.startswith("{")]
        if bench:
            (dst / f"{name}.json").write_text(bench[-1] + "\n")

counters = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(str(src / "pmc_*"))):
    for f in newest(str(Path(d) / "*" / "*_counter_collection.csv")):
        rows = list(csv.DictReader(open(f)))
        biggest = collections.defaultdict(int)
        for r in rows:
            biggest[r["Kernel_Name"]] = max(biggest[r["Kernel_Name"]], int(r["Grid_Size"]))
        for r in rows:
            if int(r["Grid_Size"]) != biggest[r["Kernel_Name"]]:
                continue                                  # the plan's one-workgroup set-up launch
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            counters[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in counters.items()
           if not k.startswith("__amd")}
(dst / "pmc_summary.json").write_text(json.dumps(summary, indent=1, sort_keys=True) + "\n")

# HBM traffic of the dominant kernel, per launch.  rocprofv3 reports FETCH_SIZE and
# WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B, so the
# read side is doubled (MI355X_MICROARCH.md, HBM section).  WRITE_SIZE is exact.
traffic_file = dst.parent / "hbm_traffic.json"
traffic = json.loads(traffic_file.read_text()) if traffic_file.exists() else {}
# (the bench run also creates the cost-mode plans, whose one-workgroup set-up launches of OTHER
# match kernels are in the trace: the dominant kernel is the one with the most waves)
match_kernels = [k for k, cs in summary.items() if k.startswith("k_match") and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs]
match_kernels = sorted(match_kernels, key=lambda k: summary[k].get("SQ_WAVES", 0))[-1:]
for k, cs in summary.items():
    if k in match_kernels:
        traffic[key] = {
            "kernel": k,
            "fetch_size_kib_raw": cs["FETCH_SIZE"],
            "write_size_kib": cs["WRITE_SIZE"],
            "bytes_per_launch": int((2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024),
            "valu_wave_instructions_per_launch": cs.get("SQ_INSTS_VALU"),
            "waves_per_launch": cs.get("SQ_WAVES"),
            "note": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE half-count "
                    "correction); L2->fabric requests, Infinity-Cache hits included",
            "source": str(dst),
        }
traffic_file.write_text(json.dumps(traffic, indent=1, sort_keys=True) + "\n")
print(json.dumps(summary, indent=1, sort_keys=True)[:3000])
