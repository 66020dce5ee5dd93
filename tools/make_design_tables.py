#!/usr/bin/env python3
"""Generate DESIGN.md section 6's result tables from the files under profiles/<tag>/ -- nothing in
them is typed by hand.

    python tools/make_design_tables.py [r05]

Reads (all written by tools/summarise_profiles.py from one tools/collect_profiles.sh call, i.e. ONE box):
bench.json, bench_driver_flags.json, bench_graph.json, bench_traced.json, bench_after.json,
kernel_stats.csv, pmc_summary.json and, if present, bench_all_configs.jsonl (tools/bench_all_configs.sh).
Writes profiles/<tag>/RESULTS.md and replaces the block between the GENERATED:results markers of DESIGN.md.
"""
import csv
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
args_ = [a for a in sys.argv[1:] if not a.startswith("--")]
tag = args_[0] if args_ else "r05"
P = ROOT / "profiles" / tag


def load(name):
    f = P / name
    if not f.exists():
        return None
    lines = [l for l in f.read_text().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


out = [f"Results of round {tag[1:].lstrip('0')}, one MI355X, resident inputs. Source: `profiles/{tag}/` "
       f"(one `tools/collect_profiles.sh {tag}` call = one box; boxes of the pool differ by up to ≈ 10 % in absolute time)."]

# ---- bench lines of the collection call
rows = [("default run (`bench.json`: 200 steps, replayed from HIP graphs)", "bench.json"),
        ("the driver's flags (`bench_driver_flags.json`: `--steps 20 --warmup 5`)", "bench_driver_flags.json"),
        ("`--no-graph`, 200 steps launched from the host (`bench_nograph.json`)", "bench_nograph.json"),
        ("`--no-graph --steps 20 --warmup 5` (`bench_nograph_driver_flags.json`)", "bench_nograph_driver_flags.json"),
        ("`--steps 2000`, untraced (`bench_graph.json`)", "bench_graph.json"),
        ("`--steps 2000` UNDER `rocprofv3 --kernel-trace` (`bench_traced.json`)", "bench_traced.json"),
        ("default flags again, after the PMC passes (`bench_after.json`)", "bench_after.json")]
tab = ["| `bench.py` run, C3 (4K pair, 128 shifts, 9×9, toroidal) | value (M Mpx-disp/s) | ms per step | `roofline.kernel_ms` | VALU instr / launch | `frac` | of sustained | `verified` |",
       "|---|---|---|---|---|---|---|---|"]
lines = {}
for label, name in rows:
    d = load(name)
    if not d:
        continue
    lines[name] = d
    r = d["roofline"]
    tab.append(f"| {label} | {d['value'] / 1e6:.2f} | {d['ms_per_step']:.4f} | {r['kernel_ms']:.4f} | "
               f"{r.get('valu_wave_instructions_per_launch', 0) / 1e6:.2f} M | {r.get('frac')} | {r.get('frac_of_sustained')} | {d.get('verified')} |")
if len(tab) > 2:
    out += ["", *tab]

# ---- kernel trace of the traced run against the untraced kernel time
ks = P / "kernel_stats.csv"
if ks.exists():
    krows = list(csv.DictReader(open(ks)))
    t2 = ["| kernel (trace of the 2000-step run, full-size dispatches) | calls | average µs | min µs | max µs | share |", "|---|---|---|---|---|---|"]
    match_avg = None
    for r in krows:
        name = r["Name"].split("(")[0].replace("void ", "")
        if not (name.startswith("k_match") or name.startswith("k_edges")):
            continue
        if int(r["Calls"]) < 10:
            continue                    # one-off launches (set-up of the cost-mode plans)
        t2.append(f"| `{name[:60]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {int(r['MinNs']) / 1e3:.2f} | "
                  f"{int(r['MaxNs']) / 1e3:.2f} | {r['Percentage']} % |")
        if name.startswith("k_match_bs") and match_avg is None:
            match_avg = float(r["AverageNs"]) / 1e3
    out += ["", *t2]
    ref = lines.get("bench_graph.json") or lines.get("bench.json")
    if match_avg and ref:
        k = ref["roofline"]["kernel_ms"] * 1e3
        tr = lines.get("bench_traced.json")
        out += ["", f"Trace against line: `k_match_bs` averages **{match_avg:.2f} µs** in the trace; the untraced run of the same "
                    f"call measured {k:.2f} µs ({(match_avg / k - 1) * 100:+.1f} %)" +
                    (f", the traced process's own line {tr['roofline']['kernel_ms'] * 1e3:.2f} µs at {tr['ms_per_step']:.4f} ms per step "
                     f"(untraced: {ref['ms_per_step']:.4f})" if tr else "") + "."]

# ---- PMC summary of the match and edge kernels
pm = P / "pmc_summary.json"
if pm.exists():
    s = json.loads(pm.read_text())
    t3 = ["| counter (per launch, separate `--pmc` passes) | `k_match_bs` | `k_edges_ext4` |", "|---|---|---|"]
    km = next((k for k in s if k.startswith("k_match_bs")), None)
    ke = next((k for k in s if k.startswith("k_edges_ext4")), None)
    if km:
        names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                 "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "FETCH_SIZE", "WRITE_SIZE"]
        for n in names:
            a = s[km].get(n)
            b = s.get(ke, {}).get(n) if ke else None
            if a is None and b is None:
                continue
            unit = " KiB" if n.endswith("_SIZE") else ""
            t3.append(f"| {n} | {a:,.0f}{unit} | " + (f"{b:,.0f}{unit} |" if b is not None else "— |"))
        out += ["", *t3]
        m = s[km]
        if "SQ_ACTIVE_INST_VALU" in m and "SQ_WAVE_CYCLES" in m:
            extra = (f"`k_match_bs`: VALU active {m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES'] * 100:.1f} % of the wave-cycles, "
                     f"waiting {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES'] * 100:.1f} %")
            if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
                tr = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
                extra += f"; traffic (2·FETCH_SIZE + WRITE_SIZE)·1024 = {tr / 1e6:.2f} MB per launch"
            out += ["", extra + "."]

# ---- the other configurations
allc = P / "bench_all_configs.jsonl"
if allc.exists():
    t4 = ["| configuration (`tools/bench_all_configs.sh`, one device) | step | match kernel | value (M Mpx-disp/s) |", "|---|---|---|---|"]
    for l in allc.read_text().splitlines():
        if not l.startswith("{"):
            continue
        d = json.loads(l)
        w = d['config']['workload'].split(';')[0] + (" — consecutive steps overlapped (`--overlap`: `sm_run_after`)" if d['config'].get('pipelined') else "")
        t4.append(f"| {w} | {d['ms_per_step']:.4f} ms | {d['roofline']['kernel_ms']:.4f} ms | {d['value'] / 1e6:.2f} |")
    out += ["", *t4]

# ---- extras of the default line
d = lines.get("bench.json")
if d:
    ex = []
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        ex.append(f"CPU baseline ({c['kind']}, {c['cores']} core): {c['value']} {c['unit']}" +
                  (f"; {c['all_cores']['value']} on {c['all_cores']['cores']} threads (usable cores: {c['all_cores'].get('usable_cores')}; "
                   f"rate by thread count {c['all_cores'].get('rate_by_threads')})" if "all_cores" in c else ""))
    if "verified" in d:
        v = d.get("verification", {})
        ex.append(f"`verified` {d['verified']} (maps of the timed steps equal host-launched runs: {v.get('maps_equal_host_launched_runs')}; "
                  f"band equals the CPU oracle: {v.get('band_equals_cpu_oracle')}; extras: {v.get('extras_verified')})")
    if "host_launched" in d:
        ex.append(f"plain host-launched `sm_run`: {d['host_launched']['ms_per_step']} ms per step")
    if "overlapped" in d and "value" in d["overlapped"]:
        o = d["overlapped"]
        ex.append(f"`overlapped` (`sm_run_after`, never `value`): {o['ms_per_step']} ms per step, {o['value'] / 1e6:.2f} M Mpx-disp/s, verified {o['verified']}")
    if "c2" in d:
        o = d["c2"]
        ex.append(f"`c2` (1080p pair, 64 shifts, 7×7): {o['ms_per_step']} ms per step, {o['value'] / 1e6:.2f} M Mpx-disp/s in stream order; "
                  f"{o['overlapped']['ms_per_step']} ms, {o['overlapped']['value'] / 1e6:.2f} M overlapped; host-launched {o['host_launched_ms_per_step']} ms; verified {o['verified']}")
    for k in ("sad", "ssd", "ssd_c3"):
        if k in d:
            o = d[k]
            ex.append(f"`{k}` ({o['workload'].split(':')[0]}, parity unpinned, verified {o.get('verified')}): {o['ms_per_launch']} ms per launch, "
                      f"{o['value'] / 1e6:.2f} M Mpx-disp/s" +
                      (f", VALU frac {o['roofline']['frac']}, {o['roofline']['lane_instructions_per_pixel_shift']} lane-instructions per pixel-shift"
                       if "roofline" in o else ""))
    if "e2e" in d and isinstance(d["e2e"], dict) and "error" not in d["e2e"]:
        e = d["e2e"]
        ex.append("`e2e` (PCIe-inclusive, never `value`): " + ", ".join(f"{k} {v}" for k, v in e.items()
                                                                          if isinstance(v, (int, float)) and "ms" in k) +
                  ", ".join(f"{k} map {v['ms_per_pair']} ms per 4K pair ({v['pcie_GBps']} GB/s over PCIe)" for k, v in e.items()
                            if isinstance(v, dict) and "ms_per_pair" in v))
    if ex:
        out += ["", "Extras of the default line: " + "; ".join(ex) + "."]

text = "\n".join(out) + "\n"
design = ROOT / "DESIGN.md"
src = design.read_text()
pat = re.compile(r"(<!-- GENERATED:results BEGIN[^\n]*-->\n).*?(<!-- GENERATED:results END -->)", re.S)
assert pat.search(src), "DESIGN.md has no GENERATED:results block"
if "--check" in sys.argv:
    # (tests/test_docs_cpu.py: the block in DESIGN.md and profiles/<tag>/RESULTS.md are what the files say)
    cur = pat.search(src).group(0).split("-->\n", 1)[1].rsplit("<!-- GENERATED:results END -->", 1)[0]
    ok = cur == text and (P / "RESULTS.md").read_text() == text
    print("DESIGN.md's generated block is up to date" if ok else "DESIGN.md's generated block is STALE: run tools/make_design_tables.py")
    sys.exit(0 if ok else 1)
(P / "RESULTS.md").write_text(text)
design.write_text(pat.sub(lambda m: m.group(1) + text + m.group(2), src))
print(text)
