"""The numbers in DESIGN.md section 6 are generated from profiles/r05, not typed: the generated block must be
what tools/make_design_tables.py produces from the committed profile files, and the documents must not
contradict the files they cite (round-3 review: DESIGN quoted numbers that were not in the files)."""
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_design_tables_are_generated_from_the_profiles():
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "make_design_tables.py"), "r05", "--check"],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr


def test_profile_files_cited_in_the_design_exist():
    text = (ROOT / "DESIGN.md").read_text()
    cited = set(re.findall(r"`(profiles/r0[1-5]/[A-Za-z0-9_.{},*-]+)`", text))
    assert cited, "DESIGN.md cites no profile files?"
    missing = []
    for c in cited:
        if "{" in c or "*" in c:
            continue
        if not (ROOT / c).exists():
            missing.append(c)
    assert not missing, missing


def test_trace_and_line_agree():
    """the committed kernel trace (taken from the graph-replayed run) and the untraced kernel time of the same
    collection call agree within 5 % (round-3 review, item 1)"""
    import csv
    rows = list(csv.DictReader(open(ROOT / "profiles" / "r05" / "kernel_stats.csv")))
    match = [r for r in rows if "k_match_bs" in r["Name"] and int(r["Calls"]) > 100]
    assert match
    trace_us = float(match[0]["AverageNs"]) / 1e3
    line = json.loads((ROOT / "profiles" / "r05" / "bench_graph.json").read_text())
    untraced_us = line["roofline"]["kernel_ms"] * 1e3
    assert abs(trace_us / untraced_us - 1) < 0.05, (trace_us, untraced_us)
    # ... and the roofline's traffic is the compulsory bytes (no wasted re-reads or double writes)
    t = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text())["C3:1"]
    assert t["source"].endswith("r05") and 35.0e6 < t["bytes_per_launch"] < 37.0e6
    # ... and every line of the collection says that what it timed was checked
    for name in ("bench", "bench_driver_flags", "bench_graph", "bench_traced", "bench_nograph", "bench_after"):
        d = json.loads((ROOT / "profiles" / "r05" / f"{name}.json").read_text())
        assert d["verified"] is True and d["verification"]["maps_equal_host_launched_runs"] is True, name


def test_cost_profile_names_kernels_that_exist():
    """profiles/cost_valu.json (what bench.py prices its `sad` / `ssd` objects with) names kernels the sources
    define, and DESIGN.md's abbreviated profile citations (`ab_*.txt` without a directory) point at files under
    profiles/r03 .. r05/"""
    src = "".join(p.read_text() for p in (ROOT / "stereomatching_amd" / "csrc").glob("*.hip"))
    for key, c in json.loads((ROOT / "profiles" / "cost_valu.json").read_text()).items():
        for name in re.findall(r"k_[a-z_0-9]+", c["kernel"]):
            assert f"void {name}(" in src, (key, name)
        assert (ROOT / c["source"].split(" ")[0]).exists(), c["source"]
    text = (ROOT / "DESIGN.md").read_text()
    missing = [f for f in set(re.findall(r"`(?:…/)?((?:ab|pmc|ds_choice|write_size|cost)_[A-Za-z0-9_]+\.(?:txt|json))`", text))
               if not any((ROOT / "profiles" / r / f).exists() for r in ("r03", "r04", "r05"))]
    assert not missing, missing


def test_integration_lists_the_translation_units_that_are_built():
    """INTEGRATION.md's account of the library (how many translation units, which) is what build.py and the
    Makefile compile (round-4 review: the text said ten, the build had thirteen)"""
    sys.path.insert(0, str(ROOT))
    from stereomatching_amd import build
    text = (ROOT / "INTEGRATION.md").read_text()
    para = text[text.index("The library itself is"):]
    para = para[:para.index("each compiled with")]
    named = set(re.findall(r"`(sm_[a-z_0-9]+\.hip)`", para))
    assert named == set(build.SOURCES), (sorted(named ^ set(build.SOURCES)))
    words = {10: "ten", 11: "eleven", 12: "twelve", 13: "thirteen", 14: "fourteen", 15: "fifteen", 16: "sixteen"}
    assert f"is {words[len(build.SOURCES)]} translation units" in para
    mk = (ROOT / "Makefile").read_text()
    kernels = re.search(r"^KERNELS := (.*)$", mk, re.M).group(1).split()
    assert {k + ".hip" for k in kernels} == set(build.SOURCES)
