"""CPU-side checks of the boundary: the C-ABI library loads, exports every
symbol include/stereo_hip.h declares, and rejects bad arguments before it
touches a device (no compute calls: there is no GPU in the CPU suite)."""
import ctypes as C
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from stereomatching_amd import capi
    syms = capi.declared_symbols()
    assert len(syms) >= 20 and "sm_match_wta" in syms and "sm_plan_create" in syms
    for s in syms:
        assert hasattr(capi.lib, s), s
        assert s in capi._SIGNATURES, f"{s} has no ctypes signature"
    # and nothing is bound that the header does not declare
    assert set(capi._SIGNATURES) == set(syms)


def test_exported_symbols_are_plain_c():
    out = subprocess.check_output(["nm", "-D", "--defined-only",
                                   str(ROOT / "stereomatching_amd" / "libstereo_hip.so")], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    from stereomatching_amd import capi
    assert set(capi.declared_symbols()) <= exported


def test_argument_validation_precedes_device_use():
    from stereomatching_amd import capi
    h = C.c_void_p(0)
    lib = capi.lib
    # messages for the checks the reference's main() performs are the reference's own
    # (/root/reference/src/stereo.c:382-385)
    assert lib.sm_plan_create(0, 64, 48, 30, 65, 0, 1, C.byref(h)) == capi.SM_ERR_ARG
    assert lib.sm_last_error() == b"error: square width must not be higher than image width/height"
    assert lib.sm_plan_create(0, 64, 48, 30, 49, 0, 1, C.byref(h)) == capi.SM_ERR_ARG
    assert lib.sm_plan_create(0, 0, 48, 30, 5, 0, 1, C.byref(h)) == capi.SM_ERR_ARG
    assert lib.sm_plan_create(0, 64, 48, 0, 5, 0, 1, C.byref(h)) == capi.SM_ERR_ARG
    assert lib.sm_plan_create(0, 64, 48, 70000, 5, 0, 1, C.byref(h)) == capi.SM_ERR_ARG
    assert lib.sm_plan_create(0, 64, 48, 30, -1, 0, 1, C.byref(h)) == capi.SM_ERR_ARG
    assert lib.sm_plan_create(0, 64, 48, 30, 5, 2, 1, C.byref(h)) == capi.SM_ERR_ARG
    assert lib.sm_plan_create(0, 64, 48, 30, 5, 0, 0, C.byref(h)) == capi.SM_ERR_ARG
    assert lib.sm_plan_create(0, 64, 48, 30, 5, 0, 1, None) == capi.SM_ERR_ARG
    assert not h.value
    assert lib.sm_match_wta(None, 1, None, None, None) == capi.SM_ERR_ARG
    assert lib.sm_find_edges(None, None, None, 0.15, 1, None, None, None) == capi.SM_ERR_ARG
    assert b"plan is NULL" in lib.sm_last_error()
    assert lib.sm_plan_prepare_threshold(None, 0.15, None) == capi.SM_ERR_ARG
    assert lib.sm_plan_time_stride(None, 8) == capi.SM_ERR_ARG
    assert lib.sm_plan_time_kernels(None, 4) == capi.SM_ERR_ARG
    assert lib.sm_plan_set_pipelined(None, 1) == capi.SM_ERR_ARG
    assert lib.sm_plan_describe(None) == b""
    lib.sm_plan_destroy(None)  # no-op


def test_plan_options_struct_and_argument_checks():
    """sm_plan_create_ex: the ctypes mirror of sm_plan_options has the header's fields in the header's
    order; a struct_size that is not one of a sm_plan_options is refused before any device work, and the
    library reads no environment variable (variants are chosen through this struct)."""
    import re
    from stereomatching_amd import capi
    text = re.sub(r"/\*.*?\*/", "", capi.HEADER.read_text(), flags=re.S)
    body = re.search(r"typedef struct sm_plan_options \{(.*?)\} sm_plan_options;", text, re.S).group(1)
    fields = re.findall(r"(?:int|unsigned)\s+(\w+)\s*;", body)
    assert fields == [n for n, _ in capi.PlanOptions._fields_]
    assert C.sizeof(capi.PlanOptions) == 4 * len(fields)
    lib = capi.lib
    h = C.c_void_p(0)
    bad = capi.PlanOptions.make(tile_h=4)
    bad.struct_size = 3
    assert lib.sm_plan_create_ex(0, 64, 48, 30, 5, 0, 1, C.byref(bad), C.byref(h)) == capi.SM_ERR_ARG
    assert b"struct_size" in lib.sm_last_error()
    bad.struct_size = -8
    assert lib.sm_plan_create_ex(0, 64, 48, 30, 5, 0, 1, C.byref(bad), C.byref(h)) == capi.SM_ERR_ARG
    assert b"struct_size" in lib.sm_last_error()
    # 0 ("no field set") and a LONGER struct (a caller built against a newer header: the known prefix is
    # taken) pass the argument checks -- without a GPU the call then fails on the device, not on the struct
    for size in (0, C.sizeof(capi.PlanOptions) + 64):
        bad.struct_size = size
        rc = lib.sm_plan_create_ex(0, 64, 48, 30, 5, 0, 1, C.byref(bad), C.byref(h))
        assert rc == capi.SM_OK or b"struct_size" not in lib.sm_last_error()
        if rc == capi.SM_OK:
            lib.sm_plan_destroy(h)
            h = C.c_void_p(0)
    assert lib.sm_plan_create_ex(0, 64, 48, 30, 5, 0, 1, None, None) == capi.SM_ERR_ARG
    assert not h.value
    csrc = ROOT / "stereomatching_amd" / "csrc"
    for f in list(csrc.glob("*.hip")) + list(csrc.glob("*.h")):
        assert "getenv" not in f.read_text(), f
    # the tuning tools load libraries by hand and carry their own mirror of the struct: same fields, and
    # every short name they accept maps onto one of them
    import sys
    sys.path.insert(0, str(ROOT))
    from tools import _options
    assert [n for n, _ in _options.PlanOptions._fields_] == fields
    for name, to_fields in _options.NAMES.items():
        probe = "popcount" if name == "SM_KERNEL" else ("0x3" if name == "SM_PATTERN" else "2")
        assert set(to_fields(probe)) <= set(fields), name


def test_product_package_never_touches_the_oracle():
    # the product path must not import, link or execute anything under oracle/
    pkg = ROOT / "stereomatching_amd"
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")) + \
            list(pkg.rglob("*.c")):
        text = p.read_text()
        if p.name == "build.py":
            continue  # builds the checker, does not use it
        assert "liboracle" not in text and "stereo_oracle" not in text and \
            "tests.oracle" not in text and "from tests" not in text, p


def test_synth_and_pgm_roundtrip(tmp_path):
    import numpy as np
    from stereomatching_amd.synth import make_pair, read_pgm, write_pgm
    a, b = make_pair(37, 21, 16, seed=3)
    a2, b2 = make_pair(37, 21, 16, seed=3)
    assert np.array_equal(a, a2) and np.array_equal(b, b2) and a.dtype == np.uint8
    write_pgm(tmp_path / "x.pgm", a)
    assert np.array_equal(read_pgm(tmp_path / "x.pgm"), a)


def test_valu_model_reproduces_the_profiled_count():
    """The analytic VALU model behind bench.py's roofline: for the headline geometry (C3: 15 x
    68 two-wave workgroups of 2 x 16 rows) it must land on the SQ_INSTS_VALU the r02 profile
    holds, and every fitted variant must have kept its residual small."""
    import json
    from pathlib import Path

    from stereomatching_amd import valu_model
    root = Path(__file__).resolve().parent.parent
    geom = dict(kernel=4, window=9, shifts_per_lane=16, shift_lanes=8, threads=128, tile_w=256, tile_h=32,
                tiles_x=15, tiles_y=68, waves_per_workgroup=2)
    m = valu_model.match_launch(geom, 3840, 2160, 128, 0, 1, want_best=False)
    assert m is not None and m["variant"] == "k4:n9:ds16:nl8:toroidal:fulld1:best0:duo"
    # the one-wave shape of the same window (SM_DUO=0) is a variant of its own
    one = valu_model.match_launch(dict(geom, threads=64, tile_h=16, tiles_y=135, waves_per_workgroup=1),
                                  3840, 2160, 128, 0, 1)
    assert one["variant"] == "k4:n9:ds16:nl8:toroidal:fulld1:best0"
    assert one["wave_instructions"] > 1.05 * m["wave_instructions"]      # 9 warm-up rows per wave, not 5
    prof = json.loads((root / "profiles" / "r02" / "pmc_summary.json").read_text())
    counted = [v["SQ_INSTS_VALU"] for k, v in prof.items() if k.startswith("k_match_bs<9, 16")][0]
    assert abs(m["wave_instructions"] - counted) / counted < 0.005
    table = json.loads(valu_model.COUNTS.read_text())
    assert all(v["max_rel_residual"] < 0.005 for v in table["variants"].values())
    # a variant nobody fitted gives no number rather than a wrong one
    assert valu_model.match_launch(dict(geom, window=13), 3840, 2160, 128, 0, 1) is None
