"""Analytic VALU instruction count of one match launch.

The bit-sliced kernel (csrc/sm_match_bs_kernel.h) runs one wave per workgroup --
or two that share their warm-up rows ("duo", a variant of its own here) -- and
every wave executes the same straight-line code per row: a fixed set-up
(staging, lane roles, the N warm-up rows of its tile) and then one loop
iteration per output row (sliding update of all sums, arg-max over the lane's
shifts, merge across the shift lanes, planes -> integers).  So

    wave-instructions(launch) = waves * A  +  B * sum over waves of rows_out
                              = waves_per_workgroup * tiles_x * tiles_y * pairs * A + B * tiles_x * H * pairs

with two coefficients per kernel variant.  They are FITTED to SQ_INSTS_VALU of
separate `rocprofv3 --pmc` passes at several tile heights
(tools/fit_valu_model.py on the GPU box -> valu_counts.json next to this file;
the raw counter values are kept under profiles/); the residual of the fit is
stored with them.  bench.py divides this count by the measured launch time to
get the achieved VALU issue rate of its roofline object.
"""
from __future__ import annotations

import json
from pathlib import Path

COUNTS = Path(__file__).resolve().parent / "valu_counts.json"


def variant_key(geom: dict, num_shifts: int, border: int, want_best: bool) -> str:
    fulld = geom["shift_lanes"] * geom["shifts_per_lane"] == num_shifts
    duo = ":duo" if geom["kernel"] == 4 and geom.get("waves_per_workgroup", 1) == 2 else ""
    # the build capped at two waves per SIMD is code of its own (the register cap can move spills)
    cap2 = ":cap2" if geom["kernel"] == 4 and geom.get("two_wave_variant") else ""
    # the lanes of a word merged through LDS every four rows: other code in the row loop
    xm = ":xm" if geom["kernel"] == 4 and geom.get("lane_merge_lds") else ""
    return (f"k{geom['kernel']}:n{geom['window']}:ds{geom['shifts_per_lane']}:nl{geom['shift_lanes']}:"
            f"{'ghost' if border else 'toroidal'}:fulld{int(fulld)}:best{int(bool(want_best))}{duo}{cap2}{xm}")


def waves_and_rows(geom: dict, height: int, pairs: int):
    """(waves of the launch, sum over waves of the output rows each produces)."""
    waves_per_wg = geom.get("waves_per_workgroup") or max(1, geom["threads"] // 64)
    waves = geom["tiles_x"] * geom["tiles_y"] * pairs * waves_per_wg
    # bit-sliced kernel: a wave owns rows of its own; the popcount kernels' waves share a tile's rows
    rows = geom["tiles_x"] * height * pairs * (1 if geom["kernel"] == 4 else waves_per_wg)
    return waves, rows


def _setup_text(geom: dict) -> str:
    n = geom["window"]
    if geom["kernel"] == 4 and geom.get("waves_per_workgroup", 1) == 2:
        return (f"set-up + {n // 2 + 1} warm-up rows + swap of the {n - 1} shared rows' sums with the "
                f"workgroup's other wave")
    return f"set-up + {n} warm-up rows"


def match_launch(geom: dict, width: int, height: int, num_shifts: int, border: int,
                 pairs: int, want_best: bool = False):
    if not COUNTS.exists():
        return None
    table = json.loads(COUNTS.read_text())
    key = variant_key(geom, num_shifts, border, want_best)
    c = table.get("variants", {}).get(key)
    if not c:
        return None
    waves, rows = waves_and_rows(geom, height, pairs)
    total = c["per_wave"] * waves + c["per_wave_row"] * rows
    return {
        "wave_instructions": int(round(total)),
        "model": f"{waves} waves x {c['per_wave']:.0f} ({_setup_text(geom)}) + "
                 f"{rows} wave-rows x {c['per_wave_row']:.0f}",
        "source": f"{table.get('source', 'valu_counts.json')}; fit residual <= "
                  f"{c.get('max_rel_residual', 0) * 100:.2f} % over tile heights "
                  f"{[p[0] for p in c.get('fit_points', [])]}",
        "variant": key,
    }
