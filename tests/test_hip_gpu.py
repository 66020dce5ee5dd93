"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle (tests/oracle.py) and the golden vectors of the compiled reference.
Everything here is integer / byte work: the bar is bit-exact equality."""
import os

import numpy as np
import pytest
import torch

from stereomatching_amd.synth import CONFIGS, make_pair
from tests import oracle
from tests.conftest import big_reference_cases, golden_cases, load_big_reference, load_golden, sha256_of

pytestmark = pytest.mark.gpu
PLANE_SHIFTS = (0, 1, 7, 29)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def rand_edges(w, h, seed, density=0.5):
    rng = np.random.default_rng(seed)
    return ((rng.random((h, w)) < density).astype(np.uint8),
            (rng.random((h, w)) < density).astype(np.uint8))


def hip_hot_path(hip, le, re, d, sw, mode, pairs=1, options=None):
    h, w = le.shape[-2:]
    plan = hip.StereoPlan(w, h, d, sw, mode, max_pairs=pairs, options=options)
    plan.load_edges(dev(le), dev(re))
    web, best = plan.match_wta(pairs, want_best=True)
    torch.cuda.synchronize()
    desc = plan.describe()
    plan.close()
    return host(best), host(web), desc


# ---------------------------------------------------------------------------
# golden vectors of the compiled reference (D = 30)
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("name", golden_cases())
def test_pipeline_matches_reference_golden(hip, name):
    z, p = load_golden(name)
    h, w = z["left"].shape
    plan = hip.StereoPlan(w, h, 30, p["square_width"], p["mode"])
    res = plan.algorithm(dev(z["left"]), dev(z["right"]),
                         hip.AlgorithmParams(p["threshold"], p["square_width"], p["times"], p["lines"]))
    for k in ("edges-1", "edges-2", "score_best-0", "web-1", "web-2", "output-0"):
        assert np.array_equal(host(res[k])[0], z[k]), (name, k, plan.describe())
    for d in PLANE_SHIFTS:
        m, sa, sc = plan.debug_planes(0, d)
        assert np.array_equal(host(m), z[f"matches-{d}"]), (name, d)
        assert np.array_equal(host(sa), z[f"score_all-{d}"]), (name, d)
        assert np.array_equal(host(sc), z[f"scores-{d}"]), (name, d)
    plan.close()


@pytest.mark.parametrize("name,mode", big_reference_cases())
def test_reference_large_test_pairs_hash_to_the_reference(hip, name, mode):
    """The reference's own 1080p and 4K test pairs (test/imgs/4-1920x1080, 5-3840x2160; test/time.sh:6-9)
    at its defaults (30 shifts, 21 x 21, 32 fill sweeps, 10 lines): every stage of the HIP pipeline
    hashes to what the COMPILED REFERENCE produced -- the 4K tiling pinned to the reference itself,
    not to the restatement."""
    left, right, d = load_big_reference(name, mode)
    p = d["params"]
    h, w = left.shape
    plan = hip.StereoPlan(w, h, p["num_shifts"], p["square_width"], mode)
    res = plan.algorithm(dev(left), dev(right),
                         hip.AlgorithmParams(p["threshold"], p["square_width"], p["times"], p["lines"]))
    for k in ("edges-1", "edges-2", "score_best-0", "web-1", "web-2", "output-0"):
        got = host(res[k])[0]
        assert str(got.dtype) == d["dtype"][k] and list(got.shape) == d["shape"], (k, got.dtype, got.shape)
        assert sha256_of(got) == d["sha256"][k], (name, mode, k, plan.describe())
    plan.close()


# ---------------------------------------------------------------------------
# hot path vs oracle: every kernel variant, ragged sizes, both borders
# ---------------------------------------------------------------------------

HOT_CASES = [
    # (w, h, D, S)          kernel / what it exercises
    (64, 48, 16, 5),        # A, one shift-lane
    (71, 53, 30, 5),        # A, D not a multiple of 16, ragged width
    (130, 70, 64, 7),       # A, 4 shift-lanes
    (300, 150, 128, 9),     # A, the headline geometry, 2 tiles wide
    (96, 64, 256, 9),       # A, 16 shift-lanes (DPP row_mirror)
    (80, 40, 500, 3),       # A, 32 shift-lanes (cross-row shuffle), D > W
    (64, 36, 1000, 1),      # A, 64 shift-lanes, 1x1 window
    (90, 61, 64, 11),       # B
    (77, 45, 30, 16),       # B, even S rounds up to 17 -> C
    (100, 50, 48, 15),      # B upper end
    (120, 80, 30, 21),      # C, the reference's default window
    (70, 64, 128, 25),      # C upper end
    (60, 50, 30, 27),       # generic (window too large for the tiled kernels)
    (50, 40, 1100, 5),      # generic (D > 1024)
    (9, 7, 16, 5),          # tiny
    (8, 5, 3, 4),           # tiny, D < 16
    (257, 129, 64, 7),      # one pixel past a tile in both directions
    (33, 300, 16, 9),       # tall and narrow
    (1024, 32, 64, 0),      # S = 0 -> 1x1 window
]


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("w,h,d,sw", HOT_CASES)
def test_hot_path_matches_oracle(hip, mode, w, h, d, sw):
    le, re = rand_edges(w, h, seed=w * 7 + h * 3 + d + sw)
    best, web, desc = hip_hot_path(hip, le, re, d, sw, mode)
    obest, oweb = oracle.hot_path(le, re, d, sw, mode)
    assert np.array_equal(web[0], oweb), desc
    assert np.array_equal(best[0], obest), desc


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("kind", ["zeros", "ones", "left_only", "sparse"])
def test_hot_path_degenerate_edges(hip, mode, kind):
    w, h, d, sw = 100, 60, 30, 9
    le = np.zeros((h, w), np.uint8)
    re = np.zeros((h, w), np.uint8)
    if kind == "ones":
        le[:] = 1; re[:] = 1
    elif kind == "left_only":       # no pixel ever matches in toroidal mode -> web = D, best = 0
        le[:] = 1
    elif kind == "sparse":
        le[::7, ::5] = 1; re[::7, 3::5] = 1
    best, web, desc = hip_hot_path(hip, le, re, d, sw, mode)
    obest, oweb = oracle.hot_path(le, re, d, sw, mode)
    assert np.array_equal(web[0], oweb), desc
    assert np.array_equal(best[0], obest), desc
    if kind == "left_only" and mode == "toroidal":
        assert (web == d).all() and (best == 0).all()


def test_batch_of_pairs_equals_single_runs(hip):
    w, h, d, sw, pairs = 140, 90, 64, 7, 5
    les, res = zip(*[rand_edges(w, h, seed=100 + i, density=0.3) for i in range(pairs)])
    le, re = np.stack(les), np.stack(res)
    best, web, _ = hip_hot_path(hip, le, re, d, sw, "toroidal", pairs=pairs)
    for i in range(pairs):
        ob, ow = oracle.hot_path(le[i], re[i], d, sw, "toroidal")
        assert np.array_equal(web[i], ow) and np.array_equal(best[i], ob), i


def test_web_only_output_and_errors(hip):
    w, h = 64, 40
    plan = hip.StereoPlan(w, h, 30, 5)
    with pytest.raises(hip.capi.StereoHipError, match="edges of only 0 are loaded"):
        plan.match_wta(1)
    le, re = rand_edges(w, h, 1)
    plan.load_edges(dev(le), dev(re))
    web, best = plan.match_wta(1, want_best=False)
    assert best is None
    assert np.array_equal(host(web)[0], oracle.hot_path(le, re, 30, 5)[1])
    with pytest.raises(hip.capi.StereoHipError, match="pairs 2 outside"):
        plan.match_wta(2)
    with pytest.raises(ValueError):
        plan.load_edges(dev(le[:10]), dev(re[:10]))
    with pytest.raises(hip.capi.StereoHipError, match="threshold must be between 0 and 1"):
        plan.find_all_edges(dev(le), dev(re), 1.5)
    plan.close()


# ---------------------------------------------------------------------------
# step 1: edges
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("thr", [0.0, 0.05, 0.15, 0.25, 0.5, 0.75, 1.0, 1.0 / 3.0, 0.1])
def test_edge_decision_exhaustive(hip, thr):
    """every pair of in-image side sums: device arithmetic == host arithmetic"""
    import ctypes as C
    tab = torch.empty((766, 766), dtype=torch.uint8, device="cuda")
    hip.capi.check(hip.capi.lib.sm_debug_edge_table(0, thr, C.c_void_p(tab.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.array_equal(host(tab), oracle.edge_table(thr))


def test_edge_threshold_tables_exhaustive(hip):
    """the integer lo/hi tables sm_find_edges decides with reproduce the exact double
    test for every pair of in-image sums, for round and for random thresholds"""
    import ctypes as C
    rng = np.random.default_rng(11)
    thresholds = [0.0, 1.0, 0.15, 0.5, 0.25, 0.75, 1.0 / 3.0, 2.0 / 3.0, 1e-9, 0.999999,
                  *rng.random(20).tolist()]
    plan = hip.StereoPlan(32, 32, 16, 5)
    tab = torch.empty((766, 766), dtype=torch.uint8, device="cuda")
    for thr in thresholds:
        bad = C.c_int(-1)
        hip.capi.check(hip.capi.lib.sm_debug_edge_table_fast(plan._h, thr, C.c_void_p(tab.data_ptr()),
                                                             C.byref(bad), None))
        assert bad.value == 0, thr
        assert np.array_equal(host(tab), oracle.edge_table(thr)), thr
    plan.close()


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("w,h,kind,thr", [
    (64, 48, "scene", 0.15), (301, 97, "scene", 0.15), (130, 77, "noise", 0.5),
    (40, 40, "noise", 0.0), (55, 33, "scene", 1.0), (3, 3, "noise", 0.15),
    (1, 9, "noise", 0.15), (9, 1, "noise", 0.15), (2, 2, "noise", 0.3), (512, 64, "constant", 0.15),
    # 4-pixel-per-lane kernel across wave and workgroup seams (neighbour pixels via DPP,
    # real loads only at lanes 0 / 63), with ties (threshold 0 sends pixels to the tables)
    (1284, 21, "noise", 0.15), (772, 13, "noise", 0.0), (2052, 9, "scene", 0.075), (4, 7, "noise", 0.2)])
def test_edges_match_oracle(hip, mode, w, h, kind, thr):
    left, right = make_pair(w, h, 16, seed=w + h, kind=kind)
    sw = min(5, w, h)
    plan = hip.StereoPlan(w, h, 16, sw, mode)
    el, er = plan.find_all_edges(dev(left), dev(right), thr)
    assert np.array_equal(host(el)[0], oracle.find_all_edges(left, thr, mode))
    assert np.array_equal(host(er)[0], oracle.find_all_edges(right, thr, mode))
    # and the packed copy the hot path consumes agrees with the u8 one
    web, best = plan.match_wta(1)
    ob, ow = oracle.hot_path(host(el)[0], host(er)[0], 16, sw, mode)
    assert np.array_equal(host(web)[0], ow) and np.array_equal(host(best)[0], ob)
    plan.close()


# ---------------------------------------------------------------------------
# step 3
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("times", [0, 1, 2, 5, 32])
def test_fill_web_holes(hip, times):
    rng = np.random.default_rng(times)
    web = rng.integers(1, 31, (2, 37, 53)).astype(np.int32)
    web[rng.random(web.shape) < 0.2] = 0
    web[:, 0, :5] = 0
    web[:, -1, -5:] = 0       # first/last rows: flat-index neighbours leave the array
    plan = hip.StereoPlan(53, 37, 30, 5, max_pairs=2)
    got = host(plan.fill_web_holes(dev(web), times))
    for i in range(2):
        assert np.array_equal(got[i], oracle.fill_web_holes(web[i], times)), (times, i)
    # no zeros -> identity
    full = np.maximum(web, 1)
    assert np.array_equal(host(plan.fill_web_holes(dev(full), times)), full)
    plan.close()


def test_min_max_and_contour(hip):
    rng = np.random.default_rng(3)
    web = rng.integers(1, 129, (3, 45, 70)).astype(np.int32)
    web[2] = 7                                  # constant image: zero interval
    plan = hip.StereoPlan(70, 45, 128, 5, max_pairs=3)
    mm = host(plan.image_min_max(dev(web)))
    assert np.array_equal(mm[:, 0], web.reshape(3, -1).min(1))
    assert np.array_equal(mm[:, 1], web.reshape(3, -1).max(1))
    for lines in (1, 3, 10, 127):
        out = host(plan.draw_contour_map(dev(web[:2]), lines))
        for i in range(2):
            assert np.array_equal(out[i], oracle.draw_contour_map(web[i], lines)), (lines, i)
    with pytest.raises(hip.capi.StereoHipError) as e:
        plan.draw_contour_map(dev(web), 10)
    assert e.value.code == hip.capi.SM_ERR_ZERO_DIV
    with pytest.raises(hip.capi.StereoHipError):
        plan.draw_contour_map(dev(web[:2]), 0)
    # the flag is cleared by the status call
    plan.draw_contour_map(dev(web[:2]), 10)
    plan.close()


# ---------------------------------------------------------------------------
# BASELINE.json configurations
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("cfg", ["C1", "C2"])
def test_baseline_config_full_size_vs_oracle(hip, cfg):
    w, h, d, sw, mode = CONFIGS[cfg]
    left, right = make_pair(w, h, d, seed=1)
    plan = hip.StereoPlan(w, h, d, sw, mode)
    el, er = plan.find_all_edges(dev(left), dev(right), 0.15)
    web, best = plan.match_wta(1)
    oel = oracle.find_all_edges(left, 0.15, mode)
    oer = oracle.find_all_edges(right, 0.15, mode)
    assert np.array_equal(host(el)[0], oel) and np.array_equal(host(er)[0], oer)
    ob, ow = oracle.hot_path(oel, oer, d, sw, mode)
    assert np.array_equal(host(web)[0], ow), plan.describe()
    assert np.array_equal(host(best)[0], ob)
    plan.close()


@pytest.mark.parametrize("cfg", ["C3", "C5"])
def test_baseline_config_4k_full_image_vs_oracle(hip, cfg):
    """The 4K configurations, EVERY pixel: edges, web and best of the whole image against
    the oracle (its separable window sum, run on row bands with a window-halo in threads)."""
    w, h, d, sw, mode = CONFIGS[cfg]
    left, right = make_pair(w, h, d, seed=2)
    plan = hip.StereoPlan(w, h, d, sw, mode)
    el, er = plan.find_all_edges(dev(left), dev(right), 0.15)
    web, best = plan.match_wta(1)
    oel = oracle.find_all_edges_banded(left, 0.15, mode)
    oer = oracle.find_all_edges_banded(right, 0.15, mode)
    assert np.array_equal(host(el)[0], oel) and np.array_equal(host(er)[0], oer), cfg
    ob, ow = oracle.hot_path_banded(oel, oer, d, sw, mode)
    assert np.array_equal(host(web)[0], ow), plan.describe()
    assert np.array_equal(host(best)[0], ob), plan.describe()
    plan.close()


@pytest.mark.parametrize("mode,d,sw", [("toroidal", 128, 9), ("ghost", 48, 9)])
def test_largest_image_8k_full_image_vs_oracle(hip, mode, d, sw):
    """Four times the largest BASELINE image: 7680 x 4320, every pixel of edges, web and best
    against the oracle -- 64-row tiles (the plan keeps the grid at one round of the chip), 30
    tile columns, and in the ghost case a shift count that does not fill the shift lanes."""
    w, h = 7680, 4320
    left, right = make_pair(w, h, d, seed=5)
    plan = hip.StereoPlan(w, h, d, sw, mode)
    g = plan.geometry()
    assert g["kernel"] == 4 and g["tiles_x"] == (30 if d == 128 else 15), plan.describe()
    el, er = plan.find_all_edges(dev(left), dev(right), 0.15)
    web, best = plan.match_wta(1)
    oel = oracle.find_all_edges_banded(left, 0.15, mode)
    oer = oracle.find_all_edges_banded(right, 0.15, mode)
    assert np.array_equal(host(el)[0], oel) and np.array_equal(host(er)[0], oer)
    ob, ow = oracle.hot_path_banded(oel, oer, d, sw, mode)
    assert np.array_equal(host(web)[0], ow), plan.describe()
    assert np.array_equal(host(best)[0], ob), plan.describe()
    plan.close()


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
def test_baseline_config_c4_batch_full_size(hip, mode):
    """C4 as one GPU sees it: a batch of 8 x 1080p pairs, 64 shifts, 7x7, in ONE launch --
    the tiling and the two-waves-per-SIMD kernel variant the batch geometry selects.
    All 8 maps, every pixel, against the oracle (toroidal is the configuration; the ghost
    border runs the same geometry)."""
    w, h, d, sw, _ = CONFIGS["C4"]
    pairs = 8
    ls, rs = zip(*[make_pair(w, h, d, seed=40 + j) for j in range(pairs)])
    left, right = np.stack(ls), np.stack(rs)
    plan = hip.StereoPlan(w, h, d, sw, mode, max_pairs=pairs)
    g = plan.geometry()
    # bit-sliced, at most two waves per SIMD: the capped one-wave build or two-wave workgroups
    assert g["kernel"] == 4 and (g["two_wave_variant"] == 1 or g["waves_per_workgroup"] == 2), plan.describe()
    assert g["tiles_x"] * g["tiles_y"] * pairs * g["waves_per_workgroup"] <= 2048, plan.describe()   # one round
    web, best = plan.run(dev(left), dev(right), 0.15, want_best=True)
    web_h, best_h = host(web), host(best)
    for j in range(pairs):
        oel = oracle.find_all_edges_banded(left[j], 0.15, mode)
        oer = oracle.find_all_edges_banded(right[j], 0.15, mode)
        ob, ow = oracle.hot_path_banded(oel, oer, d, sw, mode, n_bands=32)
        assert np.array_equal(web_h[j], ow), (j, plan.describe())
        assert np.array_equal(best_h[j], ob), (j, plan.describe())
    plan.close()


@pytest.mark.parametrize("cfg", ["C3", "C5"])
def test_baseline_config_full_size_properties(hip, cfg):
    """4K configurations: too big for the CPU oracle in seconds, so check (a) bands
    of the full-size result against the oracle run on the same rows (the result at
    a pixel depends only on rows within the window), (b) range / consistency
    invariants, (c) toroidal shift-equivariance."""
    w, h, d, sw, mode = CONFIGS[cfg]
    half = sw // 2
    left, right = make_pair(w, h, d, seed=2)
    plan = hip.StereoPlan(w, h, d, sw, mode)
    el, er = plan.find_all_edges(dev(left), dev(right), 0.15)
    web, best = plan.match_wta(1)
    web_h, best_h, el_h, er_h = host(web)[0], host(best)[0], host(el)[0], host(er)[0]

    # (a) bands of full-width rows: every window of rows [y0, y1) lies inside the
    # crop [y0-half, y1+half), so the crop's own borders never reach them.  In
    # ghost mode the image's top and bottom rows are real borders of the crop too;
    # in toroidal mode they would wrap differently and are covered by (c).
    bands = [(h // 2 - 7, h // 2 + 17)]
    if mode == "ghost":
        bands += [(0, 24), (h - 24, h)]
    for y0, y1 in bands:
        a, b = max(0, y0 - half), min(h, y1 + half)
        ob, ow = oracle.hot_path(el_h[a:b], er_h[a:b], d, sw, mode)
        lo = y0 - a
        assert np.array_equal(web_h[y0:y1], ow[lo:lo + y1 - y0]), (cfg, y0, plan.describe())
        assert np.array_equal(best_h[y0:y1], ob[lo:lo + y1 - y0]), (cfg, y0)
    e_rows = slice(h // 3, h // 3 + 16)
    crop = slice(h // 3 - 1, h // 3 + 17)
    assert np.array_equal(el_h[e_rows], oracle.find_all_edges(left[crop], 0.15, mode)[1:-1])

    # (b) invariants
    n = 2 * half + 1
    assert web_h.min() >= 1 and web_h.max() <= d
    assert best_h.min() >= 0 and best_h.max() <= n * n
    assert ((best_h == 0) <= (web_h == d)).all()

    # (c) toroidal: rolling both inputs rolls the outputs
    if mode == "toroidal":
        sx, sy = 1237, 411
        l2 = np.roll(left, (sy, sx), (0, 1)); r2 = np.roll(right, (sy, sx), (0, 1))
        web2, best2 = plan.run(dev(l2), dev(r2), 0.15, want_best=True)
        assert np.array_equal(host(web2)[0], np.roll(web_h, (sy, sx), (0, 1)))
        assert np.array_equal(host(best2)[0], np.roll(best_h, (sy, sx), (0, 1)))
    plan.close()


# ---------------------------------------------------------------------------
# non-default code paths (selected by environment at plan creation)
# ---------------------------------------------------------------------------

# kernel variants are chosen through sm_plan_create_ex (sm_plan_options); the library reads no
# environment variable
POPCOUNT = dict(kernel_family=1)
DS8 = dict(shifts_per_lane=8)
DS4 = dict(shifts_per_lane=4)
ONE_WAVE, TWO_WAVES = dict(workgroup_waves=1), dict(workgroup_waves=2)
NO_CAP2 = dict(no_two_wave_cap=1)


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("w,h,d,sw", [(300, 150, 128, 9), (71, 53, 30, 5), (130, 70, 64, 7), (90, 61, 64, 11)])
@pytest.mark.parametrize("variant", [POPCOUNT, DS8, dict(tile_h=5), dict(DS8, tile_h=7), NO_CAP2,
                                     ONE_WAVE, dict(ONE_WAVE, tile_h=5), dict(ONE_WAVE, **DS8),
                                     dict(TWO_WAVES, tile_h=5), dict(TWO_WAVES, tile_h=7, **DS8),
                                     DS4, dict(DS4, tile_h=5), dict(ONE_WAVE, **DS4), dict(TWO_WAVES, tile_h=6, **DS4),
                                     dict(no_four_shift_lanes=1)])
def test_alternative_kernels_match_oracle(hip, variant, mode, w, h, d, sw):
    """the popcount kernels (general fallback), the 8-shifts-per-lane bit-sliced
    variant and odd tile heights give the same bits as the default path"""
    le, re = rand_edges(w, h, seed=w + d)
    best, web, desc = hip_hot_path(hip, le, re, d, sw, mode, options=variant)
    if "workgroup_waves" in variant:
        assert ("two-wave workgroups" in desc) == (variant["workgroup_waves"] == 2), desc
    if "kernel_family" in variant:
        assert "tiled kernel" in desc
    elif variant.get("shifts_per_lane") == 8 and sw in (8, 9):       # the 8-per-lane variant is built for 9x9
        assert "lanes of 8" in desc
    elif variant.get("shifts_per_lane") == 4:                         # 4 per lane: every window, D <= 128
        assert "lanes of 4" in desc
    elif "no_four_shift_lanes" in variant:
        assert "lanes of 4" not in desc
    elif "no_two_wave_cap" in variant:              # small grids default to the 2-wave variant
        assert "2 waves/SIMD variant" not in desc
    elif variant.get("workgroup_waves") == 1 and "tile_h" in variant and sw in (5, 7) and "shifts_per_lane" not in variant:
        assert "2 waves/SIMD variant" in desc
    obest, oweb = oracle.hot_path(le, re, d, sw, mode)
    assert np.array_equal(web[0], oweb), desc
    assert np.array_equal(best[0], obest), desc


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("sw,d", [(3, 16), (5, 30), (7, 64), (9, 128), (11, 100), (13, 24), (17, 64), (21, 30)])
@pytest.mark.parametrize("h,tile_h", [(1, 4), (2, 4), (3, 2), (9, 4), (16, 4), (17, 4), (23, 3), (40, 16), (57, 8)])
def test_two_wave_workgroups_match_oracle(hip, mode, sw, d, h, tile_h):
    """k_match_bs<..., DUO>: the workgroup's two waves slide away from the tile's middle row
    and swap half of their first window through LDS.  Every window that is built, image
    heights around the tile boundaries (the last workgroup's lower wave with no rows at all,
    with some, with all; images shorter than one tile), both borders,
    shift counts that fill the lanes and that do not."""
    w = 100 + 3 * sw
    h = max(h, sw)              # the reference's rule: the window must fit the image
    le, re = rand_edges(w, h, seed=h * 31 + sw)
    best, web, desc = hip_hot_path(hip, le, re, d, sw, mode, options=dict(TWO_WAVES, tile_h=tile_h))
    assert "two-wave workgroups" in desc, desc
    obest, oweb = oracle.hot_path(le, re, d, sw, mode)
    assert np.array_equal(web[0], oweb), desc
    assert np.array_equal(best[0], obest), desc


@pytest.mark.parametrize("cfg,pairs", [("C3", 1), ("C4", 8), ("C5", 1)])
def test_match_launch_is_deterministic(hip, cfg, pairs):
    """300 launches on the same edges give the same maps every time, bit for bit: the waves of
    a two-wave workgroup swap sums through LDS between two barriers and share a SIMD with waves
    of other workgroups at alternating priorities -- a missing barrier or a lost wait would show
    up as a map that differs in SOME launch (the full-image parity tests look at one)."""
    w, h, d, sw, mode = CONFIGS[cfg]
    ls, rs = zip(*[make_pair(w, h, d, seed=70 + j) for j in range(pairs)])
    plan = hip.StereoPlan(w, h, d, sw, mode, max_pairs=pairs)
    plan.find_all_edges(dev(np.stack(ls)), dev(np.stack(rs)), 0.15, want_edges=False)
    web, best = plan.match_wta(pairs, want_best=True)
    ref_web, ref_best = web.clone(), best.clone()
    bad = torch.zeros((), dtype=torch.int32, device=web.device)       # launches that differed
    for i in range(300):
        web, best = plan.match_wta(pairs, want_best=True, web=web, best=best)
        bad += ((web != ref_web).any() | (best != ref_best).any()).to(torch.int32)
    assert int(bad.item()) == 0, (int(bad.item()), plan.describe())
    plan.close()


BUILT_BS = [(n, 16) for n in (3, 5, 7, 9, 11)] + [(n, 8) for n in range(3, 22, 2)] + [(n, 4) for n in range(3, 22, 2)]


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("fulld", [True, False])
@pytest.mark.parametrize("shape", [ONE_WAVE, dict(ONE_WAVE, **NO_CAP2), TWO_WAVES])
@pytest.mark.parametrize("n,ds", BUILT_BS)
def test_every_built_kernel_matches_oracle(hip, n, ds, shape, fulld, mode):
    """Every instantiation of k_match_bs the library holds -- window x shifts per lane x
    {shift range fills the lanes, does not} x border x {one wave, one wave capped at two
    per SIMD, two-wave workgroups} -- on tiles of 4 rows, i.e. with three slides per wave:
    the rows whose window rows come through the prefetched LDS reads.  (Round 2 found the
    two largest ghost windows wrong from the third row of a tile on -- registers spilled
    between an inline-asm read and its wait -- while every test used the default tile
    height of 2 rows that such small images get.)"""
    duo = shape["workgroup_waves"] == 2
    d = 2 * ds if fulld else 2 * ds - 3
    w, h = 150, n + 10
    le, re = rand_edges(w, h, seed=n * 100 + ds)
    best, web, desc = hip_hot_path(hip, le, re, d, n, mode, options=dict(shape, shifts_per_lane=ds, tile_h=4))
    assert f"lanes of {ds})" in desc and f"x{8 if duo else 4} px" in desc, desc
    assert ("two-wave workgroups" in desc) == duo, desc
    if "no_two_wave_cap" in shape:
        assert "2 waves/SIMD variant" not in desc, desc
    obest, oweb = oracle.hot_path(le, re, d, n, mode)
    assert np.array_equal(web[0], oweb), desc
    assert np.array_equal(best[0], obest), desc


LDS_MERGE = dict(lane_merge=2)


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("shape,tile_h", [(ONE_WAVE, 4), (ONE_WAVE, 6), (TWO_WAVES, 4), (TWO_WAVES, 9)])
@pytest.mark.parametrize("nl", [4, 8, 16, 32])
@pytest.mark.parametrize("n,ds", BUILT_BS)
def test_lane_merge_through_lds_matches_oracle(hip, n, ds, nl, shape, tile_h, mode):
    """k_match_bs with the lanes of a word merged through LDS every four rows (g.xmerge; the plan's choice at
    C3 / C5, forced here with lane_merge = 2 on EVERY built window x shifts per lane): 4, 8, 16 and 32 lanes
    per word (1, 2, 4, 8 lanes per item of a batch: no DPP level, one, two, three), shift ranges that fill the
    lanes and that do not, tiles whose last batch is whole (4 rows) and partial (6 = 4 + 2, 9 = 2 x 4 + 1),
    an image width that is not a multiple of 4 (scalar stores) and one that is, both borders, best map too."""
    fulld = (n + nl + tile_h) % 2 == 0
    d = nl * ds if fulld else nl * ds - 3
    w, h = (152 if (n + nl) % 4 < 2 else 150), n + 13
    le, re = rand_edges(w, h, seed=n * 100 + ds + nl)
    best, web, desc = hip_hot_path(hip, le, re, d, n, mode,
                                   options=dict(shape, shifts_per_lane=ds, tile_h=tile_h, **LDS_MERGE))
    assert "lanes merged through LDS" in desc and f"{nl} shift-lanes of {ds})" in desc, desc
    obest, oweb = oracle.hot_path(le, re, d, n, mode)
    assert np.array_equal(web[0], oweb), desc
    assert np.array_equal(best[0], obest), desc


@pytest.mark.parametrize("dtype", [torch.uint8, torch.uint16])
def test_lane_merge_through_lds_narrow_maps(hip, dtype):
    """... and its narrow result maps (dword / qword stores of 4 pixels), web only"""
    w, h, d, sw = 512, 37, 128, 9
    le, re = rand_edges(w, h, seed=3)
    _, oweb = oracle.hot_path(le, re, d, sw, "toroidal")
    for opts in (LDS_MERGE, dict(lane_merge=1)):
        plan = hip.StereoPlan(w, h, d, sw, "toroidal", options=opts)
        assert ("lanes merged through LDS" in plan.describe()) == (opts is LDS_MERGE)
        plan.load_edges(dev(le), dev(re))
        webn, _ = plan.match_wta(1, want_best=False, web_dtype=dtype)
        torch.cuda.synchronize()
        assert np.array_equal(host(webn.to(torch.int32))[0], oweb), plan.describe()
        plan.close()


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
def test_one_pixel_edge_kernel(hip, mode):
    left, right = make_pair(128, 66, 16, seed=4)
    # (the kernel used when the width is not a multiple of 4)
    plan = hip.StereoPlan(128, 66, 16, 5, mode, options=dict(edge_kernel=1))
    el, er = plan.find_all_edges(dev(left), dev(right), 0.15)
    assert np.array_equal(host(el)[0], oracle.find_all_edges(left, 0.15, mode))
    assert np.array_equal(host(er)[0], oracle.find_all_edges(right, 0.15, mode))
    plan.close()


def test_pipelined_runs_and_kernel_timing(hip):
    """sm_plan_set_pipelined: consecutive runs on alternating workspace halves give
    the same maps as sequential runs; sm_plan_time_kernels counts the launches"""
    w, h, d, sw = 320, 200, 64, 7
    pairs = [make_pair(w, h, d, seed=40 + i) for i in range(5)]
    plan = hip.StereoPlan(w, h, d, sw)
    want = []
    for l, r in pairs:
        web, _ = plan.run(dev(l), dev(r), 0.15)
        want.append(host(web)[0].copy())
    plan.set_pipelined(True)
    plan.time_kernels(len(pairs))
    inputs = [(dev(l), dev(r)) for l, r in pairs]
    torch.cuda.synchronize()
    outs = [plan.run(a, b, 0.15)[0] for a, b in inputs]      # back to back, no sync in between
    torch.cuda.synchronize()
    for got, exp in zip(outs, want):
        assert np.array_equal(host(got)[0], exp)
    ms, n = plan.kernel_ms()
    assert n == len(pairs) and ms > 0
    # and the oracle agrees with the first one
    o = oracle.pipeline(*pairs[0], 0.15, d, sw, step3=False)
    assert np.array_equal(want[0], o["web-1"])
    # sampled timing: every 3rd of 10 launches, at most 4 of them; back to sequential runs
    plan.set_pipelined(False)
    plan.time_kernels(4, every=3)
    for i in range(10):
        got = plan.run(*inputs[i % len(inputs)], 0.15)[0]
        assert np.array_equal(host(got)[0], want[i % len(inputs)])
    ms, n = plan.kernel_ms()
    assert n == 4 and ms > 0
    with pytest.raises(hip.capi.StereoHipError, match="sm_plan_time_stride"):
        plan.time_kernels(4, every=0)
    plan.time_kernels(0)
    # ordered pipelining (mode 2): the inputs of each run are produced on the launch stream
    # immediately before it (an asynchronous copy into a reused buffer); the edge kernel on the
    # plan's internal stream must wait for them
    plan.set_pipelined(2)
    buf_l, buf_r = torch.empty_like(inputs[0][0]), torch.empty_like(inputs[0][1])
    for i in range(2 * len(inputs)):
        a, b = inputs[i % len(inputs)]
        buf_l.copy_(a, non_blocking=True)
        buf_r.copy_(b, non_blocking=True)
        got = plan.run(buf_l, buf_r, 0.15)[0].clone()
        torch.cuda.synchronize()
        assert np.array_equal(host(got)[0], want[i % len(inputs)]), i
    plan.set_pipelined(False)
    # what two calls in flight could share is put in order by the library: ONE result map for all
    # calls (the last call's result is what stays), three rotating maps, a threshold that changes
    # from call to call (the decision tables are rebuilt)
    plan.set_pipelined(True)
    torch.cuda.synchronize()
    one = torch.empty((1, h, w), dtype=torch.int32, device="cuda")
    for a, b in inputs:
        plan.run(a, b, 0.15, web=one)
    torch.cuda.synchronize()
    assert np.array_equal(host(one)[0], want[-1])
    three = [torch.empty((1, h, w), dtype=torch.int32, device="cuda") for _ in range(3)]
    for i in range(3 * len(inputs)):
        plan.run(*inputs[i % len(inputs)], 0.15, web=three[i % 3])
    torch.cuda.synchronize()
    n = 3 * len(inputs)
    for i in range(n - 3, n):
        assert np.array_equal(host(three[i % 3])[0], want[i % len(inputs)]), i
    thr = [0.05, 0.3, 0.15, 0.6]
    outs = [plan.run(*inputs[0], t)[0] for t in thr]
    torch.cuda.synchronize()
    plan.set_pipelined(False)
    for t, got in zip(thr, outs):
        assert np.array_equal(host(got)[0], host(plan.run(*inputs[0], t)[0])[0]), t
    plan.close()
    # a fallback kernel has ONE int32 staging map for its narrow results: pipelined calls share it
    plan = hip.StereoPlan(w, h, d, sw, options=dict(kernel_family=1))
    narrow = [torch.empty((1, h, w), dtype=torch.uint8, device="cuda") for _ in range(len(inputs))]
    plan.set_pipelined(True)
    for (a, b), o in zip(inputs, narrow):
        plan.run(a, b, 0.15, web=o, web_dtype=torch.uint8)
    torch.cuda.synchronize()
    for o, exp in zip(narrow, want):
        assert np.array_equal(host(o)[0].astype(np.int32), exp)
    plan.close()


def test_run_after_an_input_ready_event(hip):
    """sm_run_after: the call's only input dependency is an event.  The uploads of six pairs go out on a COPY stream
    into six buffer pairs, an event behind each; the calls go out on the compute stream at once, without any other
    ordering -- every map must be its pair's (a lane that did not wait shows up as garbage or a neighbour's map).
    Small launches (this one: a few dozen waves) take the plan's two lanes by themselves; a launch of >= 2048 waves
    (a batch of 4 x 1080p... here: forced by a tall batch) stays in stream order.  Also inside a graph."""
    w, h, d, sw = 320, 200, 64, 7
    pairs = [make_pair(w, h, d, seed=300 + i) for i in range(6)]
    want = [oracle.pipeline(l, r, 0.15, d, sw, step3=False)["web-1"] for l, r in pairs]
    plan = hip.StereoPlan(w, h, d, sw)
    plan.prepare_threshold(0.15)
    hl = [torch.from_numpy(l).pin_memory() for l, _ in pairs]
    hr = [torch.from_numpy(r).pin_memory() for _, r in pairs]
    dl = [torch.zeros((h, w), dtype=torch.uint8, device="cuda") for _ in pairs]
    dr = [torch.zeros((h, w), dtype=torch.uint8, device="cuda") for _ in pairs]
    webs = [torch.zeros((1, h, w), dtype=torch.int32, device="cuda") for _ in pairs]
    copy = torch.cuda.Stream()
    for rep in range(3):
        for o in webs:
            o.zero_()
        for a, b in zip(dl, dr):
            a.zero_(); b.zero_()
        torch.cuda.synchronize()
        evs = []
        with torch.cuda.stream(copy):
            for i in range(len(pairs)):
                dl[i].copy_(hl[i], non_blocking=True)
                dr[i].copy_(hr[i], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy)
                evs.append(ev)
        for i in range(len(pairs)):
            plan.run_after(dl[i], dr[i], 0.15, inputs_ready=evs[i], web=webs[i])
        torch.cuda.synchronize()
        for i in range(len(pairs)):
            assert np.array_equal(host(webs[i])[0], want[i]), (rep, i)
    # argument checks as sm_run's
    import ctypes as C
    lib = hip.capi.lib
    assert lib.sm_run_after(plan._h, C.c_void_p(dl[0].data_ptr()), C.c_void_p(dr[0].data_ptr()), 0.15, 2,
                            C.c_void_p(webs[0].data_ptr()), 0, None, None, None) == hip.capi.SM_ERR_ARG      # pairs > max_pairs
    assert lib.sm_run_after(None, None, None, 0.15, 1, None, 0, None, None, None) == hip.capi.SM_ERR_ARG
    assert lib.sm_run_after(plan._h, C.c_void_p(dl[0].data_ptr()), C.c_void_p(dr[0].data_ptr()), 1.5, 1,
                            C.c_void_p(webs[0].data_ptr()), 0, None, None, None) == hip.capi.SM_ERR_ARG      # threshold
    assert lib.sm_run_after(plan._h, C.c_void_p(dl[0].data_ptr()), C.c_void_p(dr[0].data_ptr()), 0.15, 1,
                            None, 0, None, None, None) == hip.capi.SM_ERR_ARG                                   # no map
    # ... and a refused call leaves the plan usable (the lanes were not touched, or were joined)
    assert np.array_equal(host(plan.run_after(dl[2], dr[2], 0.15)[0])[0], want[2])
    # mixed with plain runs, a changing threshold and ONE shared map: still ordered
    one = torch.zeros((1, h, w), dtype=torch.int32, device="cuda")
    for i in range(len(pairs)):
        plan.run_after(dl[i], dr[i], 0.15, web=one)
        if i % 2:
            plan.run(dl[i], dr[i], 0.15, web=one)
    torch.cuda.synchronize()
    assert np.array_equal(host(one)[0], want[-1])
    # inside a graph: the overlap is kept (the plan forks and joins its lanes by the capture's own events)
    g = torch.cuda.CUDAGraph()
    for o in webs:
        o.zero_()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for i in range(len(pairs)):
            plan.run_after(dl[i], dr[i], 0.15, web=webs[i])
    g.replay()
    torch.cuda.synchronize()
    for i in range(len(pairs)):
        assert np.array_equal(host(webs[i])[0], want[i]), i
    plan.close()
    # a launch that fills the chip more than twice over stays on the caller's stream, behind the event
    plan = hip.StereoPlan(1920, 1080, 64, 7, max_pairs=4)
    big = [make_pair(1920, 1080, 64, seed=310 + i) for i in range(4)]
    L = torch.from_numpy(np.stack([p[0] for p in big])).pin_memory()
    R = torch.from_numpy(np.stack([p[1] for p in big])).pin_memory()
    gl, gr = torch.zeros_like(L, device="cuda"), torch.zeros_like(R, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(copy):
        gl.copy_(L, non_blocking=True)
        gr.copy_(R, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(copy)
    web, _ = plan.run_after(gl, gr, 0.15, inputs_ready=ev)
    torch.cuda.synchronize()
    ref, _ = plan.run(gl, gr, 0.15)
    assert torch.equal(web, ref)
    o = oracle.pipeline(big[2][0], big[2][1], 0.15, 64, 7, step3=False)
    assert np.array_equal(host(web)[2], o["web-1"])
    plan.close()


def test_geometry_of_a_caller_compiled_against_an_older_header(hip):
    """sm_plan_geometry_sized writes no more than the caller's struct holds (sm_geometry grows at its end from round to
    round: ADVICE r04), and zeroes what a struct newer than the library has beyond it"""
    import ctypes as C
    lib, check = hip.capi.lib, hip.capi.check
    plan = hip.StereoPlan(320, 200, 64, 7)
    full = plan.geometry()
    names = [n for n, _ in hip.capi.Geometry._fields_]
    buf = (C.c_int * 40)(*([-7] * 40))
    ptr = C.cast(buf, C.POINTER(hip.capi.Geometry))
    check(lib.sm_plan_geometry_sized(plan._h, ptr, 8 * 4))              # an old struct of 8 fields
    assert [buf[i] for i in range(8)] == [full[n] for n in names[:8]] and all(buf[i] == -7 for i in range(8, 40))
    check(lib.sm_plan_geometry_sized(plan._h, ptr, 30 * 4))             # a newer one of 30
    assert [buf[i] for i in range(len(names))] == [full[n] for n in names]
    assert all(buf[i] == 0 for i in range(len(names), 30)) and all(buf[i] == -7 for i in range(30, 40))
    assert lib.sm_plan_geometry_sized(plan._h, ptr, 2) == hip.capi.SM_ERR_ARG
    plan.close()


def test_runs_captured_into_a_graph(hip):
    """sm_run inside a stream capture (torch.cuda.graph; include/stereo_hip.h "STREAM CAPTURE"): a plain plan and a
    PIPELINED one -- whose lanes must leave and rejoin the capturing stream by events recorded inside the capture; round 4's
    attempt crashed the process -- replay to the oracle's maps, also after eager calls in between; what cannot be
    captured (threshold tables not built, timing armed, the narrow staging map not allocated) is refused with a message
    that names the remedy and leaves the capture usable."""
    w, h, d, sw = 320, 200, 64, 7
    pairs = [make_pair(w, h, d, seed=70 + i) for i in range(4)]
    inputs = [(dev(l), dev(r)) for l, r in pairs]
    want = [oracle.pipeline(l, r, 0.15, d, sw, step3=False)["web-1"] for l, r in pairs]
    for pipelined in (False, True):
        plan = hip.StereoPlan(w, h, d, sw)
        plan.prepare_threshold(0.15)
        plan.run(*inputs[0], 0.15)
        plan.set_pipelined(pipelined)
        for a, b in inputs:               # (pipelined: eager calls whose release events exist before the capture begins)
            plan.run(a, b, 0.15)
        torch.cuda.synchronize()
        webs = [torch.zeros((1, h, w), dtype=torch.int32, device="cuda") for _ in inputs]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for (a, b), o in zip(inputs, webs):
                plan.run(a, b, 0.15, web=o)
        for rep in range(3):
            for o in webs:
                o.zero_()
            g.replay()
            torch.cuda.synchronize()
            for o, exp in zip(webs, want):
                assert np.array_equal(host(o)[0], exp), (pipelined, rep)
            got = plan.run(*inputs[rep], 0.15)[0]           # an eager call between replays
            torch.cuda.synchronize()
            assert np.array_equal(host(got)[0], want[rep]), (pipelined, rep)
        # refused, cleanly
        plan.time_kernels(2)
        with pytest.raises(hip.capi.StereoHipError, match="sm_plan_time_kernels"):
            with torch.cuda.graph(torch.cuda.CUDAGraph(), capture_error_mode="thread_local"):
                plan.run(*inputs[0], 0.15)
        plan.time_kernels(0)
        with pytest.raises(hip.capi.StereoHipError, match="sm_plan_prepare_threshold"):
            with torch.cuda.graph(torch.cuda.CUDAGraph(), capture_error_mode="thread_local"):
                plan.run(*inputs[0], 0.33)
        # ... and the plan still works, captured and eager
        g2 = torch.cuda.CUDAGraph()
        o2 = torch.zeros((1, h, w), dtype=torch.int32, device="cuda")
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            plan.run(*inputs[2], 0.15, web=o2)
        g2.replay()
        torch.cuda.synchronize()
        assert np.array_equal(host(o2)[0], want[2])
        assert np.array_equal(host(plan.run(*inputs[3], 0.15)[0])[0], want[3])
        plan.close()
    # the narrow staging map of a fallback kernel: an allocation, not capturable
    plan = hip.StereoPlan(w, h, d, sw, options=dict(kernel_family=1))
    plan.prepare_threshold(0.15)
    plan.run(*inputs[0], 0.15)
    torch.cuda.synchronize()
    o8 = torch.zeros((1, h, w), dtype=torch.uint8, device="cuda")
    with pytest.raises(hip.capi.StereoHipError, match="sm_plan_reserve_narrow"):
        with torch.cuda.graph(torch.cuda.CUDAGraph(), capture_error_mode="thread_local"):
            plan.run(*inputs[0], 0.15, web=o8, web_dtype=torch.uint8)
    plan.reserve_narrow()
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3, capture_error_mode="thread_local"):
        plan.run(*inputs[1], 0.15, web=o8, web_dtype=torch.uint8)
    g3.replay()
    torch.cuda.synchronize()
    assert np.array_equal(host(o8)[0].astype(np.int32), want[1])
    plan.close()
    # the SAD / SSD cost mode captures as any launch
    plan = hip.StereoPlan(w, h, d, 9)
    l, r = inputs[0]
    ref_web, ref_best = plan.cost_wta(l, r, "sad")
    cw, cb = torch.zeros_like(ref_web), torch.zeros_like(ref_best)
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4, capture_error_mode="thread_local"):
        plan.cost_wta(l, r, "sad", web=cw, best=cb)
    g4.replay()
    torch.cuda.synchronize()
    assert torch.equal(cw, ref_web) and torch.equal(cb, ref_best)
    plan.close()


def test_c_abi_three_streams_chained_with_events(hip):
    """uploads, kernels and downloads on a stream each (the pattern of stereopar_batch.c and
    tools/e2e_bench.py): 12 DIFFERENT 1080p pairs through 3 buffer sets; sm_event_record /
    sm_stream_wait_event / sm_event_sync are all that orders them.  Every map must be the
    map of ITS pair -- a missing wait shows up as a neighbour's result or a torn one."""
    import ctypes as C
    lib, check = hip.capi.lib, hip.capi.check
    w, h, d, sw = 1920, 1080, 64, 7
    n, slots, npairs = w * h, 3, 12
    vp = C.c_void_p
    pairs = [make_pair(w, h, d, seed=900 + j) for j in range(4)]

    def make(fn, *a):
        p = vp()
        check(fn(*a, C.byref(p)))
        return p
    st_up, st_run, st_down = (make(lib.sm_stream_create, 0) for _ in range(3))
    plan = make(lib.sm_plan_create, 0, w, h, d, sw, 0, 1)
    S = [dict(hl=make(lib.sm_host_alloc, n), hr=make(lib.sm_host_alloc, n), hw=make(lib.sm_host_alloc, n),
              dl=make(lib.sm_malloc, 0, n), dr=make(lib.sm_malloc, 0, n), dw=make(lib.sm_malloc, 0, n),
              up=make(lib.sm_event_create, 0), ran=make(lib.sm_event_create, 0),
              down=make(lib.sm_event_create, 0), pair=None) for _ in range(slots)]
    want = {}
    got = []

    def collect(s):
        check(lib.sm_event_sync(0, s["down"]))
        got.append((s["pair"], np.ctypeslib.as_array(C.cast(s["hw"], C.POINTER(C.c_uint8)), (h, w)).copy()))

    for k in range(npairs):
        s = S[k % slots]
        if s["pair"] is not None:
            collect(s)
            check(lib.sm_stream_wait_event(0, st_up, s["ran"]))
        j = k % len(pairs)
        C.memmove(s["hl"], pairs[j][0].ctypes.data, n)      # the slot's host buffers are free: collected
        C.memmove(s["hr"], pairs[j][1].ctypes.data, n)
        check(lib.sm_memcpy_h2d_async(0, s["dl"], s["hl"], n, st_up))
        check(lib.sm_memcpy_h2d_async(0, s["dr"], s["hr"], n, st_up))
        check(lib.sm_event_record(0, s["up"], st_up))
        check(lib.sm_stream_wait_event(0, st_run, s["up"]))
        if s["pair"] is not None:
            check(lib.sm_stream_wait_event(0, st_run, s["down"]))
        check(lib.sm_run_typed(plan, s["dl"], s["dr"], 0.15, 1, s["dw"], hip.capi.SM_WEB_U8, None, st_run))
        check(lib.sm_event_record(0, s["ran"], st_run))
        check(lib.sm_stream_wait_event(0, st_down, s["ran"]))
        check(lib.sm_memcpy_d2h_async(0, s["hw"], s["dw"], n, st_down))
        check(lib.sm_event_record(0, s["down"], st_down))
        s["pair"] = j
    for k in range(npairs, npairs + slots):
        collect(S[k % slots])
    assert len(got) == npairs
    for j, web in got:
        if j not in want:
            want[j] = oracle.pipeline(*pairs[j], 0.15, d, sw, step3=False)["web-1"].astype(np.uint8)
        assert np.array_equal(web, want[j]), j
    with pytest.raises(hip.capi.StereoHipError, match="sm_event_record"):
        check(lib.sm_event_record(0, None, st_up))
    lib.sm_plan_destroy(plan)
    for s in S:
        for k in ("dl", "dr", "dw"):
            check(lib.sm_free(0, s[k]))
        for k in ("hl", "hr", "hw"):
            check(lib.sm_host_free(s[k]))
        for k in ("up", "ran", "down"):
            check(lib.sm_event_destroy(0, s[k]))
    for st in (st_up, st_run, st_down):
        check(lib.sm_stream_destroy(0, st))


def test_c_abi_alone_with_pinned_async_transfers(hip):
    """the boundary without torch: device memory, pinned host memory, a stream and
    async copies all come from the C ABI (what a C caller would do)"""
    import ctypes as C
    lib, check = hip.capi.lib, hip.capi.check
    w, h, d, sw = 200, 120, 30, 7
    left, right = make_pair(w, h, d, seed=8)
    n = w * h
    vp = C.c_void_p
    hl, hr, hw_, dl, dr, dw, st, plan = (vp() for _ in range(8))
    for p, nb in ((hl, n), (hr, n), (hw_, 4 * n)):
        check(lib.sm_host_alloc(nb, C.byref(p)))
    for p, nb in ((dl, n), (dr, n), (dw, 4 * n)):
        check(lib.sm_malloc(0, nb, C.byref(p)))
    C.memmove(hl, left.ctypes.data, n)
    C.memmove(hr, right.ctypes.data, n)
    check(lib.sm_stream_create(0, C.byref(st)))
    check(lib.sm_plan_create(0, w, h, d, sw, 0, 1, C.byref(plan)))
    check(lib.sm_memcpy_h2d_async(0, dl, hl, n, st))
    check(lib.sm_memcpy_h2d_async(0, dr, hr, n, st))
    check(lib.sm_run(plan, dl, dr, 0.15, 1, dw, None, st))
    check(lib.sm_memcpy_d2h_async(0, hw_, dw, 4 * n, st))
    check(lib.sm_stream_sync(0, st))
    web = np.ctypeslib.as_array(C.cast(hw_, C.POINTER(C.c_int32)), (h, w)).copy()
    want = oracle.pipeline(left, right, 0.15, d, sw, step3=False)["web-1"]
    assert np.array_equal(web, want)
    lib.sm_plan_destroy(plan)
    check(lib.sm_stream_destroy(0, st))
    for p in (dl, dr, dw):
        check(lib.sm_free(0, p))
    for p in (hl, hr, hw_):
        check(lib.sm_host_free(p))


# ---------------------------------------------------------------------------
# SAD / SSD cost mode -- parity UNPINNED: the reference has no such mode; the only
# oracle is the build's own C definition (oracle/stereo_oracle.c smo_cost_hot_path)
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("cost", ["sad", "ssd"])
@pytest.mark.parametrize("w,h,d,sw", [(64, 40, 16, 5), (131, 67, 30, 9), (200, 50, 128, 9), (96, 70, 64, 7),
                                      (80, 64, 40, 21), (57, 33, 8, 1), (300, 41, 256, 11), (40, 30, 100, 25)])
def test_cost_mode_matches_own_oracle(hip, mode, cost, w, h, d, sw):
    kind = "noise" if (w + h) % 2 else "scene"
    left, right = make_pair(w, h, d, seed=w + sw, kind=kind)
    plan = hip.StereoPlan(w, h, d, sw, mode)
    web, best = plan.cost_wta(dev(left), dev(right), cost)
    ob, ow = oracle.cost_hot_path(left, right, d, sw, mode, cost)
    assert np.array_equal(host(web)[0], ow), (cost, mode)
    assert np.array_equal(host(best)[0], ob), (cost, mode)
    plan.close()


SAD_QS = [  # (w, h, D, S): quads per lane / shift-lanes / pixels per lane of the quad-SAD kernel
    (64, 40, 16, 5),        # 5 quads
    (131, 67, 30, 9),       # 9 quads, width not a multiple of 4 (bytewise staging)
    (96, 70, 64, 7),        # 17 quads
    (200, 50, 128, 9),      # 33 quads, the headline geometry
    (300, 41, 256, 11),     # 2 shift-lanes of 33 quads
    (132, 36, 500, 3),      # 4 shift-lanes, 3x3 (no full group), D > W
    (260, 33, 100, 13),     # 33 quads of which the last 8 hold no shift below D (early exit)
    (72, 45, 140, 15),      # 2 shift-lanes, the lower one full, the upper one mostly empty; 15x15
    (40, 30, 17, 5),        # D = 4 k + 1
    (16, 12, 3, 3),         # tiny
    # windows above 15 x 15 (SAD: two packed sums per shift; SSD takes the general kernel there)
    (96, 44, 30, 21),       # the reference's defaults: 21 x 21, 30 shifts -> 9 quads x 2 pixels
    (61, 40, 12, 17),       # 5 quads x 4 pixels, width not a multiple of 4
    (140, 43, 64, 19),      # 17 quads x 2 pixels
    (120, 45, 200, 21),     # 4 shift-lanes of 17 quads, the last one all beyond D in its second chunk
    (88, 47, 240, 17),      # the most shifts the 8-bit shift field of the split keys is used for
    (90, 46, 250, 21),      # ... beyond it: the general kernel
]


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("tile_h", [0, 3])
@pytest.mark.parametrize("cost", ["sad", "ssd"])
@pytest.mark.parametrize("w,h,d,sw", SAD_QS)
def test_quad_sad_kernel_matches_own_oracle(hip, mode, tile_h, cost, w, h, d, sw):
    """k_sad_qs / k_ssd_mfma / k_ssd_dot: every shape of (quads per lane, shift-lanes), windows 3 .. 21 (SAD from 17 on with
    two packed sums per shift; SSD: .. 11, the larger ones take the general kernel), both borders, tiles of 3 rows (several slides per wave, a
    ragged last tile) and the plan's own height, unaligned input"""
    left, right = make_pair(w, h, d, seed=w * 3 + d, kind="noise" if (w + d) % 3 == 0 else "scene")
    if (w + h) % 2:             # saturate some pixels: 0 and 255 are the masked-SAD corner cases
        left[::3, ::5] = 0; left[1::4, 2::7] = 255; right[::5, ::3] = 255; right[2::3, 1::4] = 0
    plan = hip.StereoPlan(w, h, d, sw, mode, options=dict(cost_tile_h=tile_h))
    web, best = plan.cost_wta(dev(left), dev(right), cost)
    ob, ow = oracle.cost_hot_path(left, right, d, sw, mode, cost)
    assert np.array_equal(host(web)[0], ow), (mode, w, h, d, sw)
    assert np.array_equal(host(best)[0], ob), (mode, w, h, d, sw)
    # the general masked kernel (windows beyond 15 x 15 / 11 x 11, more than 512 shifts) on the same input
    gen = hip.StereoPlan(w, h, d, sw, mode, options=dict(cost_kernel=1))
    web2, best2 = gen.cost_wta(dev(left), dev(right), cost)
    assert torch.equal(web2, web) and torch.equal(best2, best)
    plan.close(); gen.close()
    if mode == "ghost":         # the border strip x < half: k_cost_strip by default, the general masked kernel on request
        old = hip.StereoPlan(w, h, d, sw, mode, options=dict(cost_kernel=3, cost_tile_h=tile_h))
        web4, best4 = old.cost_wta(dev(left), dev(right), cost)
        assert torch.equal(web4, web) and torch.equal(best4, best)
        old.close()
    if cost == "ssd":           # the plan's choice is the matrix-core kernel (k_ssd_mfma); the byte-dot kernel on request
        dot = hip.StereoPlan(w, h, d, sw, mode, options=dict(cost_kernel=2, cost_tile_h=tile_h))
        web3, best3 = dot.cost_wta(dev(left), dev(right), cost)
        assert torch.equal(web3, web) and torch.equal(best3, best)
        dot.close()


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("sw", [15, 17, 19, 21])
def test_quad_sad_largest_possible_sums(hip, mode, sw):
    """black against white: every tap costs 255, the window sums are the largest the packed 16-bit
    accumulators ever see (15 x 15: 57 375 in one; 21 x 21: 64 260 + 48 195 in two), every shift ties
    (the first must win), and a lone matching column must still be found"""
    w, h, d = 100, sw + 7, 40
    left = np.zeros((h, w), np.uint8)
    right = np.full((h, w), 255, np.uint8)
    right[:, 60] = 0                       # one column of zero cost: windows that hold it prefer its shift
    plan = hip.StereoPlan(w, h, d, sw, mode)
    web, best = plan.cost_wta(dev(left), dev(right), "sad")
    ob, ow = oracle.cost_hot_path(left, right, d, sw, mode, "sad")
    assert np.array_equal(host(web)[0], ow) and np.array_equal(host(best)[0], ob)
    if mode == "toroidal":
        assert int(ob.max()) == sw * sw * 255
    plan.close()


@pytest.mark.parametrize("cost", ["sad", "ssd"])
def test_quad_sad_unaligned_images_and_batches(hip, cost):
    """image pointers that are not dword-aligned take the bytewise staging path; pairs of a batch
    are independent launches in z"""
    w, h, d, sw, n = 120, 50, 64, 9, 3
    pairs = [make_pair(w, h, d, seed=90 + i) for i in range(n)]
    left = np.stack([p[0] for p in pairs]); right = np.stack([p[1] for p in pairs])
    plan = hip.StereoPlan(w, h, d, sw, "toroidal", max_pairs=n)
    web, best = plan.cost_wta(dev(left), dev(right), cost)
    buf_l = torch.empty(n * w * h + 1, dtype=torch.uint8, device="cuda")
    buf_r = torch.empty(n * w * h + 3, dtype=torch.uint8, device="cuda")
    ul, ur = buf_l[1:].view(n, h, w), buf_r[3:].view(n, h, w)
    ul.copy_(dev(left)); ur.copy_(dev(right))
    web_u, best_u = plan.cost_wta(ul, ur, cost)
    for i in range(n):
        ob, ow = oracle.cost_hot_path(pairs[i][0], pairs[i][1], d, sw, "toroidal", cost)
        assert np.array_equal(host(web)[i], ow) and np.array_equal(host(best)[i], ob), i
    assert torch.equal(web_u, web) and torch.equal(best_u, best)
    plan.close()


def test_cost_kernels_random_shapes_match_own_oracle(hip):
    """A seeded sweep of 160 random shapes through sm_cost_wta: widths from 8 to 300 (multiples of 4 and
    not), heights down to the window, 1 .. 300 shifts, every window 3 .. 17, both borders and costs, tile
    heights 0 (the plan's) .. 9, images with saturated pixels -- the index arithmetic of the quad-SAD and
    byte-dot kernels (aligned window starts, shift quads, ghost strip, ragged last tiles) against the
    build's own CPU definition."""
    # (SM_SOAK_COST=<cases> [SM_SOAK_SEED=<seed>]: the same sweep, longer and wider, as a one-off soak)
    soak = int(os.environ.get("SM_SOAK_COST", "0"))
    rng = np.random.default_rng(int(os.environ.get("SM_SOAK_SEED", "1")) + 977 if soak else 20261004)
    for case in range(soak or 160):
        sw = int(rng.choice([3, 5, 7, 9, 11, 13, 15, 17, 4, 8, 19, 21, 20]))
        w = int(rng.integers(max(8, sw), 701 if soak else 301))
        if case % 3 == 0:
            w = 4 * ((w + 3) // 4)
        h = int(rng.integers(sw, 60))
        d = int(rng.choice([1, 2, 3, 5, 16, 17, 30, 31, 32, 33, 63, 64, 65, 100, 128, 129, 130, 200, 256, 257, 300]))
        mode = "ghost" if rng.integers(2) else "toroidal"
        cost = "ssd" if rng.integers(2) else "sad"
        th = int(rng.choice([0, 0, 1, 2, 3, 5, 9]))
        left = rng.integers(0, 256, (h, w), dtype=np.uint8)
        right = np.roll(left, int(rng.integers(0, max(1, min(d, w)))), 1)
        right = np.clip(right.astype(np.int32) + rng.integers(-3, 4, (h, w)), 0, 255).astype(np.uint8)
        if case % 4 == 0:
            left[rng.random((h, w)) < 0.2] = 0
            right[rng.random((h, w)) < 0.2] = 255
        ck = 2 if cost == "ssd" and case % 3 == 1 else 0         # SSD: a third of the cases on the byte-dot kernel
        if cost == "sad" and case % 5 == 2:
            ck = 4                                               # SAD: a fifth on the round-4 kernel (every window row from scratch)
        wv = int(rng.choice([0, 0, 1, 2, 4]))                    # waves per workgroup sharing the staged rows (round 5)
        plan = hip.StereoPlan(w, h, d, sw, mode, options=dict(cost_tile_h=th, cost_kernel=ck, cost_workgroup_waves=wv))
        web, best = plan.cost_wta(dev(left), dev(right), cost)
        ob, ow = oracle.cost_hot_path(left, right, d, sw, mode, cost)
        assert np.array_equal(host(web)[0], ow), (case, w, h, d, sw, mode, cost, th, ck, wv)
        assert np.array_equal(host(best)[0], ob), (case, w, h, d, sw, mode, cost, th, ck, wv)
        plan.close()
    if soak:
        print(f"cost soak: {soak} cases, all equal to the build's own CPU definition")


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("cost", ["sad", "ssd"])
def test_cost_workgroups_of_several_waves_share_their_rows(hip, mode, cost):
    """k_sad_pc / k_ssd_mfma in workgroups of 1, 2 and 4 waves (round 5: the waves stage one set of rows and share it --
    one right-image span for all of them): images a few workgroups wide and not a multiple of any workgroup width,
    shift counts that give 1 .. 16 shift lanes per pixel group (SAD) and 1 .. 9 position blocks (SSD), tiles of the
    plan's height and of 5 rows, web and best against the CPU definition; the plan's own choice equals them all."""
    rng = np.random.default_rng(61)
    for sw, d, w, h in ((9, 128, 700, 41), (11, 256, 452, 33), (7, 64, 1030, 27), (5, 16, 530, 19), (13, 200, 390, 29),
                        (3, 500 if cost == "sad" else 250, 300, 17)):
        left = rng.integers(0, 256, (h, w), dtype=np.uint8)
        right = np.roll(left, int(rng.integers(0, min(d, w))), 1)
        right = np.clip(right.astype(np.int32) + rng.integers(-2, 3, (h, w)), 0, 255).astype(np.uint8)
        left[::5, ::9] = 0; right[1::4, 3::7] = 255
        if cost == "ssd" and sw > 11:
            continue
        ob, ow = oracle.cost_hot_path(left, right, d, sw, mode, cost)
        for wv in (0, 1, 2, 4):
            for th in (0, 5):
                plan = hip.StereoPlan(w, h, d, sw, mode, options=dict(cost_workgroup_waves=wv, cost_tile_h=th))
                web, best = plan.cost_wta(dev(left), dev(right), cost)
                assert np.array_equal(host(web)[0], ow), (cost, mode, sw, d, w, wv, th)
                assert np.array_equal(host(best)[0], ob), (cost, mode, sw, d, w, wv, th)
                plan.close()


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
def test_ssd_matrix_core_kernel_every_instantiation(hip, mode):
    """k_ssd_mfma<N, NB>: every window 3 .. 11 x every block count 1 .. 9, with the shift counts either side of
    each block boundary (the band's partial blocks: D a multiple of 32 takes the complementary-triangle masks,
    anything else the per-key compares on the last two blocks), web and best, against the CPU definition and the
    byte-dot kernel; ghost mode adds k_cost_strip<half, SSD> behind it."""
    rng = np.random.default_rng(41)
    for sw in (3, 5, 7, 9, 11):
        for d in (1, 2, 31, 32, 33, 34, 64, 65, 95, 96, 97, 128, 129, 160, 161, 191, 192, 193, 224, 225, 255, 256):
            w, h = 76, sw + 5
            left = rng.integers(0, 256, (h, w), dtype=np.uint8)
            right = np.roll(left, int(rng.integers(0, min(d, w))), 1)
            right = np.clip(right.astype(np.int32) + rng.integers(-2, 3, (h, w)), 0, 255).astype(np.uint8)
            if d % 3 == 0:
                left[::2, ::7] = 0; right[1::3, ::5] = 255
            plan = hip.StereoPlan(w, h, d, sw, mode, options=dict(cost_tile_h=3 if d % 2 else 0))
            web, best = plan.cost_wta(dev(left), dev(right), "ssd")
            ob, ow = oracle.cost_hot_path(left, right, d, sw, mode, "ssd")
            assert np.array_equal(host(web)[0], ow), (mode, sw, d)
            assert np.array_equal(host(best)[0], ob), (mode, sw, d)
            plan.close()


@pytest.mark.parametrize("cost", ["sad", "ssd"])
def test_ghost_strip_kernel_every_instantiation(hip, cost):
    """k_cost_strip<half, SSD, slots>: every half-window 1 .. 10 (SSD: .. 5), one and two shifts per thread,
    images no wider than the window, heights around its tile boundaries, against the CPU definition and the
    general masked kernel on the same columns (cost_kernel = 3)"""
    rng = np.random.default_rng(43)
    for half in range(1, 11 if cost == "sad" else 6):
        sw = 2 * half + 1
        for d, w, h in ((40, 64, sw + 9), (300, 48, sw + 2), (200, sw, 37), (256, 96, 70)):
            if cost == "sad" and sw > 15 and d > 240:
                d = 240                 # (the split-accumulator SAD kernel's shift field)
            left = rng.integers(0, 256, (h, w), dtype=np.uint8)
            right = rng.integers(0, 256, (h, w), dtype=np.uint8)
            plan = hip.StereoPlan(w, h, d, sw, "ghost")
            web, best = plan.cost_wta(dev(left), dev(right), cost)
            ob, ow = oracle.cost_hot_path(left, right, d, sw, "ghost", cost)
            assert np.array_equal(host(web)[0], ow), (cost, half, d, w, h)
            assert np.array_equal(host(best)[0], ob), (cost, half, d, w, h)
            old = hip.StereoPlan(w, h, d, sw, "ghost", options=dict(cost_kernel=3))
            web2, best2 = old.cost_wta(dev(left), dev(right), cost)
            assert torch.equal(web2, web) and torch.equal(best2, best)
            plan.close(); old.close()


@pytest.mark.parametrize("cfg,cost", [("C3", "sad"), ("C5", "sad"), ("C3", "ssd"), ("C5", "ssd")])
def test_cost_mode_4k_full_image_vs_own_oracle(hip, cfg, cost):
    """The 4K configurations in the SAD / SSD cost mode (PARITY UNPINNED: the build's own CPU
    definition is the only oracle), EVERY pixel of web and best: the definition run on row bands
    with a window halo, in threads."""
    w, h, d, sw, mode = CONFIGS[cfg]
    left, right = make_pair(w, h, d, seed=3)
    plan = hip.StereoPlan(w, h, d, sw, mode)
    web, best = plan.cost_wta(dev(left), dev(right), cost)
    ob, ow = oracle.cost_hot_path_banded(left, right, d, sw, mode, cost)
    assert np.array_equal(host(web)[0], ow), (cfg, cost)
    assert np.array_equal(host(best)[0], ob), (cfg, cost)
    plan.close()


def test_cost_mode_recovers_a_known_shift(hip):
    # right(x + 7) = left(x): SAD is zero exactly at shift 7 -> web = 8 away from flat regions
    rng = np.random.default_rng(2)
    left = rng.integers(0, 256, (60, 160), dtype=np.uint8)
    right = np.roll(left, 7, 1)
    plan = hip.StereoPlan(160, 60, 32, 5)
    web, best = plan.cost_wta(dev(left), dev(right), "sad")
    assert (host(web)[0] == 8).all() and (host(best)[0] == 0).all()
    plan.close()


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
def test_batched_run_from_gray_images(hip, mode):
    """sm_run on a batch: edge detection and matching of every pair of the batch in
    single launches, partial batches of a larger plan included"""
    w, h, d, sw, n = 132, 75, 30, 9, 4
    pairs = [make_pair(w, h, d, seed=60 + i, kind="noise" if i == 2 else "scene") for i in range(n)]
    left = np.stack([p[0] for p in pairs]); right = np.stack([p[1] for p in pairs])
    plan = hip.StereoPlan(w, h, d, sw, mode, max_pairs=6)
    web, best = plan.run(dev(left), dev(right), 0.2, want_best=True)
    el, er = plan.find_all_edges(dev(left), dev(right), 0.2)
    for i in range(n):
        o = oracle.pipeline(pairs[i][0], pairs[i][1], 0.2, d, sw, mode=mode, step3=False)
        assert np.array_equal(host(web)[i], o["web-1"]), i
        assert np.array_equal(host(best)[i], o["score_best-0"]), i
        assert np.array_equal(host(el)[i], o["edges-1"]) and np.array_equal(host(er)[i], o["edges-2"]), i
    # (sm_plan_prepare_threshold in advance changes nothing but when the tables are built)
    plan.prepare_threshold(0.3)
    # a different threshold on the same plan rebuilds the decision tables
    web2, _ = plan.run(dev(left[:1]), dev(right[:1]), 0.05)
    assert np.array_equal(host(web2)[0], oracle.pipeline(pairs[0][0], pairs[0][1], 0.05, d, sw, mode=mode,
                                                          step3=False)["web-1"])
    plan.close()


def _random_geometries(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        w = int(rng.integers(1, 330)); h = int(rng.integers(1, 200))
        sw = int(rng.integers(0, min(26, w, h) + 1))
        d = int(rng.choice([1, 2, 7, 16, 17, 30, 33, 64, 100, 128, 200, 255, 300]))
        mode = "ghost" if rng.integers(0, 2) else "toroidal"
        dens = float(rng.choice([0.05, 0.3, 0.5, 0.9]))
        out.append((w, h, d, sw, mode, dens))
    return out


@pytest.mark.parametrize("w,h,d,sw,mode,dens", _random_geometries(70, seed=2026))
def test_hot_path_random_geometries(hip, w, h, d, sw, mode, dens):
    """seeded sweep over the geometry space: every size class, shift count, window and
    border the plan logic can be given (all four kernel families are hit)"""
    le, re = rand_edges(w, h, seed=w * 1000 + h, density=dens)
    best, web, desc = hip_hot_path(hip, le, re, d, sw, mode)
    obest, oweb = oracle.hot_path(le, re, d, sw, mode)
    assert np.array_equal(web[0], oweb), desc
    assert np.array_equal(best[0], obest), desc


@pytest.mark.skipif(not os.environ.get("SM_SOAK"), reason="soak run: SM_SOAK=<cases> [SM_SOAK_SEED=<seed>]")
def test_soak_random_pipelines(hip):
    """One-off soak (not part of the default suite): SM_SOAK random (size, shifts, window, border,
    threshold, batch) draws, gray images in, edges + web + best compared with the oracle; sizes reach
    past 256 columns so that every block shape of the edge kernels (interior / border waves, stacked
    or side by side) and every kernel family of the plan logic is drawn."""
    n = int(os.environ["SM_SOAK"])
    rng = np.random.default_rng(int(os.environ.get("SM_SOAK_SEED", "1")))
    seen = {}
    for case in range(n):
        w = int(rng.choice([rng.integers(1, 64), rng.integers(64, 330), rng.integers(330, 900)]))
        if rng.integers(0, 3) == 0:
            w = max(4, w // 4 * 4)
        h = int(rng.integers(1, 120))
        sw = int(rng.integers(0, min(26, w, h) + 1))
        d = int(rng.choice([1, 2, 7, 16, 17, 30, 33, 64, 100, 128, 200, 255, 300]))
        mode = "ghost" if rng.integers(0, 2) else "toroidal"
        thr = float(rng.choice([0.0, 0.02, 0.15, 0.15, 0.4, 1.0]))
        pairs = int(rng.choice([1, 1, 2, 3]))
        kind = str(rng.choice(["scene", "scene", "noise"]))
        imgs = [make_pair(w, h, d, seed=int(rng.integers(1 << 30)), kind=kind) for _ in range(pairs)]
        left = np.stack([p[0] for p in imgs]); right = np.stack([p[1] for p in imgs])
        plan = hip.StereoPlan(w, h, d, sw, mode, max_pairs=pairs)
        pipelined = int(rng.integers(0, 3))              # 0 off, 1 on, 2 on + inputs ordered behind the stream
        if pipelined:
            # the two internal lanes under the soak -- consecutive calls with alternating result
            # maps, a threshold change in between (new decision tables: a sequential phase) and a narrow map;
            # the LAST call is the one compared below, the others must not disturb it
            plan.set_pipelined(pipelined)
            other = float(rng.choice([0.0, 0.3, 1.0]))
            dl, dr = dev(left), dev(right)
            nd = torch.uint8 if d <= 255 and rng.integers(0, 2) else torch.uint16
            maps = [plan._new(pairs, torch.int32) for _ in range(2)]
            for q in range(int(rng.integers(2, 6))):
                plan.run(dl, dr, other if q == 1 else thr, want_best=False, web=maps[q & 1])
            narrow, _ = plan.run(dl, dr, thr, want_best=False, web_dtype=nd)
            web, best = plan.run(dl, dr, thr, want_best=True)
            torch.cuda.synchronize()
            assert torch.equal(narrow.to(torch.int32), web), (case, "narrow map of the pipelined runs", plan.describe())
            if case % 3 == 0:
                # round 5: the same lanes INSIDE a HIP graph (forked and joined by the capture's own events), replayed
                # twice with an eager call in between
                plan.reserve_narrow()
                gw = [plan._new(pairs, torch.int32) for _ in range(3)]
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    for o in gw:
                        plan.run(dl, dr, thr, want_best=False, web=o)
                for rep in range(2):
                    for o in gw:
                        o.zero_()
                    g.replay()
                    torch.cuda.synchronize()
                    for o in gw:
                        assert torch.equal(o, web), (case, "graph replay of pipelined runs", rep, plan.describe())
                    plan.run(dl, dr, thr, want_best=False, web=maps[0])
                del g
            plan.set_pipelined(0)
        elif case % 4 == 1:
            # round 5: sm_run_after behind an upload on a copy stream (its event the only input dependency), three calls
            # into three maps, then the compared one
            dl, dr = torch.zeros_like(dev(left)), torch.zeros_like(dev(right))
            hl, hr = torch.from_numpy(left).pin_memory(), torch.from_numpy(right).pin_memory()
            copy = torch.cuda.Stream()
            torch.cuda.synchronize()
            with torch.cuda.stream(copy):
                dl.copy_(hl, non_blocking=True); dr.copy_(hr, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(copy)
            extra = [plan.run_after(dl, dr, thr, inputs_ready=ev)[0] for _ in range(3)]
            web, best = plan.run_after(dl, dr, thr, inputs_ready=ev, want_best=True)
            torch.cuda.synchronize()
            for o in extra:
                assert torch.equal(o, web), (case, "sm_run_after", plan.describe())
        else:
            web, best = plan.run(dev(left), dev(right), thr, want_best=True)
        el, er = plan.find_all_edges(dev(left), dev(right), thr)
        what = (case, w, h, d, sw, mode, thr, pairs, kind, pipelined, plan.describe())
        for i in range(pairs):
            o = oracle.pipeline(imgs[i][0], imgs[i][1], thr, d, sw, mode=mode, step3=False)
            assert np.array_equal(host(el)[i], o["edges-1"]) and np.array_equal(host(er)[i], o["edges-2"]), what
            assert np.array_equal(host(web)[i], o["web-1"]), what
            assert np.array_equal(host(best)[i], o["score_best-0"]), what
        fam = plan.describe().split(":")[0].split("(")[0].strip()
        seen[fam] = seen.get(fam, 0) + 1
        plan.close()
    print(f"soak: {n} cases, all equal to the oracle; kernel families drawn: {seen}")


# ---------------------------------------------------------------------------
# robustness of the launch geometry
# ---------------------------------------------------------------------------

def edges4_read_columns(w, g):
    """Host restatement of k_edges_ext4's toroidal address arithmetic: for every lane of
    the launch grid, the first source column of its dword load (xq) and of its single
    neighbour byte (xn).  Mirrors csrc/sm_api.hip; the kernel must never read a column
    outside [0, w)."""
    ext_px = g["ext_words"] * 32
    row = g["ext_words"] * 8
    stacked = (row + 255) // 256 * 256 > row + row // 32       # the host's choice of block shape
    lanes = (row + 63) // 64 * 64 if stacked else (row + 255) // 256 * 256
    lane = np.arange(lanes)
    xe = lane * 4
    x = xe - g["pad_l"]
    in_ext = xe < ext_px
    over = ext_px - g["pad_l"] - w
    if w >= g["pad_l"] and w >= over:
        xq = np.where(x < 0, x + w, np.where(x >= w, x - w, x))
    else:
        xq = np.mod(x, w)
    xq = np.where(in_ext, xq, 0)
    last = (lane & 63) == 63
    xn = np.where(last, np.where(xq + 4 == w, 0, xq + 4), np.where(xq == 0, w - 1, xq - 1))
    return xq, xn


@pytest.mark.parametrize("w,h,d,sw", [(640, 480, 30, 21), (1000, 64, 30, 21), (400, 64, 128, 9),
                                      (240, 135, 30, 21), (64, 48, 16, 5), (3840, 64, 128, 9),
                                      (1920, 64, 64, 7), (132, 40, 500, 3), (36, 36, 200, 25)])
def test_edge_kernel_never_reads_outside_a_row(hip, w, h, d, sw):
    plan = hip.StereoPlan(w, h, d, sw, "toroidal")
    xq, xn = edges4_read_columns(w, plan.geometry())
    assert xq.min() >= 0 and xq.max() + 4 <= w, (w, d, sw, int(xq.max()))
    assert xn.min() >= 0 and xn.max() < w
    # and the kernel's results are right on an image that ENDS at an allocation boundary
    left, right = make_pair(w, h, d, seed=5)
    el, er = plan.find_all_edges(dev(left), dev(right), 0.15)
    assert np.array_equal(host(el)[0], oracle.find_all_edges(left, 0.15, "toroidal"))
    assert np.array_equal(host(er)[0], oracle.find_all_edges(right, 0.15, "toroidal"))
    plan.close()


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
def test_unaligned_buffers_take_the_scalar_paths(hip, mode):
    """Inputs that are not 4-byte aligned and maps that are not 16-byte aligned must give
    the same results (the dword / int4 fast paths step aside)."""
    w, h, d, sw = 256, 96, 64, 7
    left, right = make_pair(w, h, d, seed=6)
    plan = hip.StereoPlan(w, h, d, sw, mode)
    n = w * h
    raw_l = torch.empty(n + 8, dtype=torch.uint8, device="cuda")
    raw_r = torch.empty(n + 8, dtype=torch.uint8, device="cuda")
    l_un = raw_l[1:1 + n].view(h, w); l_un.copy_(dev(left))
    r_un = raw_r[3:3 + n].view(h, w); r_un.copy_(dev(right))
    assert l_un.data_ptr() % 4 and r_un.data_ptr() % 4
    raw_w = torch.empty(n + 8, dtype=torch.int32, device="cuda")
    raw_b = torch.empty(n + 8, dtype=torch.int32, device="cuda")
    web_un = raw_w[1:1 + n].view(1, h, w)
    best_un = raw_b[2:2 + n].view(1, h, w)
    assert web_un.data_ptr() % 16 and best_un.data_ptr() % 16
    import ctypes as C
    from stereomatching_amd.capi import check, lib
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.sm_run(plan._h, C.c_void_p(l_un.data_ptr()), C.c_void_p(r_un.data_ptr()), 0.15, 1,
                     C.c_void_p(web_un.data_ptr()), C.c_void_p(best_un.data_ptr()), st))
    torch.cuda.synchronize()
    oel = oracle.find_all_edges(left, 0.15, mode)
    oer = oracle.find_all_edges(right, 0.15, mode)
    ob, ow = oracle.hot_path(oel, oer, d, sw, mode)
    assert np.array_equal(host(web_un)[0], ow) and np.array_equal(host(best_un)[0], ob)
    plan.close()


def test_result_buffers_are_validated(hip):
    plan = hip.StereoPlan(64, 48, 16, 5, "toroidal", max_pairs=2)
    le, re = rand_edges(64, 48, 1)
    plan.load_edges(dev(np.stack([le, le])), dev(np.stack([re, re])))
    with pytest.raises(ValueError):
        plan.match_wta(2, web=torch.empty((2, 48, 64), dtype=torch.int64, device="cuda"))
    with pytest.raises(ValueError):
        plan.match_wta(2, web=torch.empty((2, 48, 60), dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        plan.match_wta(2, web=torch.empty((1, 48, 64), dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        plan.match_wta(2, web=torch.empty((2, 48, 64), dtype=torch.int32))          # host tensor
    plan.close()


def test_plan_on_another_device(hip):
    """sm_plan_create(device = k > 0): the same parity check on the last visible device."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device")
    k = torch.cuda.device_count() - 1
    w, h, d, sw = 320, 96, 128, 9
    left, right = make_pair(w, h, d, seed=7)
    plan = hip.StereoPlan(w, h, d, sw, "toroidal", device=k)
    with torch.cuda.device(k):
        web, best = plan.run(torch.from_numpy(left).cuda(k), torch.from_numpy(right).cuda(k), 0.15,
                             want_best=True)
        torch.cuda.synchronize(k)
    oel = oracle.find_all_edges(left, 0.15, "toroidal")
    oer = oracle.find_all_edges(right, 0.15, "toroidal")
    ob, ow = oracle.hot_path(oel, oer, d, sw, "toroidal")
    assert np.array_equal(host(web)[0], ow) and np.array_equal(host(best)[0], ob)
    plan.close()


def test_bench_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` by itself: two child ranks with real kernels, both on
    device 0 over gloo (SM_BENCH_REHEARSAL; the driver's 8-GPU run uses RCCL)."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SM_BENCH_REHEARSAL"] = "1"
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--config", "C4",
                        "--pairs", "2", "--steps", "6", "--warmup", "2", "--gather"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["roofline"]["bound"] == "valu"
    assert out["roofline"]["kernel_launches_timed"] >= 6
    assert "gather_ms" in out
    # SURVEY 8e's scaling workload rides along at N > 1: 64 x 1080p pairs, 32 per rank here, each share
    # in one launch per step, then the 64 maps collected on rank 0 in pair order
    c4 = out["c4"]
    assert c4["total_pairs"] == 64 and c4["pairs_per_rank"] == 32 and c4["value"] > 0
    assert c4["gather_ms"] > 0 and c4["maps_in_pair_order"] is True


# ---------------------------------------------------------------------------
# narrow web maps (sm_match_wta_typed / sm_run_typed) and the fused step 3
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("w,h,d,sw,dtype", [
    (300, 150, 128, 9, torch.uint8),      # bit-sliced kernel, dword stores of 4 pixels
    (300, 150, 128, 9, torch.uint16),
    (257, 61, 64, 7, torch.uint8),        # width not a multiple of 4: scalar stores
    (130, 70, 300, 5, torch.uint16),      # more shifts than a byte holds
    (120, 80, 30, 23, torch.uint8),       # popcount kernel: int32 map + narrowing pass
    (60, 50, 30, 27, torch.uint16),       # generic kernel
])
def test_narrow_web_maps_equal_the_int32_map(hip, mode, w, h, d, sw, dtype):
    le, re = rand_edges(w, h, seed=w + d)
    plan = hip.StereoPlan(w, h, d, sw, mode)
    plan.load_edges(dev(le), dev(re))
    web32, best32 = plan.match_wta(1, want_best=True)
    webn, bestn = plan.match_wta(1, want_best=True, web_dtype=dtype)
    torch.cuda.synchronize()
    _, oweb = oracle.hot_path(le, re, d, sw, mode)
    assert np.array_equal(host(web32)[0], oweb), plan.describe()
    assert np.array_equal(host(webn.to(torch.int32))[0], oweb), plan.describe()
    assert torch.equal(best32, bestn)
    plan.close()


def test_narrow_staging_map_is_allocated_on_demand(hip):
    """a plan of the fallback kernels does not carry the int32 staging map of the narrow
    results until someone asks for it (sm_plan_reserve_narrow, or the first narrow request)."""
    import ctypes as C
    w, h, d, sw = 120, 80, 30, 23                      # popcount kernel
    le, re = rand_edges(w, h, seed=5)
    _, oweb = oracle.hot_path(le, re, d, sw, "toroidal")
    plan = hip.StereoPlan(w, h, d, sw, "toroidal", max_pairs=2)
    assert "bit-sliced" not in plan.describe()
    base = plan.workspace_bytes()
    plan.load_edges(dev(le), dev(re))
    web32, _ = plan.match_wta(1, want_best=False)
    assert plan.workspace_bytes() == base                # int32 results: nothing allocated
    web8, _ = plan.match_wta(1, want_best=False, web_dtype=torch.uint8)
    torch.cuda.synchronize()
    assert plan.workspace_bytes() == base + 2 * w * h * 4
    assert np.array_equal(host(web32)[0], oweb) and np.array_equal(host(web8.to(torch.int32))[0], oweb)
    plan.close()
    plan = hip.StereoPlan(w, h, d, sw, "toroidal")
    base = plan.workspace_bytes()
    hip.capi.check(hip.capi.lib.sm_plan_reserve_narrow(plan._h))
    hip.capi.check(hip.capi.lib.sm_plan_reserve_narrow(plan._h))      # idempotent
    assert plan.workspace_bytes() == base + w * h * 4
    plan.close()
    plan = hip.StereoPlan(300, 150, 128, 9, "toroidal")               # bit-sliced: stores narrow maps itself
    base = plan.workspace_bytes()
    hip.capi.check(hip.capi.lib.sm_plan_reserve_narrow(plan._h))
    assert plan.workspace_bytes() == base
    plan.close()


def test_plan_options_struct_size_rules(hip):
    """an all-zero options struct is valid (struct_size 0 = no field set); a struct LONGER
    than this library's (a caller built against a newer header) is accepted, its known prefix taken;
    absurd cost-mode tile heights are clamped instead of failing at launch."""
    import ctypes as C
    lib, vp = hip.capi.lib, C.c_void_p

    class Bigger(C.Structure):
        _fields_ = [("known", hip.capi.PlanOptions), ("future_field", C.c_int * 8)]

    zero = hip.capi.PlanOptions()
    assert zero.struct_size == 0
    h_ = vp()
    hip.capi.check(lib.sm_plan_create_ex(0, 300, 150, 128, 9, 0, 1, C.byref(zero), C.byref(h_)))
    ref_desc = lib.sm_plan_describe(h_).decode()
    lib.sm_plan_destroy(h_)
    big = Bigger()
    big.known.struct_size = C.sizeof(Bigger)
    big.known.kernel_family = 1
    for i in range(8):
        big.future_field[i] = -1
    h_ = vp()
    hip.capi.check(lib.sm_plan_create_ex(0, 300, 150, 128, 9, 0, 1, C.cast(C.byref(big), C.POINTER(hip.capi.PlanOptions)),
                                         C.byref(h_)))
    assert "tiled kernel" in lib.sm_plan_describe(h_).decode() and "bit-sliced" in ref_desc
    lib.sm_plan_destroy(h_)
    neg = hip.capi.PlanOptions()
    neg.struct_size = -4
    h_ = vp()
    assert lib.sm_plan_create_ex(0, 300, 150, 128, 9, 0, 1, C.byref(neg), C.byref(h_)) != 0
    # cost-mode options out of range: clamped / ignored, results unchanged
    left, right = make_pair(200, 50, 64, seed=3)
    for cost in ("sad", "ssd"):
        ob, ow = oracle.cost_hot_path(left, right, 64, 9, "toroidal", cost)
        for opts in (dict(cost_tile_h=100000), dict(cost_pixels_per_lane=7), dict(cost_pixels_per_lane=-3, cost_tile_h=1)):
            plan = hip.StereoPlan(200, 50, 64, 9, "toroidal", options=opts)
            web, best = plan.cost_wta(dev(left), dev(right), cost)
            torch.cuda.synchronize()
            assert np.array_equal(host(web)[0], ow) and np.array_equal(host(best)[0], ob), (cost, opts)
            plan.close()


@pytest.mark.parametrize("mode", ["toroidal", "ghost"])
@pytest.mark.parametrize("w,h,d,sw,opts", [(300, 150, 128, 9, None), (1920, 64, 64, 7, None), (257, 61, 64, 7, None),
                                          (300, 150, 128, 9, dict(edge_kernel=1)), (120, 80, 30, 23, None)])
def test_find_edges_after_load_edges_on_one_plan(hip, mode, w, h, d, sw, opts):
    """sm_load_edges writes EVERY column of the packed image (wrapped content included), the
    edge kernels only the columns a valid output pixel can reach; the columns beyond keep what the load
    left there.  No stored pixel may depend on them: load random edges, then run the gray-image pipeline
    on the same plan and compare with the oracle -- and the other way round."""
    left, right = make_pair(w, h, d, seed=w + h)
    le, re = rand_edges(w, h, seed=9, density=0.6)
    o = oracle.pipeline(left, right, 0.15, d, sw, mode=mode, step3=False)
    ob, ow = oracle.hot_path(le, re, d, sw, mode)
    plan = hip.StereoPlan(w, h, d, sw, mode, options=opts)
    plan.load_edges(dev(le), dev(re))
    web, best = plan.match_wta(1, want_best=True)
    torch.cuda.synchronize()
    assert np.array_equal(host(web)[0], ow) and np.array_equal(host(best)[0], ob), plan.describe()
    web, best = plan.run(dev(left), dev(right), 0.15, want_best=True)          # find_edges over the loaded image
    torch.cuda.synchronize()
    assert np.array_equal(host(web)[0], o["web-1"]), plan.describe()
    assert np.array_equal(host(best)[0], o["score_best-0"]), plan.describe()
    plan.load_edges(dev(le), dev(re))                                           # and back
    web, best = plan.match_wta(1, want_best=True)
    torch.cuda.synchronize()
    assert np.array_equal(host(web)[0], ow) and np.array_equal(host(best)[0], ob), plan.describe()
    plan.close()


def test_narrow_web_rejects_too_many_shifts(hip):
    plan = hip.StereoPlan(64, 48, 300, 5, "toroidal")
    le, re = rand_edges(64, 48, 2)
    plan.load_edges(dev(le), dev(re))
    with pytest.raises(hip.capi.StereoHipError):
        plan.match_wta(1, want_best=False, web_dtype=torch.uint8)
    plan.close()


@pytest.mark.parametrize("times,lines", [(32, 10), (3, 4), (0, 7)])
def test_step3_in_one_sync_equals_the_staged_calls(hip, times, lines):
    w, h, d, sw = 160, 90, 30, 9
    left, right = make_pair(w, h, d, seed=21)
    plan = hip.StereoPlan(w, h, d, sw, "toroidal", max_pairs=2)
    web, _ = plan.run(dev(np.stack([left, left])), dev(np.stack([right, right])), 0.15)
    # (a) a web from the hot path: no zero pixel, hole filling is the identity
    filled, out, mm = plan.step3(web, times, lines)
    want_f = plan.fill_web_holes(web, times)
    want_o = plan.draw_contour_map(want_f, lines)
    assert torch.equal(filled, want_f) and torch.equal(out, want_o)
    assert torch.equal(mm, plan.image_min_max(want_f))
    # (b) a web WITH holes takes the staged route inside sm_step3
    holes = web.clone()
    holes[:, 10:30, 20:60] = 0
    holes[:, 0, :] = 0
    filled, out, mm = plan.step3(holes, times, lines)
    want_f = plan.fill_web_holes(holes, times)
    want_o = plan.draw_contour_map(want_f, lines)
    assert torch.equal(filled, want_f) and torch.equal(out, want_o)
    for j in range(2):
        assert np.array_equal(host(filled)[j], oracle.fill_web_holes(host(holes)[j], times))
        assert np.array_equal(host(out)[j], oracle.draw_contour_map(host(filled)[j], lines))
    plan.close()


def test_step3_reports_a_zero_interval(hip):
    plan = hip.StereoPlan(64, 48, 30, 5, "toroidal")
    web = torch.full((1, 48, 64), 7, dtype=torch.int32, device="cuda")
    with pytest.raises(hip.capi.StereoHipError) as e:
        plan.step3(web, 4, 10)
    assert e.value.code == hip.capi.SM_ERR_ZERO_DIV
    plan.step3(torch.arange(48 * 64, dtype=torch.int32, device="cuda").view(1, 48, 64) + 1, 4, 10)  # flag cleared
    plan.close()


def test_bench_rccl_calls_on_one_rank():
    """The torch.distributed calls of the N > 1 bench path (object broadcast, barrier with
    device ids, all-reduce MAX of the timing, collection on rank 0) on the real backend
    -- nccl = RCCL -- with a one-rank group: all a one-GPU box can run of it."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SM_BENCH_NCCL_SELFTEST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--config", "C2", "--steps", "6",
                        "--warmup", "2", "--gather", "--no-cpu-baseline", "--c4", "--no-cost-modes", "--no-e2e"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")][0]
    assert out["n_gpus"] == 1 and out["value"] > 0
    # --c4 on one rank over the real backend: all 64 pairs in one launch, the collection a no-op
    assert out["c4"]["pairs_per_rank"] == 64 and out["c4"]["value"] > 0 and out["c4"]["maps_in_pair_order"] is True


def test_c_abi_rccl_collection_on_one_rank(hip):
    """sm_comm_create / sm_broadcast / sm_gather_maps (RCCL behind the C ABI: what a C host collects its
    maps with when they are to stay on one GPU).  All a one-GPU box can run of it: a one-rank
    communicator -- librccl.so is loaded, ncclCommInitAll and a grouped ncclBroadcast run, the root's own
    share of the collection is copied in rank order -- and the argument checks in front of RCCL."""
    import ctypes as C
    lib, check, vp = hip.capi.lib, hip.capi.check, C.c_void_p
    comm = vp()
    devs = (C.c_int * 2)(0, 0)
    assert lib.sm_comm_create(devs, 2, C.byref(comm)) == hip.capi.SM_ERR_ARG        # one rank per device
    assert b"twice" in lib.sm_last_error()
    assert lib.sm_comm_create((C.c_int * 1)(99), 1, C.byref(comm)) == hip.capi.SM_ERR_ARG
    check(lib.sm_comm_create(devs, 1, C.byref(comm)))
    assert lib.sm_comm_size(comm) == 1
    w, h, d, sw = 320, 200, 64, 7
    left, right = make_pair(w, h, d, seed=77)
    plan = hip.StereoPlan(w, h, d, sw, "toroidal", max_pairs=2)
    web, _ = plan.run(dev(np.stack([left, right])), dev(np.stack([right, left])), 0.15)
    torch.cuda.synchronize()
    params = torch.arange(64, dtype=torch.uint8, device="cuda")
    bufs = (vp * 1)(params.data_ptr())
    check(lib.sm_broadcast(comm, bufs, 64, None))
    gathered = torch.zeros_like(web)
    srcs = (vp * 1)(web.data_ptr())
    sizes = (C.c_size_t * 1)(web.numel() * 4)
    stream = (vp * 1)(torch.cuda.current_stream().cuda_stream)
    check(lib.sm_gather_maps(comm, srcs, sizes, vp(gathered.data_ptr()), stream))
    torch.cuda.synchronize()
    assert torch.equal(gathered, web) and torch.equal(params.cpu(), torch.arange(64, dtype=torch.uint8))
    o = oracle.pipeline(left, right, 0.15, d, sw, mode="toroidal", step3=False)
    assert np.array_equal(host(gathered)[0], o["web-1"])
    assert lib.sm_gather_maps(comm, srcs, sizes, None, None) == hip.capi.SM_ERR_ARG
    lib.sm_comm_destroy(comm)
    lib.sm_comm_destroy(None)
    plan.close()


@pytest.mark.parametrize("flags,graph", [([], True), (["--no-graph"], False)])
def test_bench_line_with_and_without_graph_replay(flags, graph):
    """bench.py's two ways of issuing its steps -- replayed from HIP graphs (default; the kernel time is sampled
    right behind the timed region) and launched from the host (--no-graph; the match launches of the timed
    region carry their own events) -- give a complete line each, and about the same rate."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "24", "--warmup", "3", "--no-cpu-baseline",
                        "--no-e2e", "--no-cost-modes", *flags], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")][0]
    assert out["steps"] == 24 and out["value"] > 1e6 and out["roofline"]["bound"] == "valu"
    assert bool(out["config"]["graph"]) == graph, out["config"]["graph"]
    assert out["roofline"]["kernel_launches_timed"] >= 10
    assert 0.3 < out["roofline"]["frac"] < 1.0 and out["roofline"]["kernel_ms"] < out["ms_per_step"]
    assert out["verified"] is True and out["verification"]["maps_equal_host_launched_runs"] is True
    assert out["graph_capture"] == ("ok" if graph else "not asked for")
    if graph:
        assert out["host_launched"]["ms_per_step"] > 0 and out["c2"]["verified"] is True and out["c2"]["value"] > 1e6
        assert out["overlapped"]["verified"] is True and out["overlapped"]["value"] > 1e6
        assert out["c2"]["overlapped"]["verified"] is True


@pytest.mark.parametrize("flags", [[], ["--overlap"]])
def test_bench_checks_what_it_timed(flags):
    """bench.py compares the result maps its timed steps left (graph replays by default; with --pipeline: consecutive
    steps overlapped INSIDE the graph) with host-launched runs on the device and with the CPU oracle on a band, says
    `verified` in its line and exits non-zero otherwise: a map corrupted behind the timed region (a test hook) must
    be noticed."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    base = [sys.executable, str(root / "bench.py"), "--config", "C2", "--steps", "8", "--warmup", "2", "--no-e2e",
            "--no-cost-modes", "--cpu-rows", "12", *flags]
    p = subprocess.run(base, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")][0]
    assert out["verified"] is True and out["verification"]["band_equals_cpu_oracle"] is True
    assert out["graph_capture"] == "ok" and out["cpu_baseline"]["value"] > 0
    ac = out["cpu_baseline"]["all_cores"]
    assert ac["value"] > 0 and 1 <= ac["cores"] <= ac["usable_cores"] and str(ac["cores"]) in ac["rate_by_threads"]
    p = subprocess.run(base + ["--corrupt-map", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    out = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")][0]
    assert out["verified"] is False and "VERIFICATION FAILED" in p.stderr
