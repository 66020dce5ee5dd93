#!/usr/bin/env python3
"""Per-wave timeline of one match launch (diagnostic build with -DSM_STAMPS).

    python -c "from stereomatching_amd import build; build.build_diag(True)"   # build box
    gpurun -- 'python3 tools/wave_timeline.py C3 1 > gpurun_out/timeline_C3.txt'

Every wave stamps start / staged / warmed-up / end with the constant 100 MHz
counter (s_memrealtime) and the shader clock (s_memtime) plus its HW_ID / XCC_ID.
Prints: launch span, when waves start and end, wave lifetime distribution, waves per
SIMD, the phase split, and the shader clock the waves saw.
"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["SM_HIP_LIB"] = str(ROOT / "tools" / "diag" /
                               f"libstereo_hip_{os.environ.get('SM_DIAG', 'stamps')}.so")

import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from stereomatching_amd import capi, pipeline  # noqa: E402
from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w, h, d, sw, mode = CONFIGS[cfg]
plan = pipeline.StereoPlan(w, h, d, sw, mode, max_pairs=pairs, options=__import__('tools._options', fromlist=['from_env']).from_env() or None)
g = plan.geometry()
print(plan.describe())
ls, rs = zip(*[make_pair(w, h, d, seed=j) for j in range(pairs)])
left, right = torch.from_numpy(np.stack(ls)).cuda(), torch.from_numpy(np.stack(rs)).cuda()
plan.find_all_edges(left, right, want_edges=False)
n_wg = g["tiles_x"] * g["tiles_y"] * pairs * (g["waves_per_workgroup"] if g["kernel"] == 4 else 1)   # records: one per wave
stamps = torch.zeros((n_wg, 10), dtype=torch.int64, device="cuda")
web = None
for _ in range(5):                       # warm: clocks, code object
    web, _ = plan.match_wta(pairs, want_best=False, web=web)
torch.cuda.synchronize()
capi.lib.sm_debug_set_stamps.argtypes = [C.c_void_p]
assert capi.lib.sm_debug_set_stamps(C.c_void_p(stamps.data_ptr())) == 0
for rep in range(3):
    stamps.zero_()
    torch.cuda.synchronize()
    # back to back with a preceding launch of the same kernel, as in a loop of match launches -- or (WT_STEP=1)
    # behind the edge kernel, as in the real step: the waves then land in other slots
    plan.match_wta(pairs, want_best=False, web=web)
    if os.environ.get("WT_STEP"):
        for _ in range(20):                     # a run of real steps in front, then the stamped one
            plan.find_all_edges(left, right, want_edges=False)
            plan.match_wta(pairs, want_best=False, web=web)
        plan.find_all_edges(left, right, want_edges=False)
        stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plan.match_wta(pairs, want_best=False, web=web)
    e1.record()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.int64)
    rt = s[:, 0:8:2].astype(np.float64) * 0.01          # us (100 MHz)
    ck = s[:, 1:8:2].astype(np.float64)
    hw = s[:, 8]
    t0 = rt[:, 0].min()
    rt -= t0
    life = rt[:, 3] - rt[:, 0]
    print(f"--- rep {rep}: event time {e0.elapsed_time(e1) * 1e3:.1f} us; first wave start -> last wave "
          f"end {rt[:, 3].max():.1f} us; waves {n_wg}")
    q = [0, 1, 10, 50, 90, 99, 100]
    print("  wave start  (us) pct", q, np.percentile(rt[:, 0], q).round(1))
    print("  wave end    (us) pct", q, np.percentile(rt[:, 3], q).round(1))
    print("  lifetime    (us) pct", q, np.percentile(life, q).round(1))
    print("  stage / warm-up / rows (us, median):", np.median(rt[:, 1] - rt[:, 0]).round(2),
          np.median(rt[:, 2] - rt[:, 1]).round(2), np.median(rt[:, 3] - rt[:, 2]).round(2))
    clk = (ck[:, 3] - ck[:, 0]) / np.maximum(life, 1e-9) / 1e3
    print("  shader clock seen by waves (GHz) pct", q, np.percentile(clk, q).round(3))
    hwid = hw & 0xffffffff
    xcc = (hw >> 32) & 0xf
    simd = (hwid >> 4) & 3
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 1
    se = (hwid >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simd_key = key * 4 + simd
    uniq, cnt = np.unique(simd_key, return_counts=True)
    print(f"  distinct CUs {len(np.unique(key))}, distinct SIMDs {len(uniq)}; waves per SIMD histogram:",
          dict(zip(*np.unique(cnt, return_counts=True))))
    slot = hwid & 0xf
    print("  hardware wave slots in use:", dict(zip(*np.unique(slot, return_counts=True))))
    # the two waves of a SIMD: do they finish together?
    order = np.argsort(simd_key, kind="stable")
    sk, lf, st = simd_key[order], life[order], rt[order, 0]
    pair = np.flatnonzero(sk[:-1] == sk[1:])
    pair = pair[np.r_[True, np.diff(pair) > 1]] if len(pair) else pair
    if len(pair):
        a, b = lf[pair], lf[pair + 1]
        fast, slow = np.minimum(a, b), np.maximum(a, b)
        first_is_fast = np.where(st[pair] <= st[pair + 1], a <= b, b <= a)
        print(f"  SIMD pairs {len(pair)}: faster wave median {np.median(fast):.1f} us, slower wave median "
              f"{np.median(slow):.1f} us; pairs whose lifetimes differ by > 10 us: {int((slow - fast > 10).sum())}; "
              f"earlier-started wave is the faster one in {100 * first_is_fast.mean():.0f} %")
        print("  slower - faster (us) pct", q, np.percentile(slow - fast, q).round(1))
        # phases of the earlier-started ("older") and the later-started wave of each pair
        rto, cko = rt[order], ck[order]
        first = np.where(st[pair] <= st[pair + 1], pair, pair + 1)
        second = np.where(st[pair] <= st[pair + 1], pair + 1, pair)
        for name, idx in (("older", first), ("younger", second)):
            ph_us = [np.median(rto[idx, k + 1] - rto[idx, k]) for k in range(3)]
            ph_cy = [np.median(cko[idx, k + 1] - cko[idx, k]) for k in range(3)]
            print(f"    {name:8s} wave: stage {ph_us[0]:6.2f} us / warm-up {ph_us[1]:6.2f} us ({ph_cy[1]:9.0f} cyc)"
                  f" / rows {ph_us[2]:6.2f} us ({ph_cy[2]:9.0f} cyc)")
    # lifetime by hardware wave slot, and (two-wave workgroups) where a workgroup's waves sit
    ok = rt[:, 3] > 0
    for sl in np.unique(slot):
        m = (slot == sl) & ok
        print(f"    slot {int(sl)}: waves {int(m.sum())}, lifetime median {np.median(life[m]):.1f} us, "
              f"start median {np.median(rt[m, 0]):.2f} us")
    wpw = g["waves_per_workgroup"] if g["kernel"] == 4 else 1
    if wpw == 2:
        s0, s1 = slot[0::2], slot[1::2]
        print("  slots of a workgroup's (wave 0, wave 1):",
              {(int(a_), int(b_)): int(((s0 == a_) & (s1 == b_)).sum()) for a_ in np.unique(s0) for b_ in np.unique(s1)})
        print("  same CU:", int((key[0::2] == key[1::2]).sum()), "of", len(s0),
              " SIMDs (wave 0, wave 1):",
              {(int(a_), int(b_)): int(((simd[0::2] == a_) & (simd[1::2] == b_)).sum()) for a_ in range(4) for b_ in range(4)
               if ((simd[0::2] == a_) & (simd[1::2] == b_)).any()})
        wg_lin = np.arange(len(s0))
        for sl in np.unique(s0):
            m = s0 == sl
            print(f"    workgroups whose wave 0 is in slot {int(sl)}: {int(m.sum())}, linear index pct "
                  f"{np.percentile(wg_lin[m], [0, 10, 50, 90, 100]).round(0)}")
    # threadgroup slot of the wave's workgroup on its CU (HW_ID.TG_ID): do the two waves that share a
    # SIMD always come from workgroups of different halves (slots 0, 1 against 2, 3)?
    tg = (hwid >> 16) & 0xf
    print("  threadgroup ids in use:", dict(zip(*np.unique(tg, return_counts=True))))
    if len(pair):
        tgo = tg[order]
        ta, tb = tgo[pair], tgo[pair + 1]
        print("  SIMD pairs by (tg, tg):", {(int(a_), int(b_)): int(((np.minimum(ta, tb) == a_) & (np.maximum(ta, tb) == b_)).sum())
                                            for a_ in np.unique(tg) for b_ in np.unique(tg)
                                            if ((np.minimum(ta, tb) == a_) & (np.maximum(ta, tb) == b_)).any()})
        print(f"  SIMD pairs whose waves differ in tg bit 1: {int((((ta ^ tb) >> 1) & 1).sum())} of {len(pair)}; "
              f"in tg bit 0: {int(((ta ^ tb) & 1).sum())}; in slot bit 0: "
              f"{int(((slot[order][pair] ^ slot[order][pair + 1]) & 1).sum())}")
    if wpw == 2:
        print("  workgroups whose two waves report the same tg:", int((tg[0::2] == tg[1::2]).sum()), "of", len(tg) // 2)
        for t_ in np.unique(tg):
            m = tg == t_
            print(f"    tg {int(t_)}: waves {int(m.sum())}, warm-up median {np.median((rt[:, 2] - rt[:, 1])[m]):.2f} us, "
                  f"lifetime median {np.median(life[m]):.1f} us, slots {dict(zip(*np.unique(slot[m], return_counts=True)))}")
    if wpw == 2:
        # by dispatch order: workgroups in blocks of one per CU
        lin = np.arange(len(tg)) // 2
        ncu = len(np.unique(key))
        for b in range((lin.max() + ncu) // ncu):
            m = (lin // ncu == b) & ok
            print(f"    workgroups {b * ncu}..{(b + 1) * ncu - 1}: tg {dict(zip(*np.unique(tg[m], return_counts=True)))}, "
                  f"lifetime median {np.median(life[m]):.1f} us (10 % {np.percentile(life[m], 10):.1f}, 90 % {np.percentile(life[m], 90):.1f})")
    single = np.flatnonzero(np.isin(simd_key, uniq[cnt == 1]))
    if len(single):
        print(f"  waves alone on their SIMD: {len(single)}, lifetime median {np.median(life[single]):.1f} us")
    # per CU: the spread of wave end times
    cu_end = {}
    for k_, e_ in zip(key, rt[:, 3]):
        cu_end.setdefault(int(k_), []).append(e_)
    last = np.array([max(v) for v in cu_end.values()])
    print("  last wave end per CU (us) pct", q, np.percentile(last, q).round(1))
    # overlap: how many waves were alive on a SIMD at once (max), and second-round waves
    late = rt[:, 0] > 5.0
    print(f"  waves starting later than 5 us after the first: {int(late.sum())}"
          f" (median start {np.median(rt[late, 0]) if late.any() else 0:.1f} us)")
    for x in range(8):
        m = xcc == x
        if m.any():
            print(f"    XCC {x}: waves {int(m.sum()):5d}  start max {rt[m, 0].max():7.1f}  end max "
                  f"{rt[m, 3].max():7.1f}  lifetime median {np.median(life[m]):7.1f}")
