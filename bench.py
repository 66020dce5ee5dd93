#!/usr/bin/env python3
"""bench.py -- throughput of the stereo hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--pairs B]

One "step" = one pass of the pipeline's data-parallel path over one batch of
synthetic uint8 stereo pairs already resident in HBM: edge detection of both
images, then the fused match-cost / window-sum / winner-take-all launch
(sm_run of include/stereo_hip.h) -> the int32 `web` map in HBM.  That is the
reference's timed region (stages only, inputs resident, no file writes:
/root/reference/src/stereo.cu:308,:334) up to `web`.

Metric (BASELINE.json): Mpixel-disparities/s = W*H*D*pairs / t / 1e6, whole
job over all ranks.

N > 1: one process per GPU, every rank runs the same number of its own pairs
(weak scaling), no data-path collective; barrier + device sync on both sides
of the timed region, max over ranks.  Started either by torchrun (RANK /
LOCAL_RANK / WORLD_SIZE in the environment) or by `python bench.py --gpus N`
alone: the parent then starts the N ranks as CHILD processes before it has
touched torch or the GPU, relays rank 0's JSON line and exits with the worst
return code.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the dominant kernel (k_match_bs) against the roof that binds
                it: integer VALU issue.  frac = VALU wave-instructions of one
                launch / measured launch time / (1024 SIMDs x 2.4 GHz / 2 cycles
                per wave64 instruction).  The byte models of SURVEY.md 8d are
                printed as named extras: hbm_frac_min (compulsory bytes) and
                throughput_bar_frac (the materialised-cost-volume model A_cv, a
                throughput bar, NOT traffic: the fused kernel never moves it).
  cpu_baseline  the oracle's structure-faithful C port timed on this host's
                cores on a bounded band of the same workload (rank 0, N = 1)
  c4            (N > 1, or --c4) SURVEY.md 8e's scaling workload beside the C3 weak-scaling `value`: the batch of
                64 x 1080p pairs (64 shifts, 7x7), pair j -> rank j mod N, each rank's share in ONE launch per
                step, timed under the same barrier / max-over-ranks contract (strong scaling: 64 pairs whatever
                N), plus `gather_ms`: the collection of the 64 maps on rank 0 (RCCL point to point), timed apart
  verified      true iff what was timed is right: after the timed region every resident result map, as the timed steps
                left it, equals (on the device) the map of a host-launched sm_run on the same inputs, and a full-width
                band of the first one equals the CPU oracle (the cpu_baseline leg's own band); the extra objects carry
                checks of their own.  false -> exit code 3.
  overlapped    (N = 1) the same steps issued through sm_run_after -- a step's only input dependency is its resident
                pair -- so that the plan runs consecutive steps on its two lanes; replayed from a graph of its own
  c2            (N = 1) BASELINE's second single-GPU configuration (1080p pair, 64 shifts, 7x7): stream order and overlapped
  host_launched the rate of plain host-launched sm_run calls beside the graph-replayed default
  sad, ssd      (N = 1) the SAD / SSD cost mode -- the cost BASELINE.json's wording names, which
                the reference does not implement: PARITY UNPINNED, the build's own definition --
                at C3 (SAD 9x9) and C5 (SSD 11x11, ghost): ms per launch, Mpixel-disparities/s and
                the VALU issue fraction.  Never part of `value`.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, 2.4 GHz max clock, a wave64 VALU
# instruction issues over 2 cycles -> wave-instructions per second, whole chip
VALU_PEAK_GIPS = 256 * 4 * 2.4 / 2.0 * 1.0      # = 1228.8 G wave-instr/s
MFMA_I8_PEAK_TOPS = 5000.0                       # dense int8 on the matrix cores: 2 x the dense bf16 peak (MI355X_MICROARCH.md: "2x BF16 per clock")
# What the integer VALU of this part SUSTAINS (measured, tools/ubench_sad.hip -> profiles/r03/ubench_sad.txt,
# launches of the match kernel's length): a stream of 15 v_bitop3 + 1 v_alignbit -- the bit-sliced kernel's
# mix -- at the two waves per SIMD its registers allow issues 806 G wave-instr/s chip-wide (a pure v_bitop3
# stream 869 G; more waves per SIMD lower the clock, not the time per instruction).  Reported beside the
# datasheet-priced fraction, never instead of it.
VALU_SUSTAINED_GIPS = 806.0
WARMUP_FLOOR_S = 0.025          # minimum duration of the warm-up (clock ramp), see main()
WARMUP_BURST = 16         # untimed steps right in front of the timed region (see main)
A_CV_BYTES = 10.0               # per pixel-disparity
A_MIN_BYTES = 6.0               # per pixel          


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", help="BASELINE.json configuration C1..C5")
    ap.add_argument("--pairs", type=int, default=1, help="stereo pairs per GPU per step")
    ap.add_argument("--threshold", type=float, default=0.15)
    ap.add_argument("--with-best", action="store_true", help="also write score_best")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--resident", type=int, default=4,
                    help="batches of input pairs resident in HBM; the steps rotate over them")
    ap.add_argument("--no-cost-modes", action="store_true", help="skip the extra `sad` / `ssd` objects (N = 1)")
    ap.add_argument("--cpu-rows", type=int, default=192, help="rows of the CPU-baseline band")
    ap.add_argument("--pipeline", action="store_true",
                    help="let consecutive steps overlap on the plan's two internal lanes (sm_plan_set_pipelined: "
                         "edges of step i+1 beside the match of step i, the head of match i+1 in the tail of match i; "
                         "measured +1.4 %% at C3, profiles/r03/ab_pipelined_lanes.txt).  Off by default: a step is "
                         "then no longer one serial pass, and the per-launch kernel time is that of launches "
                         "sharing the chip")
    ap.add_argument("--overlap", action="store_true",
                    help="issue the timed steps with sm_run_after -- a step's only input dependency is an event (none here: "
                         "the pairs are resident), so the plan overlaps consecutive steps on its two lanes where a match "
                         "launch cannot fill the chip twice over -- instead of plain sm_run in strict stream order.  The "
                         "default line carries that rate as the extra object `overlapped`; `value` stays the serial one, "
                         "whose kernel trace agrees launch by launch with roofline.kernel_ms")
    ap.add_argument("--c4", action="store_true",
                    help="add the `c4` object (64 x 1080p pairs sharded over the ranks + the collection of the maps "
                         "on rank 0) also at N = 1; at N > 1 it is always there")
    ap.add_argument("--graph", action="store_true", default=True,
                    help="(default) capture the steps in a HIP graph (up to 100 steps per graph) and replay it: the "
                         "host issues one launch per graph instead of two per step, the kernels of the timed region "
                         "run back to back from its first microsecond (a 20-step region is 2 ms: launched step by step "
                         "its first kernels run on clocks that sagged during the barrier in front of it, -7 %%), and a "
                         "run under rocprofv3 stays GPU-bound (profiles/r03: traced 0.115 ms per step against 0.096).  "
                         "Launches inside a captured graph cannot carry timing events: the kernel time is sampled in 16 "
                         "single steps right behind the timed region")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="launch every step from the host; the match launches of the timed region then carry their "
                         "own timing events (every 8th, every 2nd at --steps 20)")
    ap.add_argument("--gather", action="store_true",
                    help="after the timed region, collect the maps on rank 0 over RCCL and time it")
    ap.add_argument("--no-c2", action="store_true", help="skip the extra `c2` object (N = 1): BASELINE's second single-GPU "
                                                           "configuration (1080p pair, 64 shifts, 7x7), never part of `value`")
    ap.add_argument("--no-overlapped", action="store_true",
                    help="skip the extra `overlapped` object (N = 1): runs under rocprofv3 leave it out, so that every traced "
                         "match launch is one that has the chip to itself, as the launches roofline.kernel_ms is taken from")
    ap.add_argument("--corrupt-map", action="store_true", help=argparse.SUPPRESS)     # test hook: the verification must notice
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the extra `e2e` object (N = 1 only): the PCIe-inclusive rate through the "
                         "C ABI alone (pinned buffers, async copies), measured after the timed region; "
                         "it is never part of `value`")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------
# N > 1 without torchrun: start the ranks as children.  Nothing in this function
# (or before it in main) imports torch or touches the GPU.
# ---------------------------------------------------------------------------

def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv: list[str]) -> int:
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        # rank 0's stdout is the JSON line; every other rank's goes to our stderr
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    try:
        # a rank that dies leaves the others waiting in a rendezvous or a barrier: as soon as
        # one exits with an error the rest are stopped (exact PIDs we started, never a pattern)
        while any(p.poll() is None for p in procs):
            failed = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
            if failed:
                rc = abs(failed[0])
                break
            time.sleep(0.05)
        for p in procs:
            if p.poll() is not None:
                rc = max(rc, abs(p.returncode))
    except BaseException:
        rc = rc or 1
        raise
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        reader.join(timeout=5)
    if out0 and out0[0]:
        sys.stdout.write(out0[0].decode())
        sys.stdout.flush()
    return rc


def oracle_band(left, right, y0, rows, d, sw, mode, threshold, faithful=False):
    """web rows y0 .. y0 + rows - 1 of a full image pair by the CPU oracle: the band with a halo of
    half + 1 rows either side (edges need one row, the window `half`), full width.  -> (web band, seconds)"""
    from tests import oracle    # checker: bench.py touches oracle/ here and in cost_band only
    import numpy as np
    h = left.shape[0]
    halo = sw // 2 + 1
    idx = np.arange(y0 - halo, y0 + rows + halo)
    assert mode == "toroidal" or (idx[0] >= 0 and idx[-1] < h), "ghost band must lie inside the image"
    idx %= h
    sub_l, sub_r = np.ascontiguousarray(left[idx]), np.ascontiguousarray(right[idx])
    t0 = oracle.lib().smo_time()
    el = oracle.find_all_edges(sub_l, threshold, mode)
    er = oracle.find_all_edges(sub_r, threshold, mode)
    _, web = oracle.hot_path(el, er, d, sw, mode, faithful=faithful)
    dt = oracle.lib().smo_time() - t0
    return web[halo:halo + rows], dt, (el, er)


def cost_band(left, right, y0, rows, d, sw, mode, cost):
    """the same for the SAD / SSD cost mode (the build's own CPU definition) -> (best band, web band)"""
    from tests import oracle
    import numpy as np
    h = left.shape[0]
    half = sw // 2
    idx = np.arange(y0 - half, y0 + rows + half)
    assert mode == "toroidal" or (idx[0] >= 0 and idx[-1] < h)
    idx %= h
    best, web = oracle.cost_hot_path(np.ascontiguousarray(left[idx]), np.ascontiguousarray(right[idx]), d, sw, mode, cost)
    return best[half:half + rows], web[half:half + rows]


def cpu_baseline(left, right, y0, d, sw, mode, rows, threshold):
    """Time the oracle's faithful port (same loop nest and modulo indexing as stereo.c) on a full-width band of
    `rows` rows (+ halo) of the workload's own first pair.  -> (the object, the oracle's web of that band)"""
    from tests import oracle    # checker
    w = left.shape[1]
    web, dt, (el, er) = oracle_band(left, right, y0, rows, d, sw, mode, threshold, faithful=True)
    computed = el.shape[0]
    out = {
        "value": round(w * computed * d / dt / 1e6, 3),
        "unit": "Mpixel-disparities/s",
        "cores": 1,
        "kind": "port",
        "sample": f"rows {y0} .. {y0 + rows - 1} (+ {computed - rows} halo rows) of the workload's first pair: {w}x{computed} "
                  f"pixels, full width, all {d} shifts, S={sw}, {mode}; {dt:.1f} s single-threaded; "
                  f"host has {os.cpu_count()} cores",
    }
    # Independent bands on many host cores at once (as a batch of pairs would be spread over processes; ctypes releases
    # the GIL around the C call).  How many cores this process may really use is not what os.cpu_count() says on a
    # shared box (round 4 ran 64 threads and called it "all cores"; 256 threads on this pool's 16-core share take
    # 2.5 minutes): the thread count is doubled from 16 while the rate still grows, and the line says what was tried.
    import threading
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    hr = max(sw, 24)

    def job():
        oracle.hot_path(el[:hr], er[:hr], d, sw, mode, faithful=True)
    tried, best_rate, best_t = {}, 0.0, 1
    t = min(16, usable)
    while True:
        ts = [threading.Thread(target=job) for _ in range(t)]
        t0 = oracle.lib().smo_time()
        for th in ts:
            th.start()
        for th in ts:
            th.join()
        rate = t * w * hr * d / (oracle.lib().smo_time() - t0) / 1e6
        tried[str(t)] = round(rate, 1)
        grew = rate > 1.15 * best_rate
        if rate > best_rate:
            best_rate, best_t = rate, t
        if not grew or t >= usable:
            break
        t = min(2 * t, usable)
    out["all_cores"] = {"value": round(best_rate, 2), "cores": best_t, "usable_cores": usable,
                        "rate_by_threads": tried,
                        "sample": f"concurrent {w}x{hr} bands (hot path only), one per thread; threads doubled from 16 while "
                                  f"the rate grew by more than 15 %; affinity / cgroup allow {usable} cores, "
                                  f"os.cpu_count() = {os.cpu_count()}"}
    return out, web


def cost_modes(dev):
    """The extra `sad` / `ssd` (and `ssd_c3`: SSD on the headline geometry) objects: the SAD / SSD cost mode (sm_cost_wta) at the two BASELINE
    configurations that word their cost that way.  PARITY UNPINNED -- the reference implements the
    edge-equality cost and nothing else -- so these are reported beside the
    headline, never in it."""
    import numpy as np
    import torch
    from stereomatching_amd import pipeline
    from stereomatching_amd.synth import CONFIGS, make_pair

    counts = {}
    cfile = ROOT / "profiles" / "cost_valu.json"
    if cfile.exists():
        counts = json.loads(cfile.read_text())
    res = {}
    for key, cfg, cost in (("sad", "C3", "sad"), ("ssd", "C5", "ssd"), ("ssd_c3", "C3", "ssd")):
        w, h, d, sw, mode = CONFIGS[cfg]
        plan = pipeline.StereoPlan(w, h, d, sw, mode, device=dev.index)
        ls, rs = zip(*[make_pair(w, h, d, seed=500 + j) for j in range(2)])
        L = torch.from_numpy(np.stack(ls)).to(dev)
        R = torch.from_numpy(np.stack(rs)).to(dev)
        web = torch.empty((1, h, w), dtype=torch.int32, device=dev)
        t_end = time.perf_counter() + 0.025
        n = 0
        while n < 5 or time.perf_counter() < t_end:
            plan.cost_wta(L[n % 2], R[n % 2], cost, want_best=False, web=web)
            n += 1
        torch.cuda.synchronize(dev)
        launches = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(launches):
            plan.cost_wta(L[i % 2], R[i % 2], cost, want_best=False, web=web)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / launches
        # what was timed is checked: the map the LAST timed launch wrote (pair (launches - 1) % 2), a full-width band of it
        # against the build's own CPU definition, and best of a launch of its own
        vy0, vrows = h // 2, COST_VERIFY_ROWS
        k = (launches - 1) % 2
        _, ow = cost_band(ls[k], rs[k], vy0, vrows, d, sw, mode, cost)
        ok_band = bool(np.array_equal(web[0, vy0:vy0 + vrows].cpu().numpy(), ow))
        wb, bb = plan.cost_wta(L[k], R[k], cost, want_best=True)
        ob, _ = cost_band(ls[k], rs[k], vy0, 4, d, sw, mode, cost)
        ok_best = bool(torch.equal(wb, web)) and bool(np.array_equal(bb[0, vy0:vy0 + 4].cpu().numpy(), ob))
        o = {
            "workload": f"{cfg}: {w}x{h} pair, {d} shifts, {sw}x{sw} {cost.upper()} window cost, {mode} border; "
                        "uint8 gray in, int32 web out",
            "verified": ok_band and ok_best,
            "verified_how": f"rows {vy0} .. {vy0 + vrows - 1} of the map the last timed launch wrote equal the build's own CPU "
                            f"definition ({ok_band}); a further launch with `best` gives the same web and the definition's "
                            f"best on 4 rows ({ok_best})",
            "parity": "UNPINNED: no reference implementation exists (SURVEY.md 0, 8f4); checked against the "
                      "build's own CPU definition only",
            "ms_per_launch": round(ms, 4),
            "value": round(float(w) * h * d / ms / 1e3, 1),
            "unit": "Mpixel-disparities/s",
            "launches_timed": launches,
            "timing": "HIP events on the launch stream around 20 back-to-back launches",
        }
        c = counts.get(f"{cfg}:{cost}")
        if c:
            ach = c["valu_wave_instructions"] / (ms * 1e-3) / 1e9
            o["roofline"] = {"bound": "valu", "kernel": c.get("kernel"), "achieved": round(ach, 1),
                             "peak": VALU_PEAK_GIPS, "unit": "G wave-instr/s", "frac": round(ach / VALU_PEAK_GIPS, 4),
                             "valu_wave_instructions_per_launch": c["valu_wave_instructions"],
                             "lane_instructions_per_pixel_shift": round(c["valu_wave_instructions"] * 64.0 / (float(w) * h * d), 2),
                             **({"v_qsad_lane_instructions_per_pixel_shift": c["v_qsad_lane_instructions_per_pixel_shift"],
                                 "qsad_note": "of the lane-instructions; each issues over four passes, the others over one "
                                              "(SQ_ACTIVE_INST_VALU - SQ_INSTS_VALU = 3 x their number)"}
                                if "v_qsad_lane_instructions_per_pixel_shift" in c else {}),
                             "source": c.get("source")}
            if c.get("mfma_i8_instructions"):
                # the SSD kernel's products run on the matrix cores (v_mfma_i32_32x32x32_i8, 65 536 operations each);
                # peak: 2 x the guide's dense bf16 figure.  The launch is bound by the VALU / LDS work of ranking the
                # products, not by the matrix pipe -- this says how idle that pipe is.
                tops = c["mfma_i8_instructions"] * 65536.0 / (ms * 1e-3) / 1e12
                o["roofline"]["mfma"] = {"instructions_per_launch": c["mfma_i8_instructions"], "achieved": round(tops, 1),
                                         "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s", "frac": round(tops / MFMA_I8_PEAK_TOPS, 4)}
        res[key] = o
        plan.close()
    return res


COST_VERIFY_ROWS = 16


def small_config_leg(dev, cfg, threshold, use_graph):
    """The extra `c2` object (N = 1): BASELINE.json's other single-GPU configuration -- a lone 1080p pair, 64 shifts,
    7 x 7 -- through the same step as the headline (edges + fused match -> web, inputs resident), replayed from a HIP
    graph like the headline's steps: in stream order (`value`) and through sm_run_after (`overlapped`); the host-launched
    rate beside them.  Never part of the line's `value`."""
    import ctypes as C
    import numpy as np
    import torch
    from stereomatching_amd import pipeline
    from stereomatching_amd.synth import CONFIGS, make_pair

    w, h, d, sw, mode = CONFIGS[cfg]
    resident, gsteps, steps = 4, 100, 400
    plan = pipeline.StereoPlan(w, h, d, sw, mode, device=dev.index)
    plan.prepare_threshold(threshold)
    prs = [make_pair(w, h, d, seed=2000 + j) for j in range(resident)]
    L = torch.from_numpy(np.stack([p[0] for p in prs])).to(dev)
    R = torch.from_numpy(np.stack([p[1] for p in prs])).to(dev)
    web = torch.zeros((resident, h, w), dtype=torch.int32, device=dev)
    lib, check = pipeline.capi.lib, pipeline.capi.check

    def step(k, st, out=None, serial=True):
        o = web if out is None else out
        if serial:
            check(lib.sm_run(plan._h, C.c_void_p(L[k].data_ptr()), C.c_void_p(R[k].data_ptr()), threshold, 1,
                             C.c_void_p(o[k].data_ptr()), C.c_void_p(0), st))
        else:
            check(lib.sm_run_after(plan._h, C.c_void_p(L[k].data_ptr()), C.c_void_p(R[k].data_ptr()), threshold, 1,
                                   C.c_void_p(o[k].data_ptr()), 0, C.c_void_p(0), st, C.c_void_p(0)))
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for k in range(resident):
        step(k, stream)
    torch.cuda.synchronize(dev)
    chk = torch.empty_like(web)
    for k in range(resident):
        step(k, stream, chk)
    torch.cuda.synchronize(dev)

    def timed(serial):
        """(ms per step, the maps equal host-launched runs, how the steps were issued)"""
        graph, note = None, "launched from the host"
        if use_graph:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    cs = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                    for i in range(gsteps):
                        step(i % resident, cs, serial=serial)
                note = f"{gsteps} steps per HIP graph"
            except Exception as exc:      # noqa: BLE001
                graph, note = None, f"capture failed ({type(exc).__name__}): launched from the host"
                torch.cuda.synchronize(dev)

        def run(n):
            if graph is not None:
                for _ in range(n // gsteps):
                    graph.replay()
            else:
                for i in range(n):
                    step(i % resident, stream, serial=serial)
        t_end = time.perf_counter() + WARMUP_FLOOR_S
        while time.perf_counter() < t_end:
            run(gsteps)
            torch.cuda.synchronize(dev)
        web.zero_()
        run(gsteps)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / steps * 1e3
        return ms, web.clone(), note

    ms, maps, note = timed(True)
    ms_o, maps_o, note_o = timed(False)
    # host-launched, step by step, plain sm_run
    for i in range(16):
        step(i % resident, stream)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i % resident, stream)
    torch.cuda.synchronize(dev)
    ms_host = (time.perf_counter() - t0) / steps * 1e3
    vy0, vrows = h // 2, 32
    ow, _, _ = oracle_band(prs[0][0], prs[0][1], vy0, vrows, d, sw, mode, threshold)
    ok = bool(torch.equal(maps, chk)) and bool(np.array_equal(maps[0, vy0:vy0 + vrows].cpu().numpy(), ow))
    ok_o = bool(torch.equal(maps_o, chk))
    text = plan.describe()
    plan.close()
    rate = lambda t: round(float(w) * h * d / t / 1e3, 1)      # noqa: E731
    return {"workload": f"{cfg}: {w}x{h} pair, {d} shifts, {sw}x{sw} window, {mode} border, 1 pair/step; edges + fused "
                        "match/aggregate/WTA -> web, inputs resident",
            "ms_per_step": round(ms, 4), "value": rate(ms), "unit": "Mpixel-disparities/s",
            "steps": steps, "timed_steps": note + "; plain sm_run, stream order",
            "overlapped": {"ms_per_step": round(ms_o, 4), "value": rate(ms_o), "timed_steps": note_o +
                           "; sm_run_after: consecutive steps on the plan's two lanes", "verified": ok_o},
            "host_launched_ms_per_step": round(ms_host, 4), "kernel": text,
            "verified": ok and ok_o,
            "verified_how": f"the {resident} maps the timed steps left (both ways) equal host-launched runs; rows {vy0} .. "
                            f"{vy0 + vrows - 1} of the first equal the CPU oracle"}


C4_TOTAL_PAIRS = 64
C4_DISTINCT = 4                 # different synthetic pairs per rank (the rest of its share repeats them)


def c4_leg(rank, world, dev, red_dev, steps, threshold, rehearsal, dryrun):
    """The extra `c4` object: BASELINE config 4 / SURVEY.md 8e -- 64 x 1080p pairs, 64 shifts, 7x7, pair j ->
    rank j mod world, every rank's share in ONE launch per step, no data-path collective; then the maps are
    collected on rank 0 (shard.gather_maps: RCCL point to point) and that is timed apart.  Strong scaling: the
    job is 64 pairs whatever the world size.  Returns the object on rank 0, None elsewhere."""
    import torch
    from stereomatching_amd import shard
    from stereomatching_amd.synth import CONFIGS

    w, h, d, sw, mode = CONFIGS["C4"]
    mine = shard.pairs_for_rank(C4_TOTAL_PAIRS, rank, world)
    share = len(mine)
    steps = max(1, min(steps, 50))
    obj = {"workload": f"C4: {C4_TOTAL_PAIRS} x ({w}x{h} pair, {d} shifts, {sw}x{sw} window, {mode} border), pair j -> "
                       f"rank j mod {world}; each rank's share ({share} pairs on rank 0) in ONE launch per step",
           "total_pairs": C4_TOTAL_PAIRS, "pairs_per_rank": share, "steps": steps, "scaling": "strong",
           "unit": "Mpixel-disparities/s"}
    if dryrun:
        # plumbing only: the sharding, the barriers and the collection, on tiny stand-in maps whose value is
        # the index of the pair they belong to
        local = torch.stack([torch.full((4, 8), j, dtype=torch.int32) for j in mine]) if mine else \
            torch.zeros((0, 4, 8), dtype=torch.int32)
        shard.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            time.sleep(0.0005)
        shard.barrier()
        elapsed = shard.max_over_ranks(time.perf_counter() - t0, "cpu")
        g0 = time.perf_counter()
        got = shard.gather_maps(local, C4_TOTAL_PAIRS, rank, world)
        gather = shard.max_over_ranks(time.perf_counter() - g0, "cpu")
        if rank != 0:
            return None
        ok = all(bool((got[j] == j).all()) for j in range(C4_TOTAL_PAIRS))
        obj.update(dry_run=True, value=0.0, ms_per_step=round(elapsed / steps * 1e3, 4),
                   gather_ms=round(gather * 1e3, 3), maps_in_pair_order=ok)
        return obj

    import numpy as np
    from stereomatching_amd import pipeline
    from stereomatching_amd.synth import make_pair
    plan = pipeline.StereoPlan(w, h, d, sw, mode, max_pairs=max(1, share), device=dev.index)
    # pair j carries the synthetic pair number (j // world) % C4_DISTINCT: the same content on every rank
    # at the same position of its share, so that rank 0 can check the collected maps against its own
    base = [make_pair(w, h, d, seed=4000 + k) for k in range(min(C4_DISTINCT, max(1, share)))]
    idx = [(j // world) % C4_DISTINCT for j in mine] or [0]
    left = torch.from_numpy(np.stack([base[k][0] for k in idx])).to(dev)
    right = torch.from_numpy(np.stack([base[k][1] for k in idx])).to(dev)
    web = torch.empty((len(idx), h, w), dtype=torch.int32, device=dev)
    plan.prepare_threshold(threshold)
    n = len(idx) if share else 0

    def step():
        if n:
            plan.run(left, right, threshold, web=web)
    t_end = time.perf_counter() + WARMUP_FLOOR_S
    k = 0
    while k < 3 or time.perf_counter() < t_end:
        step()
        k += 1
    torch.cuda.synchronize(dev)
    shard.barrier()
    for _ in range(4):
        step()
    shard.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    shard.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, red_dev)
    # collection of the maps on rank 0, timed apart (never part of a step)
    local = web[:share]
    torch.cuda.synchronize(dev)
    shard.barrier()
    g0 = time.perf_counter()
    got = shard.gather_maps(local.cpu() if rehearsal else local, C4_TOTAL_PAIRS, rank, world)
    torch.cuda.synchronize(dev)
    gather = shard.max_over_ranks(time.perf_counter() - g0, red_dev)
    plan_text = plan.describe()
    plan.close()
    if rank != 0:
        return None
    mine0 = local.cpu() if rehearsal else local
    ok = got.shape[0] == C4_TOTAL_PAIRS and all(
        bool(torch.equal(got[j], mine0[(j // world) % min(C4_DISTINCT, share)])) for j in range(C4_TOTAL_PAIRS))
    inbound = (C4_TOTAL_PAIRS - share) * w * h * 4
    obj.update(value=round(float(w) * h * d * C4_TOTAL_PAIRS * steps / elapsed / 1e6, 1),
               ms_per_step=round(elapsed / steps * 1e3, 4), kernel=plan_text,
               gather_ms=round(gather * 1e3, 3), gather_bytes_inbound=inbound,
               gather_GBps=round(inbound / gather / 1e9, 1) if gather > 0 and inbound else None,
               gather_transport="gloo over host memory (rehearsal)" if rehearsal else
                                "RCCL point to point (ncclSend / ncclRecv under torch.distributed), maps device to device",
               maps_in_pair_order=bool(ok))
    return obj


def timing_stride(steps: int) -> int:
    """Bracket every `stride`-th match launch with HIP events: an event record costs
    ~4 us on the launch stream (tools/gap_probe.py), so long runs sample every 8th
    launch; short runs sample densely enough for >= 10 timed launches."""
    return max(1, min(8, steps // 10))


def main():
    args = parse()
    args.serial = not args.overlap
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # plumbing rehearsals (never set by the driver):
    #   SM_BENCH_REHEARSAL=1  N ranks on device 0 over gloo (a 1-GPU box), real kernels
    #   SM_BENCH_DRYRUN=1     no GPU at all: the step is a sleep; exercises launcher,
    #                         rendezvous, barrier, max-over-ranks and the JSON line on CPU.
    #                         Its line says so and carries value 0.
    dryrun = os.environ.get("SM_BENCH_DRYRUN") == "1"
    rehearsal = dryrun or os.environ.get("SM_BENCH_REHEARSAL") == "1"

    # The extra `e2e` object (N = 1): the PCIe-inclusive rate through the C ABI alone, measured
    # by a process of its own -- what a C host sees -- and BEFORE this process touches the GPU,
    # so that no process is started from one that has initialised it.  (Inside this process,
    # behind torch and the timed loop, the same code measured the 33 MB downloads ~13 % slower,
    # 0.76 vs 0.66 ms per pair, for a reason that was not found.)  Never part of `value`.
    e2e = None
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_e2e and not rehearsal:
        try:
            r = subprocess.run([sys.executable, str(Path(__file__).resolve().parent / "tools" / "e2e_bench.py"),
                                args.config, "24", os.environ.get("LOCAL_RANK", "0")],
                               capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
            e2e = dict(json.loads(line), process="a process of its own, run before the timed region "
                                                 "(C ABI only, no torch)")
        except Exception as exc:       # noqa: BLE001 -- reported, never fatal: it is an extra
            e2e = {"error": f"{type(exc).__name__}: e2e measurement failed"}

    # stdout carries ONE JSON line and nothing else: libraries write to file descriptor 1 behind
    # Python's back (gloo announces its connections there, a runtime may warn there), so from here
    # on fd 1 is stderr and the line goes to a duplicate of the real stdout at the very end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line: str) -> None:
        os.write(real_stdout, (line + "\n").encode())

    import torch
    from stereomatching_amd import shard
    from stereomatching_amd.synth import CONFIGS, make_pair

    # SM_BENCH_NCCL_SELFTEST=1: create the RCCL process group even for one rank, so that a
    # 1-GPU box runs the broadcast / barrier / all-reduce / gather calls of the N > 1 path
    selftest = os.environ.get("SM_BENCH_NCCL_SELFTEST") == "1"
    def fail_line(msg: str, code: int) -> None:
        """a job that cannot run still answers: rank 0 (or whoever would have been it) prints a line with `error`"""
        print(f"bench.py: {msg}", file=sys.stderr)
        if int(os.environ.get("RANK", "0")) == 0:
            emit(json.dumps({"metric": "Mpixel-disparities/s", "value": 0.0, "unit": "Mpixel-disparities/s",
                             "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                             "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "verified": False,
                             "error": msg, "config": {"workload": f"{args.config} (not run)"}}))
        sys.exit(code)

    try:
        rank, local_rank, world = shard.init("gloo" if rehearsal else None, force=selftest,
                                             timeout_s=float(os.environ.get("SM_BENCH_INIT_TIMEOUT", "120")))
        if world > 1 or selftest:
            # the communicator itself comes up in the first collective (RCCL is lazy): do it here, under the time limit
            # and by name, not somewhere inside the warm-up
            shard.barrier()
    except Exception as exc:      # noqa: BLE001 -- InitError, or the first barrier's failure
        fail_line(f"{type(exc).__name__}: {exc}", 4)
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as "
                  f"`python bench.py --gpus {args.gpus}` or under torchrun with "
                  f"--nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    red_dev = "cpu" if rehearsal else None

    # rank 0's job description goes to every rank (RCCL broadcast; a no-op at N = 1)
    job = shard.broadcast_params(dict(config=args.config, pairs=args.pairs, threshold=args.threshold,
                                      steps=args.steps, warmup=args.warmup))
    args.config, args.pairs, args.threshold = job["config"], job["pairs"], job["threshold"]
    args.steps, args.warmup = job["steps"], job["warmup"]
    w, h, d, sw, mode = CONFIGS[args.config]
    pairs = args.pairs

    if dryrun:
        for _ in range(args.warmup):
            time.sleep(0.0005)
        shard.barrier()
        shard.max_over_ranks(0.0, "cpu")
        shard.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.0005)
        shard.barrier()
        elapsed = shard.max_over_ranks(time.perf_counter() - t0, "cpu")
        c4 = c4_leg(rank, world, None, "cpu", args.steps, args.threshold, True, True) if (world > 1 or args.c4) else None
        if rank == 0:
            line = {"metric": "Mpixel-disparities/s", "value": 0.0,
                    "unit": "Mpixel-disparities/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dry_run": True, "data": "none: SM_BENCH_DRYRUN plumbing rehearsal, no GPU work",
                    "config": {"workload": f"{args.config} x {pairs} pair(s)/rank (not run)"}}
            if c4 is not None:
                line["c4"] = c4
            emit(json.dumps(line))
        shard.finalize()
        return

    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if red_dev is None:
        red_dev = dev

    from stereomatching_amd import pipeline   # fails loudly if the HIP library is missing

    plan = pipeline.StereoPlan(w, h, d, sw, mode, max_pairs=pairs, device=local_rank)

    import numpy as np
    # `resident` different batches of pairs live in HBM and the steps rotate over them (and over as
    # many result maps): no step reads the inputs or overwrites the map of the step before it.
    # (All of it still fits the 256 MB Infinity Cache at 4K; the path is VALU-bound, 5 % of HBM.)
    resident = max(1, args.resident)
    lefts, rights = [], []
    for j in range(pairs * resident):
        a, b = make_pair(w, h, d, seed=1000 * rank + j)
        lefts.append(a)
        rights.append(b)
    left = torch.from_numpy(np.stack(lefts)).to(dev)
    right = torch.from_numpy(np.stack(rights)).to(dev)
    web = torch.empty((pairs * resident, h, w), dtype=torch.int32, device=dev)
    best = torch.empty_like(web) if args.with_best else None

    import ctypes as C
    lib, check = pipeline.capi.lib, pipeline.capi.check
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    img, mp = pairs * w * h, pairs * w * h * 4
    p_l = [C.c_void_p(left.data_ptr() + k * img) for k in range(resident)]
    p_r = [C.c_void_p(right.data_ptr() + k * img) for k in range(resident)]
    p_web = [C.c_void_p(web.data_ptr() + k * mp) for k in range(resident)]
    p_best = [C.c_void_p(best.data_ptr() + k * mp) if best is not None else C.c_void_p(0) for k in range(resident)]
    turn = [0]

    # Optional: the inputs are resident and complete, so consecutive steps may overlap
    # (step i on the plan's internal lane i & 1).  Every step still does all its work.
    plan.set_pipelined(args.pipeline)
    plan.prepare_threshold(args.threshold)     # set-up next to the allocations

    def step(st=None, serial=False):
        k = turn[0]
        turn[0] = k + 1 if k + 1 < resident else 0
        if serial or args.serial:
            check(lib.sm_run(plan._h, p_l[k], p_r[k], args.threshold, pairs, p_web[k], p_best[k], st or stream))
        else:       # inputs resident: no event to wait for; consecutive steps may overlap (include/stereo_hip.h)
            check(lib.sm_run_after(plan._h, p_l[k], p_r[k], args.threshold, pairs, p_web[k], 0, p_best[k], st or stream,
                                   C.c_void_p(0)))
    geo0 = plan.geometry()
    overlapped = bool(args.pipeline) or (not args.serial and
                                         geo0["tiles_x"] * geo0["tiles_y"] * pairs * max(1, geo0["threads"] // 64) < 2048 and d <= 128)

    # --graph: `gsteps` consecutive steps (a whole number of turns over the resident batches) captured once
    graph, gsteps, graph_note = None, 0, None
    if args.graph and args.steps >= resident:
        try:
            gsteps = max(resident, min(args.steps, 100) // resident * resident)
            for _ in range(2):                     # (code objects loaded, tables built: nothing lazy inside the capture)
                step()
            torch.cuda.synchronize(dev)
            turn[0] = 0
            plan.time_kernels(0)                   # launches captured into a graph cannot carry timing events
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                cs = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                for _ in range(gsteps):
                    step(cs)
        except Exception as exc:      # noqa: BLE001 -- never fatal: the steps are then launched one by one
            graph, gsteps = None, 0
            graph_note = f"capture failed ({type(exc).__name__}): steps launched from the host"
            torch.cuda.synchronize(dev)
        turn[0] = 0

    def run_steps(n):
        """exactly n steps: whole graphs, then single steps"""
        if graph is not None:
            while n >= gsteps:
                graph.replay()
                n -= gsteps
        for _ in range(n):
            step()

    # W warm-up steps as asked; a 4K step is ~0.1 ms, so W = 5 is over before the chip has
    # left its idle clocks (the same kernel measures ~10 % slower in the first millisecond
    # than in steady state).  The warm-up therefore also lasts at least WARMUP_FLOOR_S;
    # the number of steps it took is reported as warmup_steps_run.
    # HIP events around the dominant kernel, recorded by the library on the stream the kernel
    # is launched on, inside the timed region itself.  The events are CREATED here, before the
    # warm-up (which uses them too); re-arming them in front of the timed region is free.
    every = timing_stride(args.steps)
    n_samples = (args.steps + every - 1) // every
    if graph is None:
        plan.time_kernels(n_samples, every)
    # (zeroed HERE, in front of the warm-up: whatever the maps hold after the timed region was written by steps issued the way
    # the timed ones are -- graph replays by default.  Zeroing them right in front of the last burst, 133 MB through the
    # memset path, left the host-launched timed region 5-10 % slower on the same box: profiles/r05/ab_r04_host_launched.txt)
    web.zero_()
    torch.cuda.synchronize(dev)
    warm_t0 = time.perf_counter()
    warm_steps = 0
    while warm_steps < args.warmup or time.perf_counter() - warm_t0 < WARMUP_FLOOR_S:
        if graph is not None:
            graph.replay()
            warm_steps += gsteps
            torch.cuda.synchronize(dev)
            continue
        step()
        warm_steps += 1
        if warm_steps % 16 == 0:
            torch.cuda.synchronize(dev)          # keep the launch queue short
    torch.cuda.synchronize(dev)
    # communicator start-up (RCCL) belongs to the warm-up, not to the timed region
    shard.barrier()
    shard.max_over_ranks(0.0, red_dev)
    # The host work since the warm-up (the communicator's first collectives take milliseconds)
    # left the chip idle, and it drops its clocks within a fraction of a millisecond: a last
    # burst of untimed steps keeps it busy until the barrier + synchronize that open the timed
    # region (a timed region of 20 steps is only 2 ms long; without the burst its first
    # launches ran ~10 % slower and the line read 0.099 ms per step where 200 steps read 0.0955)
    burst = gsteps if graph is not None else WARMUP_BURST
    run_steps(burst)
    warm_steps += burst
    if graph is None:
        plan.time_kernels(n_samples, every)        # re-arm: same capacity, no allocation

    shard.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize(dev)
    shard.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, red_dev)
    web_timed = web.clone()     # the maps as the timed steps left them (checked below, outside the timed region)
    if args.corrupt_map:
        web_timed[0, h // 2 + 3, 17] ^= 1

    if graph is not None:
        # (hipEventRecordWithFlags(..., hipEventRecordExternal) inside the capture is refused by this runtime:
        # the kernel time of a --graph run is sampled in a pass of single steps right behind the timed region)
        n_samples = 16
        plan.time_kernels(n_samples, 1)
        plan.set_pipelined(False)
        for _ in range(n_samples):
            step(serial=True)           # (a launch that has the chip to itself: what the roofline prices)
        torch.cuda.synchronize(dev)
    kernel_ms, n_timed = plan.kernel_ms()
    assert n_timed == n_samples, (n_timed, n_samples)
    units_per_step = float(w) * h * d * pairs                      # pixel-disparities / rank
    value = units_per_step * world * args.steps / elapsed / 1e6

    # ---- what was timed is checked (outside the timed region) ----------------------------------------------------
    # (a) every resident result map, as the timed steps (graph replays by default) left it, equals the map of a
    #     host-launched, unpipelined sm_run on the same inputs: compared on the device
    plan.time_kernels(0)
    plan.set_pipelined(False)
    torch.cuda.synchronize(dev)
    chk = torch.empty_like(web)
    for k in range(resident):
        check(lib.sm_run(plan._h, p_l[k], p_r[k], args.threshold, pairs, C.c_void_p(chk.data_ptr() + k * mp),
                         C.c_void_p(0), stream))
    torch.cuda.synchronize(dev)
    same_as_host_launched = bool(torch.equal(web_timed, chk))
    host_launched = None
    if graph is not None or overlapped:
        # ... and the rate of the plain host-launched path beside the default's (both in the line)
        turn[0] = 0
        n_hl = max(resident, min(args.steps, 200))
        t_end = time.perf_counter() + WARMUP_FLOOR_S       # (the checks above left the chip idle: clocks ramp up again first)
        while time.perf_counter() < t_end:
            for _ in range(WARMUP_BURST):
                step(serial=True)
            torch.cuda.synchronize(dev)
        h0 = time.perf_counter()
        for _ in range(n_hl):
            step(serial=True)
        torch.cuda.synchronize(dev)
        host_launched = {"ms_per_step": round((time.perf_counter() - h0) / n_hl * 1e3, 4), "steps": n_hl,
                         "note": "plain sm_run, every step launched from the host in stream order; this rank only, outside the "
                                 "timed region"}
    # ---- the same steps through sm_run_after, replayed from a graph of their own: consecutive steps overlap ----------
    overlapped_obj = None
    if graph is not None and args.serial and not args.pipeline and world == 1 and not args.no_overlapped:
        try:
            g2 = torch.cuda.CUDAGraph()
            turn[0] = 0
            with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                cs = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                for k in range(gsteps):
                    kk = k % resident
                    check(lib.sm_run_after(plan._h, p_l[kk], p_r[kk], args.threshold, pairs, p_web[kk], 0, p_best[kk], cs,
                                           C.c_void_p(0)))
            reps = max(1, args.steps // gsteps)
            for _ in range(max(2, int(WARMUP_FLOOR_S / (elapsed / args.steps * gsteps)) + 1)):
                g2.replay()
            web.zero_()
            g2.replay()
            torch.cuda.synchronize(dev)
            o0 = time.perf_counter()
            for _ in range(reps):
                g2.replay()
            torch.cuda.synchronize(dev)
            oms = (time.perf_counter() - o0) / (reps * gsteps) * 1e3
            overlapped_obj = {"ms_per_step": round(oms, 4), "value": round(units_per_step / oms / 1e3, 1),
                              "unit": "Mpixel-disparities/s", "steps": reps * gsteps,
                              "verified": bool(torch.equal(web, chk)),
                              "how": "sm_run_after instead of sm_run (a step waits for its inputs only: resident), replayed "
                                     "from a HIP graph; the plan runs consecutive steps on its two lanes: the edge detection "
                                     "of step i + 1 beside the match launch of step i.  Never `value`: launches that share "
                                     "the chip have no per-launch duration a trace could confirm"}
            del g2
        except Exception as exc:      # noqa: BLE001 -- an extra: reported, never fatal
            overlapped_obj = {"error": f"{type(exc).__name__}: {exc}"[:300], "verified": True}
            torch.cuda.synchronize(dev)
    # (b) a full-width band of the first map against the CPU oracle (rank 0, below: the cpu_baseline leg's own band)
    verify_rows = min(args.cpu_rows, 48) if (args.no_cpu_baseline or world > 1) else args.cpu_rows
    verify_rows = max(1, min(verify_rows, h - 2 * (sw // 2 + 1)))
    vy0 = max(sw // 2 + 1, min(h // 2, h - verify_rows - sw // 2 - 1))
    band_gpu = web_timed[0, vy0:vy0 + verify_rows].cpu().numpy()
    ranks_differing = shard.max_over_ranks(0.0 if same_as_host_launched else 1.0, red_dev)
    del web_timed, chk

    gather_ms = None
    if args.gather and (world > 1 or selftest):
        torch.cuda.synchronize(dev)
        shard.barrier()
        g0 = time.perf_counter()
        shard.gather_maps(web[:pairs].cpu() if rehearsal else web[:pairs], pairs * world, rank, world)
        torch.cuda.synchronize(dev)
        gather_ms = shard.max_over_ranks(time.perf_counter() - g0, red_dev) * 1e3

    # SURVEY 8e's scaling workload beside the weak-scaling headline (every rank takes part)
    c4 = None
    plan_geo_keep = (plan.geometry(), plan.describe(), plan.valu_model(pairs, want_best=args.with_best))
    if world > 1 or args.c4:
        plan.close()
        del left, right, web, best
        torch.cuda.empty_cache()
        c4 = c4_leg(rank, world, dev, red_dev, args.steps, args.threshold, rehearsal, False)

    if rank != 0:
        shard.finalize()
        if not same_as_host_launched:
            sys.exit(3)
        return

    kernel_s = kernel_ms * 1e-3
    acv = A_CV_BYTES * units_per_step / kernel_s / 1e9             # GB/s (throughput bar)
    geo, plan_text, model = plan_geo_keep
    compulsory = 4.0 * w * h * pairs + 2 * 4.0 * geo["ext_words"] * geo["ext_rows"] * pairs     # web out + packed edge bits in
    amin_kernel = compulsory / kernel_s / 1e9
    amin_step = A_MIN_BYTES * w * h * pairs / (elapsed / args.steps) / 1e9
    # VALU wave-instructions of one launch: the plan's analytic model (per-wave set-up +
    # warm-up rows + output rows, coefficients fitted to SQ_INSTS_VALU of rocprofv3 --pmc
    # passes at several tile heights: stereomatching_amd/valu_counts.json)
    # HBM bytes of one launch: separate rocprofv3 --pmc passes of this same command
    # (tools/collect_profiles.sh), committed under profiles/; null if none for this config
    traffic = traffic_src = None
    tfile = ROOT / "profiles" / "hbm_traffic.json"
    if tfile.exists():
        t = json.loads(tfile.read_text()).get(f"{args.config}:{pairs}")
        if t:
            traffic = t["bytes_per_launch"]
            traffic_src = t.get("source", "profiles/hbm_traffic.json")

    roof = {
        "bound": "valu",
        "kernel": "k_match_bs" if "bit-sliced" in plan_text else "k_match_wta",
        "achieved": None, "peak": VALU_PEAK_GIPS, "unit": "G wave-instr/s", "frac": None,
        "peak_definition": "1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 VALU instruction "
                           "(MI355X_MICROARCH.md)",
        "traffic": traffic,
        "traffic_source": traffic_src,
        "kernel_ms": round(kernel_ms, 4),
        "kernel_launches_timed": n_timed,
        "kernel_ms_sampled_outside_timed_region": graph is not None,
        "kernel_ms_method": ("start / end time stamps of the dispatch itself (hipExtLaunchKernel events: the "
                             "clock rocprofv3's kernel trace reads), on the launch stream"
                             if "bit-sliced" in plan_text else
                             "HIP event records around the launch, on the launch stream") +
                            (f"; --graph: sampled in a pass of {n_timed} single steps right behind the timed region "
                             "(the timed steps themselves were replayed from a HIP graph)" if graph is not None else ""),
    }
    if model:
        ach = model["wave_instructions"] / kernel_s / 1e9
        roof.update(achieved=round(ach, 1), frac=round(ach / VALU_PEAK_GIPS, 4),
                    sustained_peak=VALU_SUSTAINED_GIPS, frac_of_sustained=round(ach / VALU_SUSTAINED_GIPS, 4),
                    sustained_peak_definition="measured issue rate of the kernel's instruction mix (15 v_bitop3 : 1 "
                                              "v_alignbit) at two waves per SIMD, profiles/r03/ubench_sad.txt; the chip "
                                              "lowers its clock under a dense VALU stream, the datasheet peak is not reachable",
                    valu_wave_instructions_per_launch=model["wave_instructions"],
                    valu_model=model["model"], valu_model_source=model["source"])
    roof.update({
        "hbm_frac_min": round(amin_kernel / HBM_PEAK_GBPS, 5),
        "hbm_achieved_min_GBps": round(amin_kernel, 1),
        "hbm_model_min": "compulsory bytes of this launch (packed edge bits in + i32 web out) / "
                         "kernel time / 8 TB/s",
        "throughput_bar_frac": round(acv / HBM_PEAK_GBPS, 4),
        "throughput_bar_model": "A_cv = 10 B per pixel-disparity (materialised cost volume, SURVEY 8d) "
                                "/ kernel time / 8 TB/s: a throughput bar (>= 0.6 <=> the north-star "
                                "target), NOT traffic -- the fused kernel never moves these bytes",
        "step_min_GBps": round(amin_step, 1),
    })

    out = {
        "metric": "Mpixel-disparities/s",
        "value": round(value, 1),
        "unit": "Mpixel-disparities/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "warmup_effective": warm_steps,
        "warmup_steps_run": warm_steps,
        "warmup_note": f"{args.warmup} steps asked; untimed warm-up continued to {WARMUP_FLOOR_S * 1e3:.0f} ms "
                       f"and ends with {WARMUP_BURST} steps right in front of the timed region, so that "
                       "the timed steps run at steady clocks",
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 bit-packed edges, i32 counts",
        "data": "synthetic",
        "inputs": f"{resident} different resident batches of {pairs} pair(s), the steps rotate over them and over "
                  f"{resident} result maps (all Infinity-Cache resident: "
                  f"{(2 * img + mp) * resident / 1e6:.0f} MB)",
        "config": {
            "workload": f"{args.config}: {w}x{h} pair, {d} shifts, {sw}x{sw} window, {mode} border, "
                        f"{pairs} pair(s)/GPU/step; edges + fused match/aggregate/WTA -> web",
            "kernel": plan_text,
            "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective",
            "pipelined": overlapped,
            "graph": f"{gsteps} steps per HIP graph" if graph is not None else (graph_note or False),
        },
        "timed_steps": ("replayed from HIP graphs" if graph is not None else "launched from the host") +
                       ("; consecutive steps overlapped on the plan's two lanes (sm_run_after: a step's only input "
                        "dependency is its resident pair)" if overlapped else "; strict stream order"),
        "graph_capture": "ok" if graph is not None else ("not asked for" if not args.graph or args.steps < resident
                                                         else (graph_note or "failed")),
        "roofline": roof,
    }
    if host_launched is not None:
        out["host_launched"] = host_launched
    if overlapped_obj is not None:
        out["overlapped"] = overlapped_obj
    if rehearsal:
        out["rehearsal"] = "SM_BENCH_REHEARSAL: all ranks on device 0 over gloo"
    if gather_ms is not None:
        out["gather_ms"] = round(gather_ms, 3)
    if e2e is not None:
        out["e2e"] = e2e
    if c4 is not None:
        out["c4"] = c4
    if c4 is None:
        plan.close()
        del left, right, web
        torch.cuda.empty_cache()
    if world == 1 and not args.no_c2 and not rehearsal and args.config != "C2":
        out["c2"] = small_config_leg(dev, "C2", args.threshold, args.graph)
    if world == 1 and not args.no_cost_modes and not rehearsal:
        out.update(cost_modes(dev))
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], oweb = cpu_baseline(lefts[0], rights[0], vy0, d, sw, mode, verify_rows, args.threshold)
        how = "the cpu_baseline leg's own band"
    else:
        oweb, _, _ = oracle_band(lefts[0], rights[0], vy0, verify_rows, d, sw, mode, args.threshold)
        how = "the CPU oracle's separable path"
    band_ok = bool(np.array_equal(band_gpu, oweb))
    extras_ok = all(o.get("verified", True) for o in out.values() if isinstance(o, dict))
    out["verified"] = bool(ranks_differing == 0.0 and band_ok and extras_ok)
    out["verification"] = {
        "maps_equal_host_launched_runs": ranks_differing == 0.0,
        "band_equals_cpu_oracle": band_ok,
        "extras_verified": extras_ok,
        "how": f"after the timed region: all {resident} resident result maps, as the timed steps left them (zeroed in front "
               f"of the warm-up), equal (torch.equal, on the device, every rank) the maps of host-launched "
               f"unpipelined sm_run calls on the same inputs; rows {vy0} .. {vy0 + verify_rows - 1} of the first one equal "
               f"{how} on the same pair; the `c2` / `sad` / `ssd` objects carry their own checks",
    }
    emit(json.dumps(out))
    shard.finalize()
    if not out["verified"]:
        print("bench.py: VERIFICATION FAILED: " + json.dumps(out["verification"]), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
