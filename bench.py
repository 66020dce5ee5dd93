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
job over all ranks.  N > 1: one process per GPU (torchrun), every rank runs
the same number of its own pairs (weak scaling), no data-path collective;
barrier + device sync on both sides of the timed region, max over ranks.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the dominant kernel (k_match_wta) against the HBM roof, on the
                materialised-cost-volume byte model A_cv; the
                compulsory-traffic model A_min is printed next to it because a
                fused kernel moves almost no bytes
  cpu_baseline  the oracle's structure-faithful C port timed on this host's
                cores on a bounded band of the same workload (rank 0, N = 1)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
A_CV_BYTES = 10.0               # per pixel-disparity
A_MIN_BYTES = 6.0               # per pixel          


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", help="BASELINE.json configuration C1..C5")
    ap.add_argument("--pairs", type=int, default=1, help="stereo pairs per GPU per step")
    ap.add_argument("--threshold", type=float, default=0.15)
    ap.add_argument("--with-best", action="store_true", help="also write score_best")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=192, help="rows of the CPU-baseline band")
    ap.add_argument("--pipeline", action="store_true",
                    help="overlap the edge kernel of step i+1 with the match kernel of step i "
                         "(sm_plan_set_pipelined; measured: no net gain, both kernels are VALU-heavy)")
    ap.add_argument("--gather", action="store_true",
                    help="after the timed region, collect the maps on rank 0 over RCCL and time it")
    return ap.parse_args()


def cpu_baseline(w, d, sw, mode, rows, threshold):
    """Time the oracle's faithful port (same loop nest and modulo indexing as
    stereo.c) on a full-width band of `rows` rows of the workload."""
    import numpy as np
    from stereomatching_amd.synth import make_pair
    from tests import oracle    # checker: the only place bench.py touches oracle/

    left, right = make_pair(w, rows, d, seed=9)
    t0 = oracle.lib().smo_time()
    el = oracle.find_all_edges(left, threshold, mode)
    er = oracle.find_all_edges(right, threshold, mode)
    oracle.hot_path(el, er, d, sw, mode, faithful=True)
    dt = oracle.lib().smo_time() - t0
    out = {
        "value": round(w * rows * d / dt / 1e6, 3),
        "unit": "Mpixel-disparities/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{w}x{rows} band (full width, all {d} shifts, S={sw}, {mode}) of the workload, "
                  f"{dt:.1f} s single-threaded; host has {os.cpu_count()} cores",
    }
    # the same band on every host core at once (independent bands, as a batch of pairs
    # would be spread over processes; ctypes releases the GIL around the C call)
    import threading
    workers = max(1, min(os.cpu_count() or 1, 64))

    hr = max(sw, rows // 2)             # half the band per worker keeps this leg ~20 s

    def job():
        oracle.hot_path(el[:hr], er[:hr], d, sw, mode, faithful=True)
    ts = [threading.Thread(target=job) for _ in range(workers)]
    t0 = oracle.lib().smo_time()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dta = oracle.lib().smo_time() - t0
    out["all_cores"] = {"value": round(workers * w * hr * d / dta / 1e6, 2), "cores": workers,
                        "sample": f"{workers} concurrent {w}x{hr} bands (hot path only), {dta:.1f} s"}
    return out


def main():
    args = parse()
    import torch
    from stereomatching_amd import shard
    from stereomatching_amd.synth import CONFIGS, make_pair

    # rehearsal hooks for a 1-GPU box (never set by the driver): run N ranks on
    # device 0 with the gloo backend to exercise the multi-process path
    rehearsal = os.environ.get("SM_BENCH_REHEARSAL") == "1"
    rank, local_rank, world = shard.init("gloo" if rehearsal else None)
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torchrun",
                  file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from stereomatching_amd import pipeline   # fails loudly if the HIP library is missing

    # rank 0's job description goes to every rank (RCCL broadcast; a no-op at N = 1)
    job = shard.broadcast_params(dict(config=args.config, pairs=args.pairs, threshold=args.threshold,
                                      steps=args.steps, warmup=args.warmup))
    args.config, args.pairs, args.threshold = job["config"], job["pairs"], job["threshold"]
    args.steps, args.warmup = job["steps"], job["warmup"]
    w, h, d, sw, mode = CONFIGS[args.config]
    pairs = args.pairs
    plan = pipeline.StereoPlan(w, h, d, sw, mode, max_pairs=pairs, device=local_rank)

    import numpy as np
    lefts, rights = [], []
    for j in range(pairs):
        a, b = make_pair(w, h, d, seed=1000 * rank + j)
        lefts.append(a)
        rights.append(b)
    left = torch.from_numpy(np.stack(lefts)).to(dev)
    right = torch.from_numpy(np.stack(rights)).to(dev)
    web = torch.empty((pairs, h, w), dtype=torch.int32, device=dev)
    best = torch.empty_like(web) if args.with_best else None

    import ctypes as C
    lib, check = pipeline.capi.lib, pipeline.capi.check
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p_l, p_r = C.c_void_p(left.data_ptr()), C.c_void_p(right.data_ptr())
    p_web = C.c_void_p(web.data_ptr())
    p_best = C.c_void_p(best.data_ptr()) if best is not None else C.c_void_p(0)

    # Optional: the inputs are resident and complete, so consecutive steps may overlap
    # (edge kernel of step i+1 beside the match kernel of step i on the plan's internal
    # stream).  Every step still does all its work.
    plan.set_pipelined(args.pipeline)

    def step():
        check(lib.sm_run(plan._h, p_l, p_r, args.threshold, pairs, p_web, p_best, stream))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    # communicator start-up (RCCL) belongs to the warm-up, not to the timed region
    shard.barrier()
    shard.max_over_ranks(0.0, "cpu" if rehearsal else dev)
    # HIP events around the dominant kernel, recorded by the library on the stream
    # the kernel is launched on, inside the timed region itself.  Event records cost
    # ~4 us each on that stream (tools/gap_probe.py), so every 8th launch is bracketed.
    every = 8 if args.steps >= 16 else 1
    n_samples = (args.steps + every - 1) // every
    plan.time_kernels(n_samples, every)

    shard.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    shard.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, "cpu" if rehearsal else dev)

    kernel_ms, n_timed = plan.kernel_ms()
    assert n_timed == n_samples
    units_per_step = float(w) * h * d * pairs                      # pixel-disparities / rank
    value = units_per_step * world * args.steps / elapsed / 1e6

    gather_ms = None
    if args.gather and world > 1:
        torch.cuda.synchronize(dev)
        shard.barrier()
        g0 = time.perf_counter()
        shard.gather_maps(web.cpu() if rehearsal else web, pairs * world, rank, world)
        torch.cuda.synchronize(dev)
        gather_ms = shard.max_over_ranks(time.perf_counter() - g0, "cpu" if rehearsal else dev) * 1e3

    if rank != 0:
        return

    acv = A_CV_BYTES * units_per_step / (kernel_ms * 1e-3) / 1e9   # GB/s
    amin_kernel = (4.0 * w * h * pairs + plan.workspace_bytes() / 2 / plan.max_pairs * pairs) \
        / (kernel_ms * 1e-3) / 1e9
    amin_step = A_MIN_BYTES * w * h * pairs / (elapsed / args.steps) / 1e9
    # HBM bytes and VALU instruction count of one launch come from separate rocprofv3
    # --pmc passes of this same command (tools/collect_profiles.sh), committed under
    # profiles/; null if no profile exists for this configuration
    traffic = valu_issue = None
    tfile = ROOT / "profiles" / "hbm_traffic.json"
    if tfile.exists():
        t = json.loads(tfile.read_text()).get(f"{args.config}:{pairs}")
        if t:
            traffic = t["bytes_per_launch"]
            if t.get("valu_wave_instructions_per_launch"):
                # fraction of the chip's full-rate VALU issue (1024 SIMDs x one wave64
                # instruction per ~1.0 ns) this launch sustained
                valu_issue = t["valu_wave_instructions_per_launch"] / (kernel_ms * 1e-3) / (1024 * 1.0e9)

    out = {
        "metric": "Mpixel-disparities/s",
        "value": round(value, 1),
        "unit": "Mpixel-disparities/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 bit-packed edges, i32 counts",
        "data": "synthetic",
        "config": {
            "workload": f"{args.config}: {w}x{h} pair, {d} shifts, {sw}x{sw} window, {mode} border, "
                        f"{pairs} pair(s)/GPU/step; edges + fused match/aggregate/WTA -> web",
            "kernel": plan.describe(),
            "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective",
            "pipelined": args.pipeline,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "k_match_bs" if "bit-sliced" in plan.describe() else "k_match_wta",
            "achieved": round(acv, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(acv / HBM_PEAK_GBPS, 4),
            "traffic": traffic,
            "kernel_ms": round(kernel_ms, 4),
            "kernel_launches_timed": n_timed,
            "model": "A_cv = 10 B per pixel-disparity (materialised cost volume, SURVEY 8d); "
                     "the fused kernel never moves these bytes, so frac > 1 is possible",
            "achieved_min": round(amin_kernel, 1),
            "frac_min": round(amin_kernel / HBM_PEAK_GBPS, 5),
            "model_min": "compulsory bytes of this launch: packed edge bits in + i32 web out",
            "step_min_GBps": round(amin_step, 1),
            "valu_issue_frac": round(valu_issue, 3) if valu_issue else None,
        },
    }
    if gather_ms is not None:
        out["gather_ms"] = round(gather_ms, 3)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w, d, sw, mode, args.cpu_rows, args.threshold)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
