#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) throughput through the C ABI alone: pairs start in
pinned host memory, results end in pinned host memory.  Uploads, kernels and downloads
run on a stream each (upload of pair k+1, kernels of pair k, download of pair k-1 at the
same time, both directions of the link busy), chained with events over per-slot buffers.  This is NOT bench.py's `value` (which is
measured with resident inputs); bench.py (N = 1) reports it as the extra object `e2e` and
DESIGN.md quotes it next to the resident number.

    python tools/e2e_bench.py [C3] [pairs=24] [device=0]

Measured for the reference's int32 web map (sm_run) and for the narrow map
(sm_run_typed: uint8 when the shifts fit, else uint16), which moves 4x / 2x fewer
bytes back over PCIe.
"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

vp = C.c_void_p


def measure_one(cfg: str, device: int, web_type: int, npairs: int = 24, slots: int = 3) -> dict:
    """Three streams -- uploads, kernels, downloads -- so that each copy direction has a DMA
    queue to itself and both directions of the link are busy at once; `slots` buffer sets in
    flight, chained with events: upload k waits for the kernels that last read its slot's input
    buffers, kernels k wait for upload k and for the download that last read the slot's map,
    download k waits for kernels k.  The host takes a slot back (as a consumer of the result
    would) only after its download has landed."""
    from stereomatching_amd.capi import SM_WEB_I32, SM_WEB_U8, check, lib
    from stereomatching_amd.synth import CONFIGS, make_pair

    w, h, d, sw, mode = CONFIGS[cfg]
    n = w * h
    wb = 4 if web_type == SM_WEB_I32 else (1 if web_type == SM_WEB_U8 else 2)

    def hostbuf(nbytes):
        p = vp()
        check(lib.sm_host_alloc(nbytes, C.byref(p)))
        return p

    def devbuf(nbytes):
        p = vp()
        check(lib.sm_malloc(device, nbytes, C.byref(p)))
        return p

    def make(fn):
        p = vp()
        check(fn(device, C.byref(p)))
        return p

    left, right = make_pair(w, h, d, seed=3)
    st_up, st_run, st_down = (make(lib.sm_stream_create) for _ in range(3))
    plan = vp()
    check(lib.sm_plan_create(device, w, h, d, sw, 1 if mode == "ghost" else 0, 1, C.byref(plan)))
    S = []
    for i in range(slots):
        s = dict(hl=hostbuf(n), hr=hostbuf(n), hw=hostbuf(wb * n), dl=devbuf(n), dr=devbuf(n),
                 dw=devbuf(wb * n), up=make(lib.sm_event_create), ran=make(lib.sm_event_create),
                 down=make(lib.sm_event_create), used=False)
        C.memmove(s["hl"], left.ctypes.data, n)
        C.memmove(s["hr"], right.ctypes.data, n)
        S.append(s)

    def submit(s):
        if s["used"]:
            check(lib.sm_stream_wait_event(device, st_up, s["ran"]))      # inputs free again
        check(lib.sm_memcpy_h2d_async(device, s["dl"], s["hl"], n, st_up))
        check(lib.sm_memcpy_h2d_async(device, s["dr"], s["hr"], n, st_up))
        check(lib.sm_event_record(device, s["up"], st_up))
        check(lib.sm_stream_wait_event(device, st_run, s["up"]))
        if s["used"]:
            check(lib.sm_stream_wait_event(device, st_run, s["down"]))   # map buffer free again
        check(lib.sm_run_typed(plan, s["dl"], s["dr"], 0.15, 1, s["dw"], web_type, None, st_run))
        check(lib.sm_event_record(device, s["ran"], st_run))
        check(lib.sm_stream_wait_event(device, st_down, s["ran"]))
        check(lib.sm_memcpy_d2h_async(device, s["hw"], s["dw"], wb * n, st_down))
        check(lib.sm_event_record(device, s["down"], st_down))
        s["used"] = True

    # warm-up: every slot at least twice, and at least 150 ms of traffic -- a link that has been
    # idle (bench.py calls this after a timed region without transfers) takes tens of milliseconds to leave its
    # low-power state, during which a 33 MB download runs at half speed
    t_warm = time.perf_counter()
    rounds = 0
    while rounds < 2 or time.perf_counter() - t_warm < 0.150:
        for s in S:
            if s["used"]:
                check(lib.sm_event_sync(device, s["down"]))
            submit(s)
        rounds += 1
    check(lib.sm_stream_sync(device, st_down))
    t0 = time.perf_counter()
    for k in range(npairs):
        s = S[k % slots]
        if k >= slots:
            check(lib.sm_event_sync(device, s["down"]))    # the slot's previous result has landed
        submit(s)
    check(lib.sm_stream_sync(device, st_down))
    dt = time.perf_counter() - t0
    ctype = {4: C.c_int32, 2: C.c_uint16, 1: C.c_uint8}[wb]
    web = np.ctypeslib.as_array(C.cast(S[0]["hw"], C.POINTER(ctype)), (h, w))
    assert web.min() >= 1 and web.max() <= d
    lib.sm_plan_destroy(plan)
    for s in S:
        for k in ("dl", "dr", "dw"):
            check(lib.sm_free(device, s[k]))
        for k in ("hl", "hr", "hw"):
            check(lib.sm_host_free(s[k]))
        for k in ("up", "ran", "down"):
            check(lib.sm_event_destroy(device, s[k]))
    for st in (st_up, st_run, st_down):
        check(lib.sm_stream_destroy(device, st))
    mb = (2 * n + wb * n) / 1e6
    return {
        "web_dtype": {4: "int32", 2: "uint16", 1: "uint8"}[wb],
        "ms_per_pair": round(dt / npairs * 1e3, 4),
        "Mpixel_disparities_per_s": round(w * h * d * npairs / dt / 1e6, 1),
        "pcie_GBps": round(mb * npairs / dt / 1e3, 2),
        "bytes_per_pair": {"in": 2 * n, "out": wb * n},
        "pairs": npairs, "slots_in_flight": slots,
        "streams": "uploads / kernels / downloads on one stream each",
    }


def measure(cfg: str = "C3", device: int = 0, npairs: int = 24) -> dict:
    """PCIe-inclusive rates for the reference's int32 map and for the narrow map."""
    from stereomatching_amd.capi import SM_WEB_I32, SM_WEB_U8, SM_WEB_U16
    from stereomatching_amd.synth import CONFIGS
    d = CONFIGS[cfg][2]
    return {
        "what": "uint8 pairs from pinned host memory -> web maps in pinned host memory, C ABI only "
                "(sm_memcpy_*_async + sm_run_typed), never part of `value`",
        "config": cfg,
        "int32": measure_one(cfg, device, SM_WEB_I32, npairs),
        "narrow": measure_one(cfg, device, SM_WEB_U8 if d <= 255 else SM_WEB_U16, npairs),
    }


if __name__ == "__main__":
    import json
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    device = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    print(json.dumps(measure(cfg, device, npairs)))
