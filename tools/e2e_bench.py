#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) throughput through the C ABI alone: pairs start in
pinned host memory, results end in pinned host memory.  Three streams' worth of
work are kept in flight (upload of pair k+1, kernels of pair k, download of pair
k-1) with per-slot buffers.  This is NOT bench.py's `value` (which is measured
with resident inputs); DESIGN.md quotes it next to it.

    python tools/e2e_bench.py [C3] [pairs=24] [slots=3]
"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stereomatching_amd.capi import check, lib  # noqa: E402
from stereomatching_amd.synth import CONFIGS, make_pair  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 3
w, h, d, sw, mode = CONFIGS[cfg]
n = w * h
vp = C.c_void_p


def hostbuf(nbytes):
    p = vp()
    check(lib.sm_host_alloc(nbytes, C.byref(p)))
    return p


def devbuf(nbytes):
    p = vp()
    check(lib.sm_malloc(0, nbytes, C.byref(p)))
    return p


left, right = make_pair(w, h, d, seed=3)
S = []
for i in range(slots):
    s = dict(hl=hostbuf(n), hr=hostbuf(n), hw=hostbuf(4 * n), dl=devbuf(n), dr=devbuf(n), dw=devbuf(4 * n),
             st=vp(), plan=vp())
    C.memmove(s["hl"], left.ctypes.data, n)
    C.memmove(s["hr"], right.ctypes.data, n)
    check(lib.sm_stream_create(0, C.byref(s["st"])))
    check(lib.sm_plan_create(0, w, h, d, sw, 1 if mode == "ghost" else 0, 1, C.byref(s["plan"])))
    S.append(s)


def submit(s):
    st = s["st"]
    check(lib.sm_memcpy_h2d_async(0, s["dl"], s["hl"], n, st))
    check(lib.sm_memcpy_h2d_async(0, s["dr"], s["hr"], n, st))
    check(lib.sm_run(s["plan"], s["dl"], s["dr"], 0.15, 1, s["dw"], None, st))
    check(lib.sm_memcpy_d2h_async(0, s["hw"], s["dw"], 4 * n, st))


for s in S:                      # warm-up
    submit(s)
for s in S:
    check(lib.sm_stream_sync(0, s["st"]))
t0 = time.perf_counter()
for k in range(npairs):
    s = S[k % slots]
    if k >= slots:
        check(lib.sm_stream_sync(0, s["st"]))    # the slot's previous pair has fully landed
    submit(s)
for s in S:
    check(lib.sm_stream_sync(0, s["st"]))
dt = time.perf_counter() - t0
web = np.ctypeslib.as_array(C.cast(S[0]["hw"], C.POINTER(C.c_int32)), (h, w))
assert web.min() >= 1 and web.max() <= d
mb = (2 * n + 4 * n) / 1e6
print(f"{cfg}: {npairs} pairs through {slots} slots: {dt / npairs * 1e3:.3f} ms/pair end to end, "
      f"{w * h * d * npairs / dt / 1e6:.0f} Mpixel-disparities/s, {mb * npairs / dt / 1e3:.1f} GB/s over PCIe "
      f"({2 * n / 1e6:.1f} MB in + {4 * n / 1e6:.1f} MB out per pair)")
