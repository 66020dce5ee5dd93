"""Host-side mirror of the reference's pipeline interface over the C ABI.

The reference has no library API: `algorithm(first, second, width, height,
AlgorithmParams)` (/root/reference/src/stereo.cu:289-347) calls its stage
kernels in order.  `StereoPlan` exposes the same stages under the same names
and argument meaning, each one a thin call into libstereo_hip.so.  torch is
plumbing only: it owns the device buffers and the stream the launches go to.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import capi
from .capi import BORDERS, check, lib

# defaults of /root/reference/src/stereo.c:6-10
NUM_SHIFTS = 30
DEFAULT_THRESHOLD = 0.15
DEFAULT_SQUARE_WIDTH = 21
DEFAULT_TIMES = 32
DEFAULT_LINES = 10


@dataclass
class AlgorithmParams:
    """src/stereo.c:280-285"""
    threshold: float = DEFAULT_THRESHOLD
    square_width: int = DEFAULT_SQUARE_WIDTH
    times: int = DEFAULT_TIMES
    lines_to_draw: int = DEFAULT_LINES


WEB_TYPES = {torch.int32: capi.SM_WEB_I32, torch.uint16: capi.SM_WEB_U16, torch.uint8: capi.SM_WEB_U8}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class StereoPlan:
    """Geometry + device workspace for one image size / shift count / window."""

    def __init__(self, width: int, height: int, num_shifts: int = NUM_SHIFTS,
                 square_width: int = DEFAULT_SQUARE_WIDTH, border: str | int = "toroidal",
                 max_pairs: int = 1, device: int = 0, options: dict | None = None):
        """options: fields of sm_plan_options (kernel variant / tiling chosen explicitly:
        tuning, A/B measurements, tests); None = the plan's own choices"""
        self.width, self.height = int(width), int(height)
        self.num_shifts, self.square_width = int(num_shifts), int(square_width)
        self.border = BORDERS[border] if isinstance(border, str) else int(border)
        self.max_pairs, self.device = int(max_pairs), int(device)
        self._h = C.c_void_p(0)
        opts = capi.PlanOptions.make(**options) if options else None
        check(lib.sm_plan_create_ex(self.device, self.width, self.height, self.num_shifts,
                                    self.square_width, self.border, self.max_pairs,
                                    C.byref(opts) if opts is not None else None, C.byref(self._h)))
        self._dev = torch.device("cuda", self.device)

    def close(self):
        if self._h:
            lib.sm_plan_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers -----------------------------------------------------------
    def describe(self) -> str:
        return lib.sm_plan_describe(self._h).decode()

    def geometry(self) -> dict:
        """The kernel variant and tiling the plan selected (sm_plan_geometry)."""
        g = capi.Geometry()
        check(lib.sm_plan_geometry_sized(self._h, C.byref(g), C.sizeof(g)))
        return {n: getattr(g, n) for n, _ in capi.Geometry._fields_}

    def valu_model(self, pairs: int = 1, want_best: bool = False):
        """VALU wave-instructions of one match launch from the analytic model of
        valu_model.py; None where no coefficients exist for this kernel variant."""
        from . import valu_model
        return valu_model.match_launch(self.geometry(), self.width, self.height, self.num_shifts,
                                       self.border, pairs, want_best)

    def workspace_bytes(self) -> int:
        return int(lib.sm_plan_workspace_bytes(self._h))

    def set_pipelined(self, enabled: bool = True):
        """Let consecutive run() calls overlap (two internal lanes: edges of call i+1 beside
        the match of call i, the head of match i+1 in the tail of match i; give consecutive
        calls their own result maps).  With True the inputs given to run() must be complete in
        memory at call time; with 2 they may still be in flight on the current stream."""
        check(lib.sm_plan_set_pipelined(self._h, int(enabled)))

    def prepare_threshold(self, threshold: float = DEFAULT_THRESHOLD):
        """Threshold-only set-up of find_all_edges (it does this itself on first use)."""
        check(lib.sm_plan_prepare_threshold(self._h, float(threshold), self._stream()))

    def reserve_narrow(self):
        """The int32 staging map that uint8 / uint16 results of the fallback kernels go through, allocated now
        (a no-op for plans whose kernel stores narrow maps itself): keeps the allocation out of timed paths and
        out of stream captures."""
        check(lib.sm_plan_reserve_narrow(self._h))
        self._narrow_ready = True

    def time_kernels(self, capacity: int, every: int = 1):
        """Bracket every `every`-th of the coming match launches (at most `capacity` of
        them) with HIP events on the launch stream."""
        check(lib.sm_plan_time_stride(self._h, int(every)))
        check(lib.sm_plan_time_kernels(self._h, int(capacity)))

    def kernel_ms(self):
        """(mean ms, launches) of the match launches recorded since time_kernels()."""
        ms, n = C.c_double(0), C.c_int(0)
        check(lib.sm_plan_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)

    def _images(self, t: torch.Tensor, dtype, name: str):
        if t.device != self._dev or t.dtype != dtype or not t.is_contiguous():
            raise ValueError(f"{name}: need a contiguous {dtype} tensor on {self._dev}, "
                             f"got {t.dtype} on {t.device}")
        if t.dim() == 2:
            t = t.unsqueeze(0)
        if t.dim() != 3 or t.shape[1] != self.height or t.shape[2] != self.width:
            raise ValueError(f"{name}: shape {tuple(t.shape)} is not (pairs, {self.height}, {self.width})")
        return t

    def _out(self, t, pairs, name, dtype=torch.int32):
        """A caller-supplied result buffer gets the same checks as an input (a wrong
        buffer would otherwise become an out-of-bounds device write)."""
        if t is None:
            return self._new(pairs, dtype)
        t = self._images(t, dtype, name)
        if t.shape[0] < pairs:
            raise ValueError(f"{name}: room for {t.shape[0]} maps, {pairs} pairs requested")
        return t

    def _new(self, pairs, dtype):
        return torch.empty((pairs, self.height, self.width), dtype=dtype, device=self._dev)

    # ---- step 1 ------------------------------------------------------------
    def find_all_edges(self, left, right, threshold=DEFAULT_THRESHOLD, want_edges=True):
        """find_all_edges x2 (src/stereo.cu:27-92,:312-313).  uint8 gray in; fills the
        plan's packed edge workspace; returns the u8 {0,1} edge images if wanted."""
        left = self._images(left, torch.uint8, "left")
        right = self._images(right, torch.uint8, "right")
        pairs = left.shape[0]
        el = self._new(pairs, torch.uint8) if want_edges else None
        er = self._new(pairs, torch.uint8) if want_edges else None
        check(lib.sm_find_edges(self._h, _ptr(left), _ptr(right), float(threshold), pairs,
                                _ptr(el), _ptr(er), self._stream()))
        return (el, er) if want_edges else None

    def load_edges(self, left_edges, right_edges):
        """Start from u8 {0,1} edge images (the arguments of fillup_matches)."""
        le = self._images(left_edges, torch.uint8, "left_edges")
        re = self._images(right_edges, torch.uint8, "right_edges")
        check(lib.sm_load_edges(self._h, _ptr(le), _ptr(re), le.shape[0], self._stream()))
        return le.shape[0]

    # ---- step 2: the hot path ----------------------------------------------
    def match_wta(self, pairs=1, want_best=True, web=None, best=None, web_dtype=torch.int32):
        """fillup_matches + fillup_scores + find_highest_scoring_shifts
        (src/stereo.cu:127-225) in one launch -> (web, best).  web_dtype torch.uint16 /
        torch.uint8 asks for the narrow map (sm_match_wta_typed); int32 is the reference's."""
        web = self._out(web, pairs, "web", web_dtype)
        best = self._out(best, pairs, "best") if want_best else None
        check(lib.sm_match_wta_typed(self._h, pairs, _ptr(web), WEB_TYPES[web_dtype],
                                     _ptr(best if want_best else None), self._stream()))
        return web, (best if want_best else None)

    def run(self, left, right, threshold=DEFAULT_THRESHOLD, want_best=False, web=None, best=None,
            web_dtype=torch.int32):
        """steps 1+2: uint8 pairs in, web (and optionally best) out."""
        left = self._images(left, torch.uint8, "left")
        right = self._images(right, torch.uint8, "right")
        pairs = left.shape[0]
        web = self._out(web, pairs, "web", web_dtype)
        best = self._out(best, pairs, "best") if want_best else None
        if web_dtype != torch.int32 and not getattr(self, "_narrow_ready", False) and \
                not torch.cuda.is_current_stream_capturing():
            self.reserve_narrow()       # (inside a capture the library says what to call first)
        check(lib.sm_run_typed(self._h, _ptr(left), _ptr(right), float(threshold), pairs, _ptr(web),
                               WEB_TYPES[web_dtype], _ptr(best if want_best else None), self._stream()))
        return web, (best if want_best else None)

    def run_after(self, left, right, threshold=DEFAULT_THRESHOLD, inputs_ready=None, want_best=False, web=None,
                  best=None, web_dtype=torch.int32):
        """run() whose only input dependency is `inputs_ready` (a torch.cuda.Event or None = complete now): consecutive
        calls may overlap, and on launches that cannot fill the chip twice over the plan lets them (sm_run_after)."""
        left = self._images(left, torch.uint8, "left")
        right = self._images(right, torch.uint8, "right")
        pairs = left.shape[0]
        web = self._out(web, pairs, "web", web_dtype)
        best = self._out(best, pairs, "best") if want_best else None
        if web_dtype != torch.int32 and not getattr(self, "_narrow_ready", False) and \
                not torch.cuda.is_current_stream_capturing():
            self.reserve_narrow()
        ev = C.c_void_p(inputs_ready.cuda_event) if inputs_ready is not None else C.c_void_p(0)
        check(lib.sm_run_after(self._h, _ptr(left), _ptr(right), float(threshold), pairs, _ptr(web),
                               WEB_TYPES[web_dtype], _ptr(best if want_best else None), self._stream(), ev))
        return web, (best if want_best else None)

    def cost_wta(self, left, right, cost="sad", want_best=True, web=None, best=None):
        """SAD / SSD cost mode on the uint8 images (parity unpinned: the reference has no
        such mode) -> (web, best): arg-min over the shifts, first shift wins."""
        left = self._images(left, torch.uint8, "left")
        right = self._images(right, torch.uint8, "right")
        pairs = left.shape[0]
        if web is None:
            web = self._new(pairs, torch.int32)
        if best is None and want_best:
            best = self._new(pairs, torch.int32)
        check(lib.sm_cost_wta(self._h, _ptr(left), _ptr(right), {"sad": 1, "ssd": 2}[cost], pairs,
                              _ptr(web), _ptr(best), self._stream()))
        return web, best

    def debug_planes(self, pair: int, shift: int):
        """matches-i, score_all-i, scores-i of the reference's debug build."""
        m = torch.empty((self.height, self.width), dtype=torch.uint8, device=self._dev)
        sa = torch.empty((self.height, self.width), dtype=torch.int32, device=self._dev)
        sc = torch.empty_like(sa)
        check(lib.sm_debug_planes(self._h, pair, shift, _ptr(m), _ptr(sa), _ptr(sc), self._stream()))
        return m, sa, sc

    # ---- step 3 ------------------------------------------------------------
    def fill_web_holes(self, web, times=DEFAULT_TIMES):
        """src/stereo.cu:235-256; returns the buffer the reference would return."""
        web = self._images(web, torch.int32, "web").clone()
        tmp = torch.empty_like(web)
        which = C.c_int(0)
        check(lib.sm_fill_web_holes(self._h, _ptr(web), _ptr(tmp), int(times), web.shape[0],
                                    C.byref(which), self._stream()))
        return tmp if which.value else web

    def image_min_max(self, image):
        image = self._images(image, torch.int32, "image")
        mm = torch.empty((image.shape[0], 2), dtype=torch.int32, device=self._dev)
        check(lib.sm_min_max(self._h, _ptr(image), image.shape[0], _ptr(mm), self._stream()))
        return mm

    def draw_contour_map(self, web, num_lines=DEFAULT_LINES):
        """src/stereo.cu:261-285.  Raises StereoHipError(SM_ERR_ZERO_DIV) where the
        reference would divide by zero."""
        web = self._images(web, torch.int32, "web")
        mm = self.image_min_max(web)
        out = self._new(web.shape[0], torch.uint8)
        check(lib.sm_draw_contour_map(self._h, _ptr(web), _ptr(mm), int(num_lines), web.shape[0],
                                      _ptr(out), self._stream()))
        check(lib.sm_plan_status(self._h, self._stream()))
        return out

    def step3(self, web, times=DEFAULT_TIMES, num_lines=DEFAULT_LINES):
        """fill_web_holes + min/max + draw_contour_map with a single synchronisation
        (sm_step3) -> (hole-filled web, contour image, minmax)."""
        web = self._images(web, torch.int32, "web").clone()
        tmp = torch.empty_like(web)
        pairs = web.shape[0]
        mm = torch.empty((pairs, 2), dtype=torch.int32, device=self._dev)
        out = self._new(pairs, torch.uint8)
        which = C.c_int(0)
        check(lib.sm_step3(self._h, _ptr(web), _ptr(tmp), int(times), int(num_lines), pairs, _ptr(mm),
                           _ptr(out), C.byref(which), self._stream()))
        return (tmp if which.value else web), out, mm

    # ---- the whole of algorithm() --------------------------------------------
    def algorithm(self, first, second, params: AlgorithmParams = AlgorithmParams(), step3=True):
        """Stage order of src/stereo.cu:289-347; returns the images the reference dumps
        (minus the per-shift planes: see debug_planes)."""
        if params.square_width != self.square_width:
            raise ValueError("params.square_width differs from the plan's")
        el, er = self.find_all_edges(first, second, params.threshold)
        web1, best = self.match_wta(el.shape[0], want_best=True)
        res = {"edges-1": el, "edges-2": er, "score_best-0": best, "web-1": web1}
        if step3:
            web2 = self.fill_web_holes(web1, params.times)
            res["web-2"] = web2
            res["output-0"] = self.draw_contour_map(web2, params.lines_to_draw)
        return res


__all__ = ["AlgorithmParams", "StereoPlan", "capi"]
