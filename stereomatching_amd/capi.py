"""ctypes binding of include/stereo_hip.h (libstereo_hip.so).

This is the only way the Python host side reaches the kernels, and it is the
same C ABI a C caller links against.  There is no fallback: if the library is
missing or does not export a declared symbol, importing fails loudly.
"""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

PKG = Path(__file__).resolve().parent
import os as _os
# diagnostics only (tools/wave_timeline.py loads an instrumented build of the same sources)
LIB_PATH = Path(_os.environ["SM_HIP_LIB"]) if _os.environ.get("SM_HIP_LIB") else PKG / "libstereo_hip.so"
HEADER = PKG.parent / "include" / "stereo_hip.h"

SM_OK, SM_ERR_ARG, SM_ERR_HIP, SM_ERR_NOMEM, SM_ERR_ZERO_DIV = range(5)
SM_TOROIDAL, SM_GHOST = 0, 1
SM_WEB_I32, SM_WEB_U16, SM_WEB_U8 = 0, 1, 2
BORDERS = {"toroidal": SM_TOROIDAL, "ghost": SM_GHOST}


class StereoHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[sm error {code}] {message}")
        self.code = code
        self.message = message


def declared_symbols() -> list[str]:
    """Every function include/stereo_hip.h declares."""
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(sm_[a-z0-9_]+)\s*\(", text)))


_vp, _int, _sz, _dbl = C.c_void_p, C.c_int, C.c_size_t, C.c_double
_intp = C.POINTER(C.c_int)

class Geometry(C.Structure):
    """sm_geometry of include/stereo_hip.h"""
    _fields_ = [(n, C.c_int) for n in (
        "kernel", "window", "shifts_per_lane", "shift_lanes", "threads", "tile_w", "tile_h",
        "tiles_x", "tiles_y", "ext_words", "ext_rows", "pad_l", "lds_bytes", "two_wave_variant",
        "edge_rows_per_wave", "waves_per_workgroup", "lane_merge_lds")]


class PlanOptions(C.Structure):
    """sm_plan_options of include/stereo_hip.h (every field 0 = the plan's own choice)"""
    _fields_ = [("struct_size", C.c_int), ("kernel_family", C.c_int), ("tile_h", C.c_int),
                ("shifts_per_lane", C.c_int), ("workgroup_waves", C.c_int), ("no_two_wave_cap", C.c_int),
                ("priority_pattern", C.c_uint), ("edge_kernel", C.c_int), ("timing_by_records", C.c_int),
                ("cost_pixels_per_lane", C.c_int), ("cost_tile_h", C.c_int), ("cost_kernel", C.c_int),
                ("priority_class", C.c_int), ("priority_on_change", C.c_int), ("lane_merge", C.c_int),
                ("no_four_shift_lanes", C.c_int), ("priority_unit_log2", C.c_int), ("cost_workgroup_waves", C.c_int)]

    @classmethod
    def make(cls, **kw):
        o = cls(**kw)
        o.struct_size = C.sizeof(cls)
        return o


_SIGNATURES = {
    "sm_last_error": (C.c_char_p, []),
    "sm_device_count": (_int, [_intp]),
    "sm_malloc": (_int, [_int, _sz, C.POINTER(_vp)]),
    "sm_free": (_int, [_int, _vp]),
    "sm_memcpy_h2d": (_int, [_int, _vp, _vp, _sz]),
    "sm_memcpy_d2h": (_int, [_int, _vp, _vp, _sz]),
    "sm_stream_sync": (_int, [_int, _vp]),
    "sm_host_alloc": (_int, [_sz, C.POINTER(_vp)]),
    "sm_host_free": (_int, [_vp]),
    "sm_memcpy_h2d_async": (_int, [_int, _vp, _vp, _sz, _vp]),
    "sm_memcpy_d2h_async": (_int, [_int, _vp, _vp, _sz, _vp]),
    "sm_stream_create": (_int, [_int, C.POINTER(_vp)]),
    "sm_stream_destroy": (_int, [_int, _vp]),
    "sm_event_create": (_int, [_int, C.POINTER(_vp)]),
    "sm_event_destroy": (_int, [_int, _vp]),
    "sm_event_record": (_int, [_int, _vp, _vp]),
    "sm_stream_wait_event": (_int, [_int, _vp, _vp]),
    "sm_event_sync": (_int, [_int, _vp]),
    "sm_comm_set_rccl_library": (_int, [C.c_char_p]),
    "sm_comm_create": (_int, [_intp, _int, C.POINTER(_vp)]),
    "sm_comm_destroy": (None, [_vp]),
    "sm_comm_size": (_int, [_vp]),
    "sm_broadcast": (_int, [_vp, C.POINTER(_vp), _sz, C.POINTER(_vp)]),
    "sm_gather_maps": (_int, [_vp, C.POINTER(_vp), C.POINTER(_sz), _vp, C.POINTER(_vp)]),
    "sm_plan_create": (_int, [_int, _int, _int, _int, _int, _int, _int, C.POINTER(_vp)]),
    "sm_plan_create_ex": (_int, [_int, _int, _int, _int, _int, _int, _int, C.POINTER(PlanOptions), C.POINTER(_vp)]),
    "sm_plan_destroy": (None, [_vp]),
    "sm_plan_describe": (C.c_char_p, [_vp]),
    "sm_plan_workspace_bytes": (_sz, [_vp]),
    "sm_plan_reserve_narrow": (_int, [_vp]),
    "sm_plan_geometry": (_int, [_vp, C.POINTER(Geometry)]),
    "sm_plan_geometry_sized": (_int, [_vp, C.POINTER(Geometry), _sz]),
    "sm_find_edges": (_int, [_vp, _vp, _vp, _dbl, _int, _vp, _vp, _vp]),
    "sm_load_edges": (_int, [_vp, _vp, _vp, _int, _vp]),
    "sm_match_wta": (_int, [_vp, _int, _vp, _vp, _vp]),
    "sm_match_wta_typed": (_int, [_vp, _int, _vp, _int, _vp, _vp]),
    "sm_run_typed": (_int, [_vp, _vp, _vp, _dbl, _int, _vp, _int, _vp, _vp]),
    "sm_run_after": (_int, [_vp, _vp, _vp, _dbl, _int, _vp, _int, _vp, _vp, _vp]),
    "sm_plan_set_pipelined": (_int, [_vp, _int]),
    "sm_plan_prepare_threshold": (_int, [_vp, C.c_double, _vp]),
    "sm_plan_time_kernels": (_int, [_vp, _int]),
    "sm_plan_time_stride": (_int, [_vp, _int]),
    "sm_plan_kernel_ms": (_int, [_vp, C.POINTER(C.c_double), _intp]),
    "sm_run": (_int, [_vp, _vp, _vp, _dbl, _int, _vp, _vp, _vp]),
    "sm_cost_wta": (_int, [_vp, _vp, _vp, _int, _int, _vp, _vp, _vp]),
    "sm_debug_planes": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp]),
    "sm_debug_edge_table": (_int, [_int, _dbl, _vp, _vp]),
    "sm_debug_edge_table_fast": (_int, [_vp, _dbl, _vp, _intp, _vp]),
    "sm_fill_web_holes": (_int, [_vp, _vp, _vp, _int, _int, _intp, _vp]),
    "sm_min_max": (_int, [_vp, _vp, _int, _vp, _vp]),
    "sm_draw_contour_map": (_int, [_vp, _vp, _vp, _int, _int, _vp, _vp]),
    "sm_plan_status": (_int, [_vp, _vp]),
    "sm_step3": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp, _vp, _intp, _vp]),
}


def _load() -> C.CDLL:
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -m stereomatching_amd.build` (needs hipcc; cross-compiles for gfx950 "
            "without a GPU). There is no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} does not export {missing} declared in {HEADER}")
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int) -> None:
    if rc != SM_OK:
        raise StereoHipError(rc, lib.sm_last_error().decode(errors="replace"))
