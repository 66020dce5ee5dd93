"""Plan options for the tuning tools.  The LIBRARY reads no environment variable (variants are
chosen through sm_plan_create_ex); the tools accept the short names they always used --
"SM_TILE_H=5,SM_DUO=1" on the command line or in the tool's own environment -- and turn them
into sm_plan_options fields here."""
import ctypes as C
import os

NAMES = {
    "SM_KERNEL": lambda v: {"kernel_family": 1 if v == "popcount" else 0},
    "SM_TILE_H": lambda v: {"tile_h": int(v)},
    "SM_DS": lambda v: {"shifts_per_lane": int(v)},
    "SM_DUO": lambda v: {"workgroup_waves": 2 if int(v) else 1},
    "SM_NO_CAP2": lambda v: {"no_two_wave_cap": int(v)},
    "SM_PATTERN": lambda v: {"priority_pattern": int(v, 16)},
    "SM_EDGES1": lambda v: {"edge_kernel": int(v)},
    "SM_TIMING_RECORDS": lambda v: {"timing_by_records": int(v)},
    "SM_COST_PX": lambda v: {"cost_pixels_per_lane": int(v)},
    "SM_COST_TILE_H": lambda v: {"cost_tile_h": int(v)},
    "SM_COST_KERNEL": lambda v: {"cost_kernel": int(v)},
    "SM_PRIO_CLASS": lambda v: {"priority_class": int(v)},
    "SM_PRIO_ON_CHANGE": lambda v: {"priority_on_change": int(v)},
    "SM_LANE_MERGE": lambda v: {"lane_merge": int(v)},
    "SM_NO_DS4": lambda v: {"no_four_shift_lanes": int(v)},
    "SM_PRIO_UNIT": lambda v: {"priority_unit_log2": int(v)},
    "SM_COST_WAVES": lambda v: {"cost_workgroup_waves": int(v)},
}


def from_spec(spec: dict) -> dict:
    out = {}
    for k, v in spec.items():
        out.update(NAMES[k](v))
    return out


def from_env() -> dict:
    return from_spec({k: os.environ[k] for k in NAMES if k in os.environ})


class PlanOptions(C.Structure):     # sm_plan_options, for tools that load a library by hand
    _fields_ = [("struct_size", C.c_int), ("kernel_family", C.c_int), ("tile_h", C.c_int),
                ("shifts_per_lane", C.c_int), ("workgroup_waves", C.c_int), ("no_two_wave_cap", C.c_int),
                ("priority_pattern", C.c_uint), ("edge_kernel", C.c_int), ("timing_by_records", C.c_int),
                ("cost_pixels_per_lane", C.c_int), ("cost_tile_h", C.c_int), ("cost_kernel", C.c_int),
                ("priority_class", C.c_int), ("priority_on_change", C.c_int), ("lane_merge", C.c_int),
                ("no_four_shift_lanes", C.c_int), ("priority_unit_log2", C.c_int), ("cost_workgroup_waves", C.c_int)]


def struct_from_spec(spec: dict) -> PlanOptions:
    o = PlanOptions(**from_spec(spec))
    o.struct_size = C.sizeof(PlanOptions)
    return o
