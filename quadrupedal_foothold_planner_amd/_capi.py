"""ctypes binding of libfpe.so (include/fpe.h).  Record layouts are numpy structured dtypes that
mirror the C structs byte for byte; `check_layout()` asserts the sizes against the library."""
import ctypes as C
import os

import numpy as np

from . import build as _build

FPE_OK = 0
FPE_E_INVALID_ARG = -1
FPE_E_NO_MAP = -2
FPE_E_HIP = -3
FPE_E_NO_DEVICE = -4
FPE_E_UNSUPPORTED = -5
FPE_E_NOMEM = -6
FPE_E_SERVICE_FALSE = -7
FPE_POSE_OPT_SUBMAP_FAILED = 1

MAX_POLYGON_VERTICES = 8

PARAMS_DTYPE = np.dtype(
    [
        ("footRadius", "<f4"),
        ("defaultFootholdThreshold", "<f4"),
        ("candidateFootholdThreshold", "<f4"),
        ("searchRadius", "<f4"),
        ("stepLength", "<f4"),
        ("length", "<f4"),
        ("width", "<f4"),
        ("l1", "<f4"),
        ("skew", "<f4"),
        ("RF_FIRST", "<i4"),
        ("h", "<f8"),
        ("lateralDrift", "<f8"),
    ],
    align=True,
)
POSE_DTYPE = np.dtype(
    [("position", "<f8", (3,)), ("gait", "<i4"), ("leg_search_radius", "<f4", (4,)), ("leg_polygon_kind", "<i4", (4,))],
    align=True,
)
FOOTHOLD_DTYPE = np.dtype(
    [("row", "<i4"), ("col", "<i4"), ("x", "<f8"), ("y", "<f8"), ("z", "<f4"), ("valid", "u1"), ("source", "u1"),
     ("foot_id", "u1"), ("gait_cycle_id", "u1")],
    align=True,
)
CENTROID_DTYPE = np.dtype(
    [("x", "<f8"), ("y", "<f8"), ("z", "<f4"), ("row", "<i4"), ("col", "<i4"), ("code", "u1"), ("pad", "u1", (3,))],
    align=True,
)
# fpe_selected_foothold: the 16-byte multi-GPU exchange record
SELECTED_DTYPE = np.dtype(
    [("row", "<i4"), ("col", "<i4"), ("z", "<f4"), ("valid", "u1"), ("source", "u1"), ("foot_id", "u1"), ("gait_cycle_id", "u1")],
    align=True,
)
# fpe_selected_packed: the 8-byte exchange record ((row + 256) | (col + 256) << 14 | valid << 28 | source << 29, z)
PACKED_DTYPE = np.dtype([("cell", "<u4"), ("z", "<f4")], align=True)
PACKED_BIAS = 256
PACKED_MAX_CELLS = 16383 - 2 * PACKED_BIAS


def unpack_selected(packed):
    """fpe_selected_packed records -> SELECTED_DTYPE records (foot_id / gait_cycle_id from the positions [..., n_cycles, 4])."""
    c = packed["cell"]
    out = np.zeros(packed.shape, dtype=SELECTED_DTYPE)
    out["row"] = (c & 0x3FFF).astype(np.int32) - PACKED_BIAS
    out["col"] = ((c >> 14) & 0x3FFF).astype(np.int32) - PACKED_BIAS
    out["z"] = packed["z"]
    out["valid"] = (c >> 28) & 1
    out["source"] = (c >> 29) & 3
    if packed.ndim >= 2:
        out["foot_id"] = np.arange(packed.shape[-1], dtype=np.uint8)
        out["gait_cycle_id"] = np.arange(packed.shape[-2], dtype=np.uint8)[:, None]
    return out


QUERY_DTYPE = np.dtype(
    [("cx", "<f8"), ("cy", "<f8"), ("search_radius", "<f4"), ("n_vertices", "<i4"),
     ("vx", "<f8", (MAX_POLYGON_VERTICES,)), ("vy", "<f8", (MAX_POLYGON_VERTICES,))],
    align=True,
)
MSG_FOOTHOLD_DTYPE = np.dtype(
    [("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("foot_id", "u1"), ("gait_cycle_id", "u1"), ("pad", "u1", (6,))],
    align=True,
)
GLOBAL_FOOTHOLDS_DTYPE = np.dtype(
    [("success", "u1"), ("gait_cycles", "u1"), ("gait_cycles_succeed", "u1"), ("pad", "u1"), ("n_footholds", "<i4"),
     ("footholds", MSG_FOOTHOLD_DTYPE, (4 + 4 * 255,))],
    align=True,
)


TRACK_REPORT_DTYPE = np.dtype(
    [("n_path", "<i4"), ("n_kpi", "<i4"), ("feet_center_path", "<f8", (510, 3)), ("feet_distance", "<f8", (510,)),
     ("cog_speed", "<f8", (510,))],
    align=True,
)
# the opt track (include/fpe.h: fpe_opt_params, fpe_opt_foothold, fpe_opt_cycle)
OPT_PARAMS_DTYPE = np.dtype(
    [("w1", "<f8"), ("w2", "<f8"), ("w3", "<f8"), ("w4", "<f8"), ("wr", "<f8"), ("wc", "<f8"),
     ("use_inequality_constraints", "<i4"), ("reserved", "<i4"), ("ctol", "<f8"),
     ("hip_lower_scale", "<f8"), ("hip_upper_scale", "<f8"), ("skew_lower_scale", "<f8"), ("skew_upper_scale", "<f8"),
     ("lf_current_row0", "<f8"), ("rh_current_row0", "<f8")],
    align=True,
)
OPT_FOOTHOLD_DTYPE = np.dtype(
    [("x", "<f8"), ("y", "<f8"), ("z", "<f4"), ("row", "<i4"), ("col", "<i4"), ("foot_id", "u1"), ("gait_cycle_id", "u1"),
     ("committed", "u1"), ("pad", "u1")],
    align=True,
)
OPT_CYCLE_DTYPE = np.dtype(
    [("gait_top_left", "<i4", (2,)), ("gait_size", "<i4", (2,)), ("nominal_index", "<i4", (8,)), ("centroid_index", "<i4", (8,)),
     ("traversable_row", "<i4", (2, 4)), ("x_lower", "<i4", (8,)), ("x_upper", "<i4", (8,)), ("x", "<i4", (8,)),
     ("minf", "<f8"), ("lf_current_row", "<f8"), ("rh_current_row", "<f8"), ("centroid_code", "u1", (4,)),
     ("gate_failed", "u1"), ("committed", "u1"), ("solver_status", "u1"), ("pad", "u1")],
    align=True,
)
assert OPT_FOOTHOLD_DTYPE.itemsize == 32 and OPT_CYCLE_DTYPE.itemsize == 240 and OPT_PARAMS_DTYPE.itemsize == 112


class MapDesc(C.Structure):
    _fields_ = [
        ("rows", C.c_int32),
        ("cols", C.c_int32),
        ("resolution", C.c_double),
        ("position", C.c_double * 2),
        ("start_index", C.c_int32 * 2),
        ("storage_order", C.c_int32),
    ]


class PlanOut(C.Structure):
    _fields_ = [
        ("nominal", C.c_void_p),
        ("centroid", C.c_void_p),
        ("default_next", C.c_void_p),
        ("cycle_ok", C.c_void_p),
        ("stance", C.c_void_p),
        ("selected", C.c_void_p),
        ("pose_status", C.c_void_p),
        ("selected_packed", C.c_void_p),
    ]


class MultiDeviceIO(C.Structure):
    """fpe_multi_device_io: one device's block of a device-resident multi-GPU plan."""
    _fields_ = [("d_poses", C.c_void_p), ("d_out", PlanOut), ("d_gathered", C.c_void_p), ("stream", C.c_void_p)]


class ServiceGate(C.Structure):
    """fpe_service_gate: verdict of the handler's gate for the last service call of this thread."""
    _fields_ = [("fail_cycle", C.c_uint8), ("fail_kind", C.c_uint8), ("chain_ran", C.c_uint8), ("returned_false", C.c_uint8),
                ("pad", C.c_uint8 * 4), ("lf_current_row", C.c_double), ("rh_current_row", C.c_double)]


GATE_NONE, GATE_CYCLE0, GATE_LATERAL, GATE_BUILD_DEFINED = 0, 1, 2, 3
EXCHANGE_NONE, EXCHANGE_SELECTED, EXCHANGE_PACKED = 0, 1, 2


class OptOut(C.Structure):
    _fields_ = [("footholds", C.c_void_p), ("cycles", C.c_void_p), ("gate_fail_cycle", C.c_void_p), ("rows_after", C.c_void_p)]


class FilterParams(C.Structure):
    """fpe_filter_params (include/fpe.h): the producer's default filter chain."""
    _fields_ = [
        ("normal_radius", C.c_double),
        ("slope_critical", C.c_double),
        ("step_critical", C.c_double),
        ("step_first_radius", C.c_double),
        ("step_second_radius", C.c_double),
        ("step_critical_cells", C.c_int32),
        ("reserved", C.c_int32),
        ("roughness_critical", C.c_double),
        ("roughness_radius", C.c_double),
    ]


ABI_VERSION = 5  # FPE_ABI_VERSION of include/fpe.h: the ctypes structures below mirror that layout
FILTER_LAYERS = ("normal_x", "normal_y", "normal_z", "slope", "step_height", "step", "roughness", "traversability")

# every symbol include/fpe.h declares (tests check the library exports each of them)
EXPORTED_SYMBOLS = [
    "fpe_params_yaml",
    "fpe_params_code_defaults",
    "fpe_create",
    "fpe_destroy",
    "fpe_last_error",
    "fpe_version",
    "fpe_abi_version",
    "fpe_upload_map",
    "fpe_upload_map_device",
    "fpe_map_info",
    "fpe_filter_params_defaults",
    "fpe_traversability",
    "fpe_traversability_device",
    "fpe_set_max_leg_search_radius",
    "fpe_set_tuning",
    "fpe_describe_plan",
    "fpe_host_alloc",
    "fpe_host_free",
    "fpe_plan",
    "fpe_plan_device",
    "fpe_search_legs",
    "fpe_search_legs_device",
    "fpe_plan_service",
    "fpe_plan_service_ex",
    "fpe_plan_service_report",
    "fpe_plan_service_opt",
    "fpe_last_service_gate",
    "fpe_opt_params_yaml",
    "fpe_opt_params_code_defaults",
    "fpe_plan_opt",
    "fpe_plan_opt_device",
    "fpe_multi_create",
    "fpe_multi_destroy",
    "fpe_multi_device_count",
    "fpe_multi_engine",
    "fpe_multi_last_error",
    "fpe_multi_upload_map",
    "fpe_multi_set_tuning",
    "fpe_multi_plan",
    "fpe_multi_shard_range",
    "fpe_multi_plan_device",
    "fpe_multi_stream",
    "fpe_multi_synchronize",
    "fpe_spiral_offsets",
    "fpe_tile_halfwidth",
    "fpe_algorithmic_bytes_per_foothold",
]

_lib = None


class EngineUnavailable(RuntimeError):
    """libfpe.so is missing / not loadable, or no gfx950 device: the product has no CPU path."""


def lib():
    """Load libfpe.so (building it with hipcc first if the in-tree .so is missing or stale)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # FPE_LIB: load a specific prebuilt engine (kernel-variant A/B runs); default = in-tree libfpe.so
        path = os.environ.get("FPE_LIB") or _build.build_engine()
    except Exception as e:  # hipcc missing or compile error: fail loudly — never load a stale library instead
        raise EngineUnavailable(f"cannot build libfpe.so: {e}") from e
    try:
        L = C.CDLL(path)
    except OSError as e:
        raise EngineUnavailable(f"cannot load {path}: {e}") from e
    vp, i32, f32, f64 = C.c_void_p, C.c_int32, C.c_float, C.c_double
    # a library that predates a symbol of include/fpe.h (FPE_LIB pointing at an old scratch build) is "unavailable", not an
    # AttributeError out of ctypes that callers catching EngineUnavailable would not see
    missing = [name for name in EXPORTED_SYMBOLS if not hasattr(L, name)]
    if missing:
        raise EngineUnavailable(f"{path} predates include/fpe.h (ABI version {ABI_VERSION}): it does not export {', '.join(missing[:4])}"
                                + (" ..." if len(missing) > 4 else ""))
    L.fpe_version.restype = C.c_char_p
    L.fpe_abi_version.restype = C.c_int
    if L.fpe_abi_version() != ABI_VERSION:
        raise EngineUnavailable(f"libfpe.so has struct layout version {L.fpe_abi_version()}, this binding mirrors version {ABI_VERSION} of include/fpe.h")
    L.fpe_last_error.restype = C.c_char_p
    L.fpe_last_error.argtypes = [vp]
    L.fpe_params_yaml.argtypes = [vp]
    L.fpe_params_code_defaults.argtypes = [vp]
    L.fpe_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.fpe_destroy.argtypes = [vp]
    L.fpe_upload_map.argtypes = [vp, C.POINTER(MapDesc), vp, vp]
    L.fpe_upload_map_device.argtypes = [vp, C.POINTER(MapDesc), vp, vp, vp]
    L.fpe_filter_params_defaults.argtypes = [C.POINTER(FilterParams)]
    L.fpe_traversability.argtypes = [vp, C.POINTER(MapDesc), C.POINTER(FilterParams), vp, vp, vp]
    L.fpe_traversability_device.argtypes = [vp, C.POINTER(MapDesc), C.POINTER(FilterParams), vp, vp, vp, vp]
    L.fpe_map_info.argtypes = [vp, C.POINTER(MapDesc)]
    L.fpe_set_max_leg_search_radius.argtypes = [vp, f32]
    L.fpe_set_tuning.argtypes = [vp, C.c_char_p, i32]
    L.fpe_describe_plan.argtypes = [vp, vp, C.c_char_p, i32]
    L.fpe_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.fpe_host_free.argtypes = [vp, vp]
    L.fpe_plan.argtypes = [vp, vp, vp, i32, i32, C.POINTER(PlanOut)]
    L.fpe_plan_device.argtypes = [vp, vp, vp, i32, i32, C.POINTER(PlanOut), vp]
    L.fpe_search_legs.argtypes = [vp, vp, vp, i32, vp]
    L.fpe_search_legs_device.argtypes = [vp, vp, vp, i32, vp, vp]
    L.fpe_plan_service.argtypes = [vp, vp, vp, C.c_uint8, vp]
    L.fpe_plan_service_ex.argtypes = [vp, vp, vp, C.c_uint8, vp, vp, vp, vp]
    L.fpe_plan_service_report.argtypes = [vp, vp, vp, C.c_uint8, vp, vp, vp, vp, vp, vp]
    L.fpe_plan_service_opt.argtypes = [vp, vp, vp, vp, C.c_uint8] + [vp] * 9
    L.fpe_last_service_gate.argtypes = [vp, C.POINTER(ServiceGate)]
    L.fpe_opt_params_yaml.argtypes = [vp]
    L.fpe_opt_params_code_defaults.argtypes = [vp]
    L.fpe_plan_opt.argtypes = [vp, vp, vp, vp, i32, i32, vp, C.POINTER(OptOut)]
    L.fpe_plan_opt_device.argtypes = [vp, vp, vp, vp, i32, i32, vp, C.POINTER(OptOut), vp]
    L.fpe_multi_create.argtypes = [vp, i32, C.POINTER(vp)]
    L.fpe_multi_destroy.argtypes = [vp]
    L.fpe_multi_device_count.argtypes = [vp]
    L.fpe_multi_engine.restype = vp
    L.fpe_multi_engine.argtypes = [vp, i32]
    L.fpe_multi_last_error.restype = C.c_char_p
    L.fpe_multi_last_error.argtypes = [vp]
    L.fpe_multi_upload_map.argtypes = [vp, C.POINTER(MapDesc), vp, vp]
    L.fpe_multi_set_tuning.argtypes = [vp, C.c_char_p, i32]
    L.fpe_multi_plan.argtypes = [vp, vp, vp, i32, i32, C.POINTER(PlanOut)]
    L.fpe_multi_shard_range.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.fpe_multi_plan_device.argtypes = [vp, vp, C.POINTER(MultiDeviceIO), i32, i32, i32]
    L.fpe_multi_stream.restype = vp
    L.fpe_multi_stream.argtypes = [vp, i32]
    L.fpe_multi_synchronize.argtypes = [vp]
    L.fpe_spiral_offsets.argtypes = [i32, vp, i32]
    L.fpe_tile_halfwidth.argtypes = [f32, f32, f64]
    L.fpe_algorithmic_bytes_per_foothold.restype = f64
    L.fpe_algorithmic_bytes_per_foothold.argtypes = [f32, f32, f64]
    _lib = L
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def params_yaml():
    p = np.zeros(1, dtype=PARAMS_DTYPE)
    assert lib().fpe_params_yaml(ptr(p)) == FPE_OK
    return p


def params_code_defaults():
    p = np.zeros(1, dtype=PARAMS_DTYPE)
    assert lib().fpe_params_code_defaults(ptr(p)) == FPE_OK
    return p


def opt_params_yaml():
    p = np.zeros(1, dtype=OPT_PARAMS_DTYPE)
    assert lib().fpe_opt_params_yaml(ptr(p)) == FPE_OK
    return p


def opt_params_code_defaults():
    p = np.zeros(1, dtype=OPT_PARAMS_DTYPE)
    assert lib().fpe_opt_params_code_defaults(ptr(p)) == FPE_OK
    return p


def spiral_offsets(n_rings):
    n = lib().fpe_spiral_offsets(n_rings, None, 0)
    if n < 0:
        raise ValueError("n_rings out of range")
    out = np.zeros((n, 3), dtype=np.int32)
    lib().fpe_spiral_offsets(n_rings, ptr(out), n)
    return out


def algorithmic_bytes_per_foothold(search_radius, foot_radius, resolution):
    return float(lib().fpe_algorithmic_bytes_per_foothold(np.float32(search_radius), np.float32(foot_radius), float(resolution)))
