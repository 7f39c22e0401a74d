"""Host-side mirror of the reference's interface for the hot path, over the C ABI (include/fpe.h).

`FootholdPlanner` keeps the reference's names for the seam it replaces
(/root/reference/foothold_planner/src/FootholdPlanner.cpp): `gridmapCallback` (cpp:504),
`globalFootholdPlan` (cpp:539), `checkFoothold` (cpp:2001).  The batch entry points `plan` /
`plan_device` are the build-defined batch axis.  There is no CPU path: if
libfpe.so cannot be loaded or no gfx950 GPU is present, construction raises EngineUnavailable.
"""
import contextlib
import ctypes as C

import threading

import numpy as np

from . import _capi
from ._capi import (CENTROID_DTYPE, FOOTHOLD_DTYPE, GLOBAL_FOOTHOLDS_DTYPE, OPT_CYCLE_DTYPE, OPT_FOOTHOLD_DTYPE, OPT_PARAMS_DTYPE,
                    PACKED_DTYPE, POSE_DTYPE, PARAMS_DTYPE, QUERY_DTYPE, SELECTED_DTYPE, TRACK_REPORT_DTYPE, EngineUnavailable, MapDesc,
                    OptOut, PlanOut, ptr)

# products of a chained plan in the order of fpe_plan_out's fields (= the order of the engine's device arena)
PRODUCT_ORDER = ("nominal", "centroid", "default", "cycle_ok", "stance", "selected", "pose_status", "selected_packed")
PRODUCT_FIELDS = {"nominal": "nominal", "centroid": "centroid", "default": "default_next", "cycle_ok": "cycle_ok", "stance": "stance",
                  "selected": "selected", "pose_status": "pose_status", "selected_packed": "selected_packed"}


def product_shapes(B, n_cycles):
    return {
        "nominal": ((B, n_cycles, 4), FOOTHOLD_DTYPE), "centroid": ((B, n_cycles, 4), CENTROID_DTYPE),
        "default": ((B, n_cycles, 4, 3), np.float64), "cycle_ok": ((B, n_cycles), np.uint8),
        "stance": ((B, 4, 3), np.float64), "selected": ((B, n_cycles, 4), SELECTED_DTYPE), "pose_status": ((B,), np.uint8),
        "selected_packed": ((B, n_cycles, 4), PACKED_DTYPE),
    }


class FpeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fpe error {code}: {msg}")
        self.code = code


def make_poses(xyz, gait=0, leg_search_radius=None, leg_polygon_kind=None):
    """Build an fpe_pose array from [B,3] positions (+ optional build-defined extensions)."""
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    poses = np.zeros(xyz.shape[0], dtype=POSE_DTYPE)
    poses["position"] = xyz
    poses["gait"] = gait
    if leg_search_radius is not None:
        poses["leg_search_radius"] = leg_search_radius
    if leg_polygon_kind is not None:
        poses["leg_polygon_kind"] = leg_polygon_kind
    return poses


# fpe_set_tuning's knobs whose engine default is not 0 (tuning() restores these after a with-block)
TUNING_DEFAULTS = {"service_opt_gate": 2, "service_overlap": 1, "service_poll": 1}


class FootholdPlanner:
    """One engine per process per GPU."""

    def __init__(self, device_id=0, params=None):
        self._lib = _capi.lib()
        self._h = C.c_void_p()
        rc = self._lib.fpe_create(int(device_id), C.byref(self._h))
        if rc != _capi.FPE_OK:
            msg = self._lib.fpe_last_error(None).decode()
            self._h = None
            if rc == _capi.FPE_E_NO_DEVICE:
                raise EngineUnavailable(f"fpe_create failed: {msg} (the engine has no CPU fallback)")
            raise FpeError(rc, msg)
        self.params = _capi.params_yaml() if params is None else np.array(params, dtype=PARAMS_DTYPE).reshape(1)
        self.opt_params = _capi.opt_params_yaml()  # nlopt/* of the yaml
        self.device_id = int(device_id)
        self._tuning = {}  # last value set per knob (tuning() restores these, not zeros)

    def close(self):
        if getattr(self, "_h", None):
            for p in getattr(self, "_pinned", []):  # arrays from host_array() must not be used after close()
                self._lib.fpe_host_free(self._h, p)
            self._pinned = []
            self._lib.fpe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != _capi.FPE_OK:
            raise FpeError(rc, self._lib.fpe_last_error(self._h).decode())

    # ---- map ingest (gridmapCallback, cpp:504-536) -------------------------------------------------
    def gridmapCallback(self, traversability, elevation, resolution, position=(0.0, 0.0), start_index=(0, 0),
                        storage_order="row"):
        """Upload both layers once to HBM.  `storage_order`: "row" ((rows, cols) C arrays) or
        "col" (grid_map_msgs column-major buffers, flat or (cols, rows))."""
        trav = np.ascontiguousarray(traversability, dtype=np.float32)
        elev = np.ascontiguousarray(elevation, dtype=np.float32)
        if storage_order == "row":
            rows, cols = trav.shape
        else:
            cols, rows = trav.shape
        d = MapDesc(rows, cols, float(resolution), (C.c_double * 2)(*map(float, position)),
                    (C.c_int32 * 2)(*map(int, start_index)), 1 if storage_order == "row" else 0)
        self._check(self._lib.fpe_upload_map(self._h, C.byref(d), ptr(trav), ptr(elev)))
        self.rows, self.cols, self.resolution = rows, cols, float(resolution)

    def upload_map_device(self, d_trav_ptr, d_elev_ptr, rows, cols, resolution, position=(0.0, 0.0), start_index=(0, 0),
                          storage_order="row", stream=None):
        d = MapDesc(rows, cols, float(resolution), (C.c_double * 2)(*map(float, position)),
                    (C.c_int32 * 2)(*map(int, start_index)), 1 if storage_order == "row" else 0)
        self._check(self._lib.fpe_upload_map_device(self._h, C.byref(d), C.c_void_p(d_trav_ptr), C.c_void_p(d_elev_ptr),
                                                    C.c_void_p(stream or 0)))
        self.rows, self.cols, self.resolution = rows, cols, float(resolution)

    # ---- the producer of the map (launch/mapping.launch:12-13) ----------------------
    def filter_params(self, **overrides):
        fp = _capi.FilterParams()
        self._check(self._lib.fpe_filter_params_defaults(C.byref(fp)))
        for k, v in overrides.items():
            setattr(fp, k, v)
        return fp

    def traversability_from_elevation(self, elevation, resolution, position=(0.0, 0.0), start_index=(0, 0), storage_order="row",
                                      params=None, want_layers=False):
        """Elevation layer (host array in the message's layout) -> traversability layer (rows x cols, canonical) through
        the device filters; with want_layers also the dict of all FPE_FILTER_LAYERS layers."""
        elev = np.ascontiguousarray(elevation, dtype=np.float32)
        rows, cols = elev.shape if storage_order == "row" else elev.shape[::-1]
        d = MapDesc(rows, cols, float(resolution), (C.c_double * 2)(*map(float, position)),
                    (C.c_int32 * 2)(*map(int, start_index)), 1 if storage_order == "row" else 0)
        fp = params if params is not None else self.filter_params()
        trav = np.empty((rows, cols), np.float32)
        layers = np.empty((len(_capi.FILTER_LAYERS), rows, cols), np.float32) if want_layers else None
        self._check(self._lib.fpe_traversability(self._h, C.byref(d), C.byref(fp), ptr(elev), ptr(trav),
                                                 ptr(layers) if want_layers else None))
        if want_layers:
            return trav, {name: layers[k] for k, name in enumerate(_capi.FILTER_LAYERS)}
        return trav

    def traversability_device(self, d_elev_ptr, d_trav_ptr, rows, cols, resolution, position=(0.0, 0.0), params=None,
                              d_layers_ptr=0, stream=None):
        """Device-resident form (canonical row-major layers), asynchronous on `stream`."""
        d = MapDesc(rows, cols, float(resolution), (C.c_double * 2)(*map(float, position)), (C.c_int32 * 2)(0, 0), 1)
        fp = params if params is not None else self.filter_params()
        self._check(self._lib.fpe_traversability_device(self._h, C.byref(d), C.byref(fp), C.c_void_p(d_elev_ptr),
                                                        C.c_void_p(d_trav_ptr), C.c_void_p(d_layers_ptr or 0), C.c_void_p(stream or 0)))

    def map_info(self):
        d = MapDesc()
        self._check(self._lib.fpe_map_info(self._h, C.byref(d)))
        return {"rows": d.rows, "cols": d.cols, "resolution": d.resolution, "position": tuple(d.position)}

    def set_max_leg_search_radius(self, r):
        self._check(self._lib.fpe_set_max_leg_search_radius(self._h, np.float32(r)))

    def describe_plan(self):
        """Name and shape of the kernel a chained plan launches with the current parameters and map."""
        buf = C.create_string_buffer(256)
        self._check(self._lib.fpe_describe_plan(self._h, ptr(self.params), buf, 256))
        return buf.value.decode()

    def set_tuning(self, **kw):
        """fpe_set_tuning: plan_group, literal_discs, no_mid_variant, no_bits, service_opt_gate, service_overlap, service_poll (build-defined test / tuning knobs)."""
        for k, v in kw.items():
            self._check(self._lib.fpe_set_tuning(self._h, k.encode(), int(v)))
            self._tuning[k] = int(v)

    @contextlib.contextmanager
    def tuning(self, **kw):
        """Set knobs for the duration of a with-block, then restore the values they had through this object (a knob
        this object never set goes back to 0, the automatic default; knobs seeded from the environment in fpe_create are
        not visible here — set them through set_tuning instead when with-blocks are used)."""
        before = {k: self._tuning.get(k, TUNING_DEFAULTS.get(k, 0)) for k in kw}
        self.set_tuning(**kw)
        try:
            yield self
        finally:
            self.set_tuning(**before)

    # ---- pinned host arrays (fpe_host_alloc): results are written into them by DMA, no copy-out ----------
    def host_array(self, shape, dtype):
        """A numpy array over pinned host memory of the engine (kept alive by the planner until close())."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._check(self._lib.fpe_host_alloc(self._h, max(n, 1), C.byref(p)))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        buf = (C.c_ubyte * max(n, 1)).from_address(p.value)
        a = np.frombuffer(buf, dtype=np.uint8, count=n).view(dtype).reshape(shape)
        a[...] = np.zeros((), dtype)
        return a

    def plan_outputs(self, B, n_cycles, products=("nominal", "centroid", "default", "cycle_ok", "stance", "selected", "pose_status"),
                     pinned=False):
        shapes = product_shapes(B, n_cycles)
        if not pinned:
            return {k: np.zeros(shapes[k][0], dtype=shapes[k][1]) for k in products}
        # ONE pinned block, the products behind one another in the order of the engine's device arena (each rounded up to
        # 256 bytes as there): fpe_plan then moves neighbours without padding in between in one DMA transfer
        order = [k for k in PRODUCT_ORDER if k in products]
        sizes = [int(np.prod(shapes[k][0])) * np.dtype(shapes[k][1]).itemsize for k in order]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 255) & ~255
        arena = self.host_array((max(total, 1),), np.uint8)
        return {k: arena[o:o + n].view(shapes[k][1]).reshape(shapes[k][0]) for k, o, n in zip(order, offs, sizes)}

    # ---- chained plan, host buffers ------------------------------------------------------------------
    def plan(self, poses, n_cycles, products=("nominal", "centroid", "default", "cycle_ok", "stance", "selected", "pose_status"),
             out=None):
        """fpe_plan with host buffers.  `out`: a dict returned by an earlier call with the same shapes (timing loops
        reuse the arrays instead of allocating ~100 B per foothold per call)."""
        poses = np.ascontiguousarray(poses, dtype=POSE_DTYPE)
        B = poses.shape[0]
        shapes = product_shapes(B, n_cycles)
        fields = PRODUCT_FIELDS
        if out is None:
            out = {k: np.zeros(shapes[k][0], dtype=shapes[k][1]) for k in products}
        else:  # the arrays of `out` are the products asked for (whatever `products` says)
            products = tuple(k for k in PRODUCT_ORDER if k in out)
            assert len(products) == len(out), f"unknown product in out: {sorted(set(out) - set(products))}"
        # the checked argument block of an `out` dict is kept for its next use (timing loops call with the same arrays: comparing
        # seven structured dtypes and taking seven pointers costs more Python time than the engine needs for its launches)
        key = (id(out), B, int(n_cycles), products) + tuple(out[k].ctypes.data for k in products)
        cached = getattr(self, "_plan_out_cache", None)
        if cached is not None and cached[0] == key:
            po = cached[1]
        else:
            po = PlanOut()
            for k in products:
                assert out[k].shape == shapes[k][0] and out[k].dtype == shapes[k][1]
                setattr(po, fields[k], ptr(out[k]))
            self._plan_out_cache = (key, po, out)
        self._check(self._lib.fpe_plan(self._h, ptr(self.params), ptr(poses), B, int(n_cycles), C.byref(po)))
        return out

    # ---- chained plan, device-resident (torch tensors / raw pointers) ----------------------------------
    def plan_device(self, d_poses_ptr, B, n_cycles, d_nominal_ptr=0, d_centroid_ptr=0, d_default_ptr=0,
                    d_cycle_ok_ptr=0, d_stance_ptr=0, stream=0, d_selected_ptr=0, d_pose_status_ptr=0, d_selected_packed_ptr=0):
        po = PlanOut(d_nominal_ptr or None, d_centroid_ptr or None, d_default_ptr or None, d_cycle_ok_ptr or None,
                     d_stance_ptr or None, d_selected_ptr or None, d_pose_status_ptr or None, d_selected_packed_ptr or None)
        self._check(self._lib.fpe_plan_device(self._h, ptr(self.params), C.c_void_p(d_poses_ptr), int(B), int(n_cycles),
                                              C.byref(po), C.c_void_p(stream or 0)))

    # ---- the opt track of a batch (cpp:913-1319, 1485-1568; build-defined optimiser) ---------------------------
    def plan_opt(self, poses, n_cycles, cycle_ok=None):
        """fpe_plan_opt with host buffers; cycle_ok: the nominal plan's flags [B, n_cycles] (None: the engine plans first).
        Returns {"footholds" [B, n, 4], "cycles" [B, n], "gate_fail_cycle" [B], "rows_after" [B, 2]}."""
        poses = np.ascontiguousarray(poses, dtype=POSE_DTYPE)
        B = poses.shape[0]
        out = {"footholds": np.zeros((B, n_cycles, 4), OPT_FOOTHOLD_DTYPE), "cycles": np.zeros((B, n_cycles), OPT_CYCLE_DTYPE),
               "gate_fail_cycle": np.zeros(B, np.uint8), "rows_after": np.zeros((B, 2), np.float64)}
        ok = None if cycle_ok is None else np.ascontiguousarray(cycle_ok, dtype=np.uint8).reshape(B, n_cycles)
        oo = OptOut(ptr(out["footholds"]), ptr(out["cycles"]), ptr(out["gate_fail_cycle"]), ptr(out["rows_after"]))
        self._check(self._lib.fpe_plan_opt(self._h, ptr(self.params), ptr(self.opt_params), ptr(poses), B, int(n_cycles), ptr(ok),
                                           C.byref(oo)))
        return out

    def plan_opt_device(self, d_poses_ptr, B, n_cycles, d_cycle_ok_ptr, d_footholds_ptr=0, d_cycles_ptr=0, d_gate_ptr=0, stream=0):
        oo = OptOut(d_footholds_ptr or None, d_cycles_ptr or None, d_gate_ptr or None)
        self._check(self._lib.fpe_plan_opt_device(self._h, ptr(self.params), ptr(self.opt_params), C.c_void_p(d_poses_ptr), int(B),
                                                  int(n_cycles), C.c_void_p(d_cycle_ok_ptr), C.byref(oo), C.c_void_p(stream or 0)))

    # ---- open-loop per-leg search (checkFoothold, cpp:2001-2036) -----------------------------------------
    def checkFoothold(self, queries):
        queries = np.ascontiguousarray(queries, dtype=QUERY_DTYPE)
        out = np.zeros(queries.shape[0], dtype=FOOTHOLD_DTYPE)
        self._check(self._lib.fpe_search_legs(self._h, ptr(self.params), ptr(queries), queries.shape[0], ptr(out)))
        return out

    def search_legs_device(self, d_queries_ptr, n, d_out_ptr, stream=0):
        self._check(self._lib.fpe_search_legs_device(self._h, ptr(self.params), C.c_void_p(d_queries_ptr), int(n),
                                                     C.c_void_p(d_out_ptr), C.c_void_p(stream or 0)))

    # ---- the service (globalFootholdPlan, cpp:539-1602): response content for one pose ---------------------
    @staticmethod
    def _msg(m):
        n = int(m["n_footholds"])
        return {
            "success": bool(m["success"]),
            "gait_cycles": int(m["gait_cycles"]),
            "gait_cycles_succeed": int(m["gait_cycles_succeed"]),
            "footholds": m["footholds"][:n].copy(),
        }

    def last_service_gate(self):
        """fpe_last_service_gate: verdict of the handler's gate for this thread's last globalFootholdPlan call."""
        g = _capi.ServiceGate()
        self._check(self._lib.fpe_last_service_gate(self._h, C.byref(g)))
        return {"fail_cycle": int(g.fail_cycle), "fail_kind": int(g.fail_kind), "chain_ran": bool(g.chain_ran),
                "returned_false": bool(g.returned_false), "lf_current_row": float(g.lf_current_row), "rh_current_row": float(g.rh_current_row)}

    @staticmethod
    def _report(r):
        return {"path": r["feet_center_path"][: int(r["n_path"])].copy(),
                "feet_distance": r["feet_distance"][: int(r["n_kpi"])].copy(),
                "cog_speed": r["cog_speed"][: int(r["n_kpi"])].copy()}

    def globalFootholdPlan(self, gait_cycles, initial_position, all_tracks=False):
        """Response content of the service, or False where the reference's handler returns false
        (getGaitCycleSearchGridMap fails, cpp:920-934: in the first gait cycle, or on its lateral side in any cycle — the x side of
        later cycles follows the build-defined optimiser and refuses under set_tuning(service_opt_gate=2), the default; last_service_gate()
        tells the kinds apart); with all_tracks also the centroid message, the default-track
        rows (global_footholds_centroid, globalFootholdsResult_.defaultFootholds), and per track the
        feet-centre path and KPIs (nominal/centroid_feet_center_path, footholdsKPI_)."""
        if not all_tracks:
            # the latency path: message and position buffers of the planner's own, their addresses and the parameter block's taken once
            # (numpy's .ctypes and a zeroed 5 KB message per call were ~4 us of a 105 us call); the library fills every field it reports
            tls = self.__dict__.get("_svc_tls")
            if tls is None:
                tls = self.__dict__.setdefault("_svc_tls", threading.local())
            sv = getattr(tls, "sv", None)  # (per thread: the engine serves concurrent callers of one handle)
            if sv is None or sv[4] is not self.params:
                m, q = np.zeros(1, dtype=GLOBAL_FOOTHOLDS_DTYPE), np.zeros(3, dtype=np.float64)
                sv = tls.sv = (m, q, ptr(m), ptr(q), self.params, ptr(self.params),
                               m["success"], m["gait_cycles"], m["gait_cycles_succeed"], m["n_footholds"], m["footholds"][0])  # field views, made once
            sv[1][:] = initial_position
            rc = self._lib.fpe_plan_service(self._h, sv[5], sv[3], int(gait_cycles) & 0xFF, sv[2])
            if rc == _capi.FPE_E_SERVICE_FALSE:
                return False  # the reference's handler returns false here (cpp:931-934): the ROS call fails
            if rc != _capi.FPE_OK:
                self._check(rc)
            return {"success": bool(sv[6][0]), "gait_cycles": int(sv[7][0]), "gait_cycles_succeed": int(sv[8][0]),
                    "footholds": sv[10][: int(sv[9][0])].copy()}  # (= self._msg(sv[0][0]))
        msg = np.zeros(1, dtype=GLOBAL_FOOTHOLDS_DTYPE)
        pos = np.ascontiguousarray(initial_position, dtype=np.float64).reshape(3)
        cen = np.zeros(1, dtype=GLOBAL_FOOTHOLDS_DTYPE)
        optm = np.zeros(1, dtype=GLOBAL_FOOTHOLDS_DTYPE)
        dflt = np.zeros((1 + int(gait_cycles), 4, 3), dtype=np.float64)
        nrows = C.c_int32(0)
        rep = np.zeros(3, dtype=TRACK_REPORT_DTYPE)
        cyc = np.zeros(max(int(gait_cycles), 1), dtype=OPT_CYCLE_DTYPE)
        rc = self._lib.fpe_plan_service_opt(self._h, ptr(self.params), ptr(self.opt_params), ptr(pos), int(gait_cycles) & 0xFF,
                                            ptr(msg), ptr(cen), ptr(dflt), C.cast(C.byref(nrows), C.c_void_p), ptr(rep[0:1]),
                                            ptr(rep[1:2]), ptr(optm), ptr(rep[2:3]), ptr(cyc))
        if rc == _capi.FPE_E_SERVICE_FALSE:
            return False
        self._check(rc)
        out = self._msg(msg[0])
        out["centroid"] = self._msg(cen[0])
        out["default_footholds"] = dflt[: nrows.value].copy()
        out["report"] = self._report(rep[0])
        out["centroid"]["report"] = self._report(rep[1])
        out["opt"] = self._msg(optm[0])  # global_footholds_opt (cpp:221, 1510-1532)
        out["opt"]["report"] = self._report(rep[2])
        out["opt"]["cycles"] = cyc[: int(gait_cycles)].copy()
        return out


class MultiFootholdPlanner:
    """Several GPUs in ONE process behind the C ABI (fpe_multi_*): the map is replicated on every device and a pose
    batch is split into contiguous blocks, one host thread per device (include/fpe.h, "several GPUs")."""

    def __init__(self, device_ids, params=None):
        self._lib = _capi.lib()
        ids = np.ascontiguousarray(device_ids, dtype=np.int32)
        self._h = C.c_void_p()
        rc = self._lib.fpe_multi_create(ptr(ids), ids.size, C.byref(self._h))
        if rc != _capi.FPE_OK:
            msg = self._lib.fpe_multi_last_error(None).decode()
            self._h = None
            if rc == _capi.FPE_E_NO_DEVICE:
                raise EngineUnavailable(f"fpe_multi_create failed: {msg} (the engine has no CPU fallback)")
            raise FpeError(rc, msg)
        self.params = _capi.params_yaml() if params is None else np.array(params, dtype=PARAMS_DTYPE).reshape(1)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fpe_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != _capi.FPE_OK:
            raise FpeError(rc, self._lib.fpe_multi_last_error(self._h).decode())

    @property
    def device_count(self):
        return int(self._lib.fpe_multi_device_count(self._h))

    def set_tuning(self, **kw):
        for k, v in kw.items():
            self._check(self._lib.fpe_multi_set_tuning(self._h, k.encode(), int(v)))

    def gridmapCallback(self, traversability, elevation, resolution, position=(0.0, 0.0)):
        trav = np.ascontiguousarray(traversability, dtype=np.float32)
        elev = np.ascontiguousarray(elevation, dtype=np.float32)
        rows, cols = trav.shape
        d = MapDesc(rows, cols, float(resolution), (C.c_double * 2)(*map(float, position)), (C.c_int32 * 2)(0, 0), 1)
        self._check(self._lib.fpe_multi_upload_map(self._h, C.byref(d), ptr(trav), ptr(elev)))

    def engine(self, k):
        return self._lib.fpe_multi_engine(self._h, int(k))

    def shard_range(self, B, k):
        first, count = C.c_int32(0), C.c_int32(0)
        rc = self._lib.fpe_multi_shard_range(int(B), int(k), self.device_count, C.byref(first), C.byref(count))
        if rc != _capi.FPE_OK:
            raise FpeError(rc, "bad shard arguments")
        return first.value, first.value + count.value

    def stream(self, k):
        return self._lib.fpe_multi_stream(self._h, int(k))

    def synchronize(self):
        self._check(self._lib.fpe_multi_synchronize(self._h))

    def plan_device(self, B, n_cycles, ios, record_kind=_capi.EXCHANGE_SELECTED):
        """fpe_multi_plan_device.  `ios`: per device a dict {"d_poses": ptr, "d_gathered": ptr, "stream": ptr or 0, and any of the
        product names of PRODUCT_ORDER: ptr} of DEVICE pointers on that device."""
        arr = (_capi.MultiDeviceIO * len(ios))()
        for k, d in enumerate(ios):
            arr[k].d_poses = d["d_poses"]
            for name, field in PRODUCT_FIELDS.items():
                if d.get(name):
                    setattr(arr[k].d_out, field, d[name])
            arr[k].d_gathered = d.get("d_gathered") or None
            arr[k].stream = d.get("stream") or None
        self._check(self._lib.fpe_multi_plan_device(self._h, ptr(self.params), arr, int(B), int(n_cycles), int(record_kind)))

    def plan(self, poses, n_cycles):
        poses = np.ascontiguousarray(poses, dtype=POSE_DTYPE)
        B = poses.shape[0]
        out = {
            "nominal": np.zeros((B, n_cycles, 4), dtype=FOOTHOLD_DTYPE), "centroid": np.zeros((B, n_cycles, 4), dtype=CENTROID_DTYPE),
            "default": np.zeros((B, n_cycles, 4, 3), dtype=np.float64), "cycle_ok": np.zeros((B, n_cycles), dtype=np.uint8),
            "stance": np.zeros((B, 4, 3), dtype=np.float64), "selected": np.zeros((B, n_cycles, 4), dtype=SELECTED_DTYPE),
            "pose_status": np.zeros(B, dtype=np.uint8),
        }
        po = PlanOut(ptr(out["nominal"]), ptr(out["centroid"]), ptr(out["default"]), ptr(out["cycle_ok"]), ptr(out["stance"]),
                     ptr(out["selected"]), ptr(out["pose_status"]))
        self._check(self._lib.fpe_multi_plan(self._h, ptr(self.params), ptr(poses), B, int(n_cycles), C.byref(po)))
        return out
