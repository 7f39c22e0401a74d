/*
 * fpe.h — C ABI of the MI355X foothold-search engine (libfpe.so).
 *
 * Drop-in boundary for ONE hot path of lukechencqu/quadrupedal_foothold_planner: the per-leg
 * foothold search and per-cycle recurrence inside the `plan_global_footholds` service.
 * File:line citations are relative to the reference checkout
 * (foothold_planner/src/FootholdPlanner.cpp = "cpp", include/foothold_planner/FootholdPlanner.hpp
 * = "hpp").  The reference has no FFI of its own; each entry point names the reference seam it
 * replaces and INTEGRATION.md shows the binding a maintainer adds to FootholdPlanner.cpp.
 *
 * Conventions: every function returns an int status (FPE_OK = 0, negative = error); nothing
 * throws, nothing prints.  Callers own every host buffer; the engine owns device memory.  Inputs
 * are read-only and may be freed on return (synchronous entry points) or once the given stream
 * has passed the call (…_device entry points).  the fpe_upload_map, fpe_plan and fpe_search_legs families
 * may be called concurrently from different threads: a plan uses the map snapshot that was
 * current when it entered (the reference instead races on gridmap_, cpp:506 vs cpp:818).
 */
#ifndef FPE_H
#define FPE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fpe_engine* fpe_handle;

enum fpe_status {
    FPE_OK = 0,
    FPE_E_INVALID_ARG = -1, /* null pointer, non-positive size, non-finite pose, too many vertices … */
    FPE_E_NO_MAP = -2,      /* plan/search before any fpe_upload_map (reference: empty gridmap_) */
    FPE_E_HIP = -3,         /* a HIP runtime call failed; fpe_last_error() has the text */
    FPE_E_NO_DEVICE = -4,   /* no usable gfx950 device / HIP code object not loadable */
    FPE_E_UNSUPPORTED = -5, /* search radius / foot radius too large for the on-chip tile */
    FPE_E_NOMEM = -6,
    FPE_E_SERVICE_FALSE = -7 /* the reference's service handler returns false for this request:
                                getGaitCycleSearchGridMap's getSubmap failed (cpp:920-934, 2345-2349) — for a reason
                                that does not depend on the optimiser (fpe_service_gate; "service_opt_gate" 2 widens it) */
};

/* ROS parameters of the path, with the reference's member types (readParameters cpp:248-314;
 * hpp:609-619, hpp:657-697).  Floats stay floats: the reference promotes them to double at each
 * use and that decides chosen indices (e.g. ceil(double(0.1f)/0.02) = 6 rings, not 5). */
typedef struct fpe_params {
    float footRadius;                 /* cpp:255 */
    float defaultFootholdThreshold;   /* cpp:257 */
    float candidateFootholdThreshold; /* cpp:258 */
    float searchRadius;               /* cpp:261 */
    float stepLength;                 /* cpp:262 */
    float length, width, l1;          /* laikago_kinematics/{length,width,l1}, cpp:285-287 */
    float skew;                       /* laikago_kinematics/skewLength -> isos_.skew, cpp:290 */
    int32_t RF_FIRST;                 /* cpp:264 */
    double h;                         /* h_ (cpp:336), hard-coded 0.01 in the reference */
    double lateralDrift;              /* ajustedPose_[1] += -0.007 per cycle (cpp:1578) */
} fpe_params;

/* yaml values of foothold_planner/config/foothold_planner.yaml:10-64 */
int fpe_params_yaml(fpe_params* out);
/* code defaults of readParameters (cpp:255-290) */
int fpe_params_code_defaults(fpe_params* out);

/* One element of the batch axis.  position = initialPose_ (cpp:293-295).  The remaining fields are
 * build-defined extensions (not in the reference): zeros reproduce the reference exactly. */
typedef struct fpe_pose {
    double position[3];
    int32_t gait;                /* 0 = trot (reference), 1 = 4-phase walk (build-defined) */
    float leg_search_radius[4];  /* RF,RH,LH,LF; <= 0 -> fpe_params.searchRadius */
    int32_t leg_polygon_kind[4]; /* 0 = reference rectangle getSearchPolygon (cpp:2496-2517);
                                    1 = build-defined hexagon */
} fpe_pose;

/* Geometry + storage of the traversability map message handed to gridmapCallback (cpp:504-536).
 * storage_order 0 = the grid_map_msgs/GridMap Float32MultiArray / Eigen::MatrixXf layout
 * (column-major: buffer cell (i,j) at i + j*rows); 1 = row-major.  start_index is the circular
 * buffer origin GridMap::getStartIndex() (= msg.outer_start_index / inner_start_index mapped by
 * GridMapRosConverter).  The engine canonicalises to start index (0,0),
 * row-major, on the device. */
typedef struct fpe_map_desc {
    int32_t rows, cols;     /* getSize()(0), getSize()(1): rows span x, cols span y */
    double resolution;      /* getResolution() */
    double position[2];     /* getPosition() (map centre) */
    int32_t start_index[2]; /* getStartIndex() (row, col) */
    int32_t storage_order;  /* 0 column-major, 1 row-major */
} fpe_map_desc;

/* One nominal-track foothold = result of checkFoothold (cpp:2001-2036, hpp:94-100). */
typedef struct fpe_foothold {
    int32_t row, col; /* chosen grid index: spiral cell, or getIndex(centre) for a default hit; -1 none */
    double x, y;      /* footholdResult.point.x/y (cell centre, or the continuous centre) */
    float z;          /* getFootholdMeanHeight (returns float, cpp:2520); 0 when invalid */
    uint8_t valid;    /* footholdValidation (cpp:2013, 2023) */
    uint8_t source;   /* 0 default-disc hit, 1 spiral candidate, 2 none, 3 radius over the tile bound */
    uint8_t foot_id;  /* Foothold.msg foot_id: 0 RF, 1 RH, 2 LH, 3 LF (cpp:686-698) */
    uint8_t gait_cycle_id; /* Foothold.msg gait_cycle_id = gaitCycleIndex (cpp:1378) */
} fpe_foothold;

/* One centroid-track foothold = result of checkFootholdUseCentroidMethod (cpp:1605-1997). */
typedef struct fpe_centroid_foothold {
    double x, y;
    float z;
    int32_t row, col; /* getIndex(x,y) on the full map; -1 when the result was left untouched */
    uint8_t code;     /* 0 whole region valid, 1 case 1, 2/3 case 2 upper/lower, 4 case 3,
                         5 no case (result (0,0,0)), 6 getSubmap failed (result (0,0,0)) */
    uint8_t pad[3];
} fpe_centroid_foothold;

/* The exchange record of a selected (nominal) foothold, 16 bytes: the chosen grid index, the height
 * and the flag bytes of the nominal record; x / y stay with the rank that planned the pose.  This is what
 * the multi-GPU all-gather moves. */
typedef struct fpe_selected_foothold {
    int32_t row, col;
    float z;
    uint8_t valid, source, foot_id, gait_cycle_id;
} fpe_selected_foothold;

/* The same record in 8 bytes (build-defined; the exchange of a multi-GPU step moves HALF the bytes): one word holding
 * the grid index and the two flags, and the height.  foot_id / gait_cycle_id are positional (record (b, g, leg) sits at
 * ((b * n_cycles) + g) * 4 + leg on every rank).  An index is stored with a bias of FPE_PACKED_BIAS in 14 bits: a default
 * hit carries getIndex(centre) (cpp:2016), which lies up to a foot radius OUTSIDE the map for a centre just over its edge
 * (found by the random campaign: col -3 of a valid foothold), and -1 ("no foothold") needs no code of its own.  Maps of up
 * to FPE_PACKED_MAX_CELLS rows / columns (fpe_plan* fail with FPE_E_UNSUPPORTED when the product is requested on a larger map). */
typedef struct fpe_selected_packed {
    uint32_t cell; /* bits 0-13 row + 256, 14-27 col + 256, bit 28 valid, bits 29-30 source, bit 31 zero */
    float z;
} fpe_selected_packed;
#define FPE_PACKED_BIAS 256
#define FPE_PACKED_MAX_CELLS (16383 - 2 * FPE_PACKED_BIAS)
#define FPE_PACKED_ROW(c) ((int32_t)((c) & 0x3FFFu) - FPE_PACKED_BIAS)
#define FPE_PACKED_COL(c) ((int32_t)(((c) >> 14) & 0x3FFFu) - FPE_PACKED_BIAS)
#define FPE_PACKED_VALID(c) (((c) >> 28) & 1u)
#define FPE_PACKED_SOURCE(c) (((c) >> 29) & 3u)

/* fpe_plan_out.pose_status bits */
#define FPE_POSE_OPT_SUBMAP_FAILED 1u /* getGaitCycleSearchGridMap (cpp:2307-2349) fails in the FIRST gait cycle:
                                         getSubmap(next feet centre, isos_.length x isos_.width) — the reference's
                                         service handler returns false there (cpp:920-934).  This bit of the PLAN kernels
                                         covers cycle 0; later cycles follow the opt track's own feet: fpe_plan_opt*
                                         (fpe_opt_out.gate_fail_cycle; see fpe_service_gate for what the service calls do). */

/* Output buffers of a chained plan; any pointer may be NULL (that product is skipped).
 * Index convention: record (b, g, leg) at ((b * n_cycles) + g) * 4 + leg. */
typedef struct fpe_plan_out {
    fpe_foothold* nominal;           /* [B * n_cycles * 4] — what the service returns (cpp:1588) */
    fpe_centroid_foothold* centroid; /* [B * n_cycles * 4] — global_footholds_centroid (cpp:1448-1462) */
    double* default_next;            /* [B * n_cycles * 4 * 3] default track x,y,z (cpp:1344-1347) */
    uint8_t* cycle_ok;               /* [B * n_cycles] footholdValidation_ per cycle (cpp:1323) */
    double* stance;                  /* [B * 4 * 3] RF/RH/LH/LF_initialPosition_ (cpp:350-378) */
    fpe_selected_foothold* selected; /* [B * n_cycles * 4] 16-byte form of `nominal` (multi-GPU exchange record) */
    uint8_t* pose_status;            /* [B] FPE_POSE_* bits */
    fpe_selected_packed* selected_packed; /* [B * n_cycles * 4] 8-byte form of `selected` (halves the exchange) */
} fpe_plan_out;

/* ---- the "opt" track: cpp:54-148 objective + constraints, cpp:913-1319 per-cycle driver,
 * cpp:1485-1570 commit, cpp:2307-2408 getGaitCycleSearchGridMap, cpp:2557-2568 getMapIndex ---------------------------
 * Reproduced exactly (bit for bit against the oracle): the gait-cycle submap gaitMap_, the four next default
 * positions, nominalIndex, checkFootholdUseCentroidMethod ON gaitMap_ with traversableBeginRow / traversableEndRow,
 * centroidIndex, the integer bounds xBounds, t1..t4, the objective and the eight constraints, positions and heights
 * taken from gaitMap_ at the optimiser's (truncated) x, the commit rule, lfCurrentRow / rhCurrentRow.
 * BUILD-DEFINED: the optimiser.  The reference runs NLopt's LN_COBYLA (yaml:60); NLopt is absent from this image and
 * unpinned by the reference, and COBYLA's iterates are not reproducible.  In its place: exhaustive search of the same
 * objective under the same constraints (tolerance ctol) over the INTEGER points of the same box — the reference
 * truncates x to int before using it (cpp:1287-1312) — feasible points by objective, else the point of least
 * constraint violation; deterministic tie-breaks (oracle/fpo_opt.cpp::solveLattice states the rule).  Against scipy's
 * COBYLA driving the same literal chain (tests/golden/cobyla_vs_lattice.json, build container only): identical x in 29 %
 * of the cycles, median distance 2 rows + 1 column, the handler's gate verdict identical for 94.8 % of 1 280 poses — the
 * opt products are a PLAUSIBLE track, not the reference's, and the gate verdict built on them is right about 19 times in 20
 * against a different COBYLA (against NLopt's: unmeasurable here). */
typedef struct fpe_opt_params {
    double w1, w2, w3, w4, wr, wc;         /* nlopt/w1..wc, cpp:297-303 */
    int32_t use_inequality_constraints;    /* nlopt/useInequalityConstraits, cpp:306 (code default 0, yaml 1) */
    int32_t reserved;
    double ctol;                           /* cpp:34: 1e-2 */
    double hip_lower_scale, hip_upper_scale;   /* cpp:48: 0.9, 1.1 */
    double skew_lower_scale, skew_upper_scale; /* cpp:49: 0.8, 1.2 */
    double lf_current_row0, rh_current_row0;   /* file-scope lfCurrentRow / rhCurrentRow (cpp:36) at entry of the call:
                                                  0 at node start, afterwards whatever the previous call left — the
                                                  adapter carries fpe_opt_out.rows_after / fpe_service_gate.lf/rh_current_row */
} fpe_opt_params;
int fpe_opt_params_yaml(fpe_opt_params* out);          /* yaml:53-63 + cpp:28-51 */
int fpe_opt_params_code_defaults(fpe_opt_params* out); /* cpp:297-307 (constraints off) */

/* RF/RH/LH/LF_footholdResult_opt of one cycle (cpp:1283-1314): record (b, g, leg) at ((b * n_cycles) + g) * 4 + leg. */
typedef struct fpe_opt_foothold {
    double x, y;      /* gaitMap_.getPosition((int)x[2k], (int)x[2k+1]) */
    float z;          /* getFootholdMeanHeight ON gaitMap_ (cpp:1290) */
    int32_t row, col; /* the truncated optimiser variables: an index of gaitMap_ */
    uint8_t foot_id, gait_cycle_id;
    uint8_t committed; /* the cycle committed (cpp:1332): the record is part of global_footholds_opt (cpp:1515-1532) */
    uint8_t pad;
} fpe_opt_foothold;

/* The optimisation problem of one gait cycle of one pose and its solution: record (b, g) at b * n_cycles + g.
 * Index order of the 8-vectors: LF, RH, RF, LH x (row, col) — the reference's x (cpp:50-51, 1058-1059). */
typedef struct fpe_opt_cycle {
    int32_t gait_top_left[2], gait_size[2]; /* gaitMap_ (cpp:2345) inside gridmap_: index of its cell (0,0); rows, cols */
    int32_t nominal_index[8];               /* cpp:965-976 */
    int32_t centroid_index[8];              /* cpp:1030-1041 */
    int32_t traversable_row[2][4];          /* cpp:1009-1013 / cpp:1608-1609: begin, end row x RF,RH,LH,LF (rows of gaitMap_);
                                               0 where the reference leaves its (uninitialised) storage untouched */
    int32_t x_lower[8], x_upper[8];         /* xBounds, cpp:1057-1076 */
    int32_t x[8];                           /* the optimiser's x as the reference uses it: truncated to int */
    double minf;                            /* objective at x */
    double lf_current_row, rh_current_row;  /* the values this cycle's objective and constraints 7-8 used */
    uint8_t centroid_code[4];               /* checkFootholdUseCentroidMethod on gaitMap_, RF,RH,LH,LF (fpe_centroid_foothold.code) */
    uint8_t gate_failed;                    /* getGaitCycleSearchGridMap returned false in THIS cycle (cpp:931-934) */
    uint8_t committed;
    uint8_t solver_status;                  /* 0 feasible optimum; 1 NLopt precondition (lb > ub or x0 outside the box: the
                                               reference's call throws, x stays x0); 2 least-violation point of an
                                               infeasible problem; 3 box too large to enumerate (x0 kept) */
    uint8_t pad;
} fpe_opt_cycle;

/* Outputs of the opt track; any pointer may be NULL.  Cycles from a pose's failing gate on (and every cycle of a
 * walk-gait pose: the opt track is the reference's, i.e. trot only) are zero records. */
typedef struct fpe_opt_out {
    fpe_opt_foothold* footholds; /* [B * n_cycles * 4] */
    fpe_opt_cycle* cycles;       /* [B * n_cycles] */
    uint8_t* gate_fail_cycle;    /* [B] first cycle whose getGaitCycleSearchGridMap failed — the cycle in which the
                                    reference's service handler returns false (cpp:931-934); 255 = none */
    double* rows_after;          /* [B * 2] lfCurrentRow, rhCurrentRow as the call leaves them (cpp:1561-1568: gaitMap_.getIndex
                                    of the committed LF / RH positions; unchanged by cycles that do not commit): what the
                                    NEXT call's fpe_opt_params.lf/rh_current_row0 must be (file-scope globals, cpp:36) */
} fpe_opt_out;

/* One open-loop checkFoothold call (hpp:94-100) with an arbitrary polygon (grid_map::Polygon). */
#define FPE_MAX_POLYGON_VERTICES 8
typedef struct fpe_leg_query {
    double cx, cy;       /* center */
    float search_radius; /* searchRadius */
    int32_t n_vertices;
    double vx[FPE_MAX_POLYGON_VERTICES], vy[FPE_MAX_POLYGON_VERTICES];
} fpe_leg_query;

/* foothold_planner_msgs/Foothold (Foothold.msg:1-3) and foothold_planner_msgs/GlobalFootholds
 * (GlobalFootholds.msg:1-5) as filled by globalFootholdPlan (cpp:591-699, 1378-1396, 1574, 1588):
 * 4 stance entries (gait_cycle_id 0) then 4 per VALID cycle (gait_cycle_id = cycle index). */
typedef struct fpe_msg_foothold {
    double x, y, z;        /* geometry_msgs/Point point */
    uint8_t foot_id;       /* 0 RF, 1 RH, 2 LH, 3 LF */
    uint8_t gait_cycle_id;
    uint8_t pad[6];
} fpe_msg_foothold;
typedef struct fpe_global_footholds {
    uint8_t success;             /* validity of the LAST cycle (cpp:1380, 1574) */
    uint8_t gait_cycles;         /* request.gait_cycles (cpp:592-593) */
    uint8_t gait_cycles_succeed; /* index+1 of the last valid cycle (cpp:1379) */
    uint8_t pad;
    int32_t n_footholds;         /* entries used in `footholds` */
    fpe_msg_foothold footholds[4 + 4 * 255];
} fpe_global_footholds;

/* ---- lifecycle ---------------------------------------------------------------------------- */
/* One engine per process per GPU (device_id = LOCAL_RANK under torch.distributed). */
int fpe_create(int device_id, fpe_handle* out);
int fpe_destroy(fpe_handle h);
/* Build-defined tuning / test knobs of one engine (never read from the environment per call; the
 * environment variables FPE_PLAN_GROUP, FPE_LITERAL_DISCS, FPE_NO_MID_VARIANT, FPE_NO_BITS only seed the
 * defaults once, in fpe_create).  Keys: "plan_group" (0 automatic; 4/8/16/64 lanes per leg, 65 = one
 * wavefront per pose), "literal_discs" (1: force the literal CircleIterator walk), "no_mid_variant"
 * (1: never launch the 3x3-only kernel variants), "no_bits" (1: never launch the bit-window kernels),
 * "service_opt_gate" — what the fpe_plan_service* calls do about the gate of gait cycles >= 1, whose x side follows the
 * opt track's feet, i.e. the BUILD-DEFINED optimiser (see fpe_service_gate below): 2 (DEFAULT) enforce — the opt track's
 * chain runs next to the plan (its own stream) and the call returns FPE_E_SERVICE_FALSE on its verdict too, as the
 * reference's handler does at that gate whatever its optimiser (cpp:920-934); 1 advisory: the chain runs, its verdict is
 * reported by fpe_last_service_gate, the return value stays optimiser-independent; 0: the chain is not run for the gate
 * (latency-critical callers: 48 us instead of 110 us per call; it still runs when an opt product is asked for) and the call
 * returns FPE_E_SERVICE_FALSE only on the optimiser-independent failures.  In modes 0 and 1 a call whose chain stopped at
 * its gate still answers FPE_OK with the nominal / centroid / default products, but the OPT products (message, report, the
 * opt points of the centroid path) come back EMPTY — never the truncated track of an aborted chain — and
 * fpe_last_service_gate names kind FPE_GATE_BUILD_DEFINED and the cycle: a caller that wants the handler's behaviour checks
 * fail_kind, not only the return value.  "service_cycle0_gate_only" (older name): 1 = service_opt_gate 0, 0 = 2.
 * "service_overlap" (default 1): a call that runs the plan AND the opt track for at most four poses launches the opt
 * track's chain on a second stream beside the plan kernel, on nominal cycle flags of 1 (the flags only decide which cycles the
 * chain commits, cpp:1323-1332, and are 1 unless a nominal search fails), and runs it again on the real flags in the call where
 * they differ: the same products as 0 (one kernel after the other), the plan kernel's time off the common call's latency.
 * "service_poll" (default 1): in such a call for ONE pose the chain's last instruction writes a completion word into the
 * call's pinned arena (after a system-scope fence behind its product stores) and the host polls that word instead of waiting
 * for the stream's completion signal — the end-of-kernel processing and the wake-up of a stream wait cost several microseconds
 * of a ~100 us call; bounded: after 5 ms without the word the ordinary stream synchronisation takes over.  0: stream waits only.
 * Thread-safe: every plan / search call copies the knobs once, under the engine's lock, so a concurrent call runs
 * entirely with the values before or entirely with the values after a change (one key per call: callers that change
 * several keys while other threads plan get each key's change at its own moment). */
int fpe_set_tuning(fpe_handle h, const char* key, int32_t value);
const char* fpe_last_error(fpe_handle h); /* thread-local text of the last failure on this thread */
const char* fpe_version(void);
/* Layout version of the structs this header passes by pointer (fpe_plan_out, fpe_opt_out, fpe_params, ... have no size
 * field: fields are only ever APPENDED, and every append bumps this number).  A host compiled against this header checks
 * fpe_abi_version() == FPE_ABI_VERSION once after loading the library — a library that reads a longer struct than the host
 * passes would take whatever lies behind it for a device pointer (csrc/ros_adapter/fpe_ros_adapter.hpp and _capi.py do). */
#define FPE_ABI_VERSION 5
int fpe_abi_version(void);

/* ---- map ingest: replaces GridMapRosConverter::fromMessage in gridmapCallback (cpp:504-536) ---
 * Host pointers, `rows*cols` floats per layer in desc->storage_order.  Uploads both layers to HBM
 * once, canonicalised (row-major, start index 0); the new snapshot becomes current atomically. */
int fpe_upload_map(fpe_handle h, const fpe_map_desc* desc, const float* traversability, const float* elevation);
/* Same with DEVICE pointers (e.g. a map RCCL-broadcast from rank 0); async on `stream`.  Ordering is the
 * engine's job, not the caller's: a plan / search on ANY stream waits (GPU-side) for the upload of the snapshot it
 * uses, and the layer buffers of a snapshot retired while asynchronous launches may still read it are recycled only
 * behind a device synchronisation performed by the NEXT upload — plans never pay for it: the upload also builds the
 * search bit planes for the threshold pairs the previous snapshot was planned with, and a plan that needs planes for
 * a pair never seen before allocates fresh memory rather than wait for a recycled buffer. */
int fpe_upload_map_device(fpe_handle h, const fpe_map_desc* desc, const float* d_traversability,
                          const float* d_elevation, void* stream);
int fpe_map_info(fpe_handle h, fpe_map_desc* out); /* geometry of the current snapshot */

/* ---- the producer of the map: elevation layer -> traversability layer -----------------
 * The reference starts leggedrobotics/traversability_estimation (launch/mapping.launch:12-13,
 * launch/all.launch:21-22, README.md:29) and subscribes to its output (cpp:188); package and filter
 * configuration are not part of the reference, no version is pinned.  These entry points restate that package's
 * published default chain (grid_map_filters NormalVectorsFilter, area method; SlopeFilter, StepFilter,
 * RoughnessFilter; traversability = (1/3)(slope + step + roughness) on float layers) as disc stencils on the
 * device, so that `elevation message -> traversability -> fpe_upload_map_device -> fpe_plan_device` never leaves
 * HBM.  PARITY UNPINNED: the arithmetic contract is the build's own oracle (oracle/fpo_filters.cpp). */
typedef struct fpe_filter_params {
    double normal_radius;        /* NormalVectorsFilter radius [m]                      (0.05) */
    double slope_critical;       /* SlopeFilter critical_value [rad]                    (1.0)  */
    double step_critical;        /* StepFilter critical_value [m]                       (0.12) */
    double step_first_radius;    /* StepFilter first_window_radius [m]                  (0.08) */
    double step_second_radius;   /* StepFilter second_window_radius [m]                 (0.08) */
    int32_t step_critical_cells; /* StepFilter critical_cell_number                     (4)    */
    int32_t reserved;
    double roughness_critical;   /* RoughnessFilter critical_value [m]                  (0.05) */
    double roughness_radius;     /* RoughnessFilter estimation_radius [m]               (0.05) */
} fpe_filter_params;
int fpe_filter_params_defaults(fpe_filter_params* out);
#define FPE_FILTER_LAYERS 8 /* surface_normal_x, _y, _z, traversability_slope, step_height, traversability_step,
                               traversability_roughness, traversability — each rows*cols floats */
/* Host buffers: `elevation` in desc's layout (storage order, start index); `traversability` (required) and `layers`
 * (optional, FPE_FILTER_LAYERS * rows * cols floats) come back CANONICAL (row-major, start index 0).  Synchronous. */
int fpe_traversability(fpe_handle h, const fpe_map_desc* desc, const fpe_filter_params* fp, const float* elevation,
                       float* traversability, float* layers);
/* Device buffers, asynchronous on `stream`; d_layers optional (scratch from the engine's pool when null).
 * Several producers may call on different streams concurrently (the reference's callbacks run under AsyncSpinner(0),
 * foothold_planner_node.cpp:12): no call synchronises the device or blocks the host — the engine keeps up to four
 * internal step-height buffers, each guarded by an event recorded behind the chain that used it last; a fifth concurrent
 * stream waits GPU-side for the least recently used one.  `stream` must outlive the chains queued on it. */
int fpe_traversability_device(fpe_handle h, const fpe_map_desc* desc, const fpe_filter_params* fp, const float* d_elevation,
                              float* d_traversability, float* d_layers, void* stream);

/* Name and shape of the kernel a chained plan with these parameters launches on the current map (evidence for
 * benchmarks and profiles; e.g. "plan_bits_kernel<2, true> (8 lanes per leg, 13 x 13 bit window, 3x3-only fast path)"). */
int fpe_describe_plan(fpe_handle h, const fpe_params* params, char* buf, int32_t n);

/* Build-defined: upper bound of fpe_pose.leg_search_radius the device-resident entry points size
 * their LDS tile for (the host-buffer entry points scan the poses themselves).  Default 0 =
 * fpe_params.searchRadius only; legs asking for more come back invalid with source = 3. */
int fpe_set_max_leg_search_radius(fpe_handle h, float radius);

/* ---- chained plan: replaces the body of the per-cycle loop of globalFootholdPlan (cpp:762-1579)
 * — getDefaultFootholds, getFootholdSearchGridMap, 4x checkFootholdUseCentroidMethod (cpp:818-821),
 * 4x std::thread(checkFoothold) + join (cpp:863-909), the commit rule (cpp:1323-1576) and the
 * lateral drift (cpp:1578) — for B independent initial poses.  The "opt" track (cpp:913-1319) does not feed
 * these products; it is a launch of its own (fpe_plan_opt*). */
int fpe_plan(fpe_handle h, const fpe_params* params, const fpe_pose* poses, int32_t B, int32_t n_cycles,
             const fpe_plan_out* out);
/* Host arrays the GPU can write by DMA (pinned).  fpe_plan / fpe_plan_opt copy a product whose destination lies in such
 * memory — allocated here, by hipHostMalloc or registered with hipHostRegister — from the device straight into it;
 * other destinations are served through the engine's own pinned arena and a copy (overlapped, chunk by chunk).  A ROS
 * adapter that keeps its result arrays across service calls allocates them once with fpe_host_alloc.  Products whose
 * pinned destinations lie directly behind one another in the order of fpe_plan_out's fields (nominal, centroid, default_next,
 * cycle_ok, stance, selected, pose_status; no gap, sizes that are multiples of 256 bytes) leave in ONE transfer: carve them
 * out of one block in that order (a transfer has a fixed cost of some ten microseconds). */
int fpe_host_alloc(fpe_handle h, size_t bytes, void** out);
int fpe_host_free(fpe_handle h, void* p);

/* Device-resident variant: d_poses and every non-NULL pointer of d_out are DEVICE pointers; the
 * launch is asynchronous on `stream` (a hipStream_t; NULL = default stream). */
int fpe_plan_device(fpe_handle h, const fpe_params* params, const fpe_pose* d_poses, int32_t B,
                    int32_t n_cycles, const fpe_plan_out* d_out, void* stream);

/* ---- the opt track of the same batch: replaces cpp:913-1319 + cpp:1485-1568 of the cycle loop ----------------
 * cycle_ok = fpe_plan_out.cycle_ok of the SAME poses / parameters / map (the opt track commits with the nominal
 * track's validity, cpp:1323-1332).  Host variant: cycle_ok may be NULL — the engine then runs the chained plan
 * itself first.  Device variant: asynchronous on `stream`, ordered after the plan launch that wrote d_cycle_ok. */
int fpe_plan_opt(fpe_handle h, const fpe_params* params, const fpe_opt_params* opt, const fpe_pose* poses, int32_t B,
                 int32_t n_cycles, const uint8_t* cycle_ok, const fpe_opt_out* out);
int fpe_plan_opt_device(fpe_handle h, const fpe_params* params, const fpe_opt_params* opt, const fpe_pose* d_poses,
                        int32_t B, int32_t n_cycles, const uint8_t* d_cycle_ok, const fpe_opt_out* d_out, void* stream);

/* ---- open-loop per-leg search: replaces checkFoothold (cpp:2001-2036) one call per query ------ */
int fpe_search_legs(fpe_handle h, const fpe_params* params, const fpe_leg_query* queries, int32_t n,
                    fpe_foothold* out);
int fpe_search_legs_device(fpe_handle h, const fpe_params* params, const fpe_leg_query* d_queries, int32_t n,
                           fpe_foothold* d_out, void* stream);

/* ---- service-shaped call: one pose, response content of plan_global_footholds (cpp:539-1602) --
 * The reference's handler returns false when getGaitCycleSearchGridMap's getSubmap fails (cpp:920-934, 2345-2349) — in
 * ANY gait cycle.  That submap is centred at (x of the opt track's next feet centre, initialPose_[1] + ajustedPose_[1]):
 *   * in gait cycle 0 the feet are the stance: exact, decided by the plan kernels (FPE_POSE_OPT_SUBMAP_FAILED);
 *   * its y side depends on the cycle number only (the lateral drift, cpp:1578): exact in EVERY cycle, decided on the
 *     host with the same grid arithmetic — a request whose y side fails in some cycle g < gait_cycles is refused by the
 *     reference in cycle g at the latest, whatever its optimiser does;
 *   * its x side in cycles >= 1 follows the opt track's feet, i.e. NLopt's COBYLA iterates (cpp:1116-1211), which this
 *     build cannot reproduce (fpe_opt_params): BUILD-DEFINED, advisory by default.
 * These calls return FPE_E_SERVICE_FALSE (response zeroed) on the first two kinds always, on the third under
 * fpe_set_tuning("service_opt_gate", 2) — the default.  fpe_last_service_gate tells the kinds apart. */
typedef struct fpe_service_gate {
    uint8_t fail_cycle;  /* gait cycle in which the handler's gate fails; 255 = it never does */
    uint8_t fail_kind;   /* FPE_GATE_* */
    uint8_t chain_ran;   /* the opt track's chain was run for this call (its verdict is known) */
    uint8_t returned_false; /* the call returned FPE_E_SERVICE_FALSE */
    uint8_t pad[4];
    double lf_current_row, rh_current_row; /* chain_ran: lfCurrentRow / rhCurrentRow as the call leaves them
                                              (fpe_opt_out.rows_after): the adapter carries them into the next call */
} fpe_service_gate;
#define FPE_GATE_NONE 0          /* no failing cycle (x side of cycles >= 1 unknown when chain_ran = 0) */
#define FPE_GATE_CYCLE0 1        /* exact: first gait cycle, stance feet */
#define FPE_GATE_LATERAL 2       /* exact: y side of cycle fail_cycle (lateral drift), optimiser-independent */
#define FPE_GATE_BUILD_DEFINED 3 /* x side of a cycle >= 1: follows the build-defined optimiser's feet */
/* Gate verdict of the last fpe_plan_service* call made on THIS thread with engine h. */
int fpe_last_service_gate(fpe_handle h, fpe_service_gate* out);
int fpe_plan_service(fpe_handle h, const fpe_params* params, const double initial_position[3],
                     uint8_t gait_cycles, fpe_global_footholds* response);

/* Same call, plus the other two result products the node publishes/logs; either
 * extra pointer may be NULL.
 *   centroid: content of global_footholds_centroid for THIS call (cpp:709-727 stance entries,
 *             cpp:1444-1462 per valid cycle; the reference never clears that message between
 *             calls, cpp:715 — appending across calls is left to the adapter).  Its bookkeeping is NOT
 *             the nominal one: success = 1 iff any cycle committed (set false once at cpp:711, true at
 *             cpp:1446, never cleared by a later failed cycle — cpp:1574 touches the nominal message
 *             only), gait_cycles_succeed = last committed cycle + 1, gait_cycles is never written (0);
 *   default_footholds: rows of globalFootholdsResult_.defaultFootholds (cpp:666-671, 1344-1348):
 *             (1 + gait_cycles_succeed') x 12 doubles = RF,RH,LH,LF x (x,y,z), stance row first, one row
 *             per VALID cycle; *n_default_rows receives the number of rows (capacity 1 + gait_cycles). */
int fpe_plan_service_ex(fpe_handle h, const fpe_params* params, const double initial_position[3],
                        uint8_t gait_cycles, fpe_global_footholds* response, fpe_global_footholds* centroid,
                        double* default_footholds, int32_t* n_default_rows);

/* Evaluation products of one track of a service call:
 *   feet_center_path: poses of nominal_feet_center_path / centroid_feet_center_path (cpp:231-232,
 *             1410, 1477): getPolygonCenter of the track's CURRENT feet, one per planned cycle whether
 *             or not it commits (getFootholdSearchGridMap, cpp:2191-2196);
 *   feet_distance / cog_speed: footholdsKPI_ (hpp:732-745), two entries per COMMITTED cycle
 *             (getHipDistance cpp:2571-2584, getCogSpeed cpp:2587-2623 with gaitCycle_ = 1.0, cpp:332). */
typedef struct fpe_track_report {
    int32_t n_path; /* nominal: gait_cycles.  centroid: 2 x gait_cycles — the opt track's getFootholdSearchGridMap call is
                       handed centroidFeetCenterPath too (cpp:946), so every cycle appends the centroid track's centre
                       and then the opt track's */
    int32_t n_kpi;  /* = 2 x committed cycles */
    double feet_center_path[2 * 255][3];
    double feet_distance[2 * 255];
    double cog_speed[2 * 255];
} fpe_track_report;
/* fpe_plan_service_ex plus the per-track reports; every pointer after `response` may be NULL. */
int fpe_plan_service_report(fpe_handle h, const fpe_params* params, const double initial_position[3],
                            uint8_t gait_cycles, fpe_global_footholds* response, fpe_global_footholds* centroid,
                            double* default_footholds, int32_t* n_default_rows, fpe_track_report* nominal_report,
                            fpe_track_report* centroid_report);

/* fpe_plan_service_report plus the opt track's products; every pointer after `response` may be
 * NULL, `opt` NULL = fpe_opt_params_yaml.
 *   opt_msg:    content of global_footholds_opt for THIS call (cpp:737-755 stance entries, cpp:1510-1532 per committed
 *               cycle; bookkeeping as the centroid message: success = any cycle committed, gait_cycles never written);
 *   opt_report: footholdsKPI_.feetDistance_opt / cogSpeed_opt (cpp:1488-1499); feet_center_path holds the opt
 *               track's feet centres (the entries it interleaves into the centroid path);
 *   opt_cycles: [gait_cycles] the per-cycle problems and solutions. */
int fpe_plan_service_opt(fpe_handle h, const fpe_params* params, const fpe_opt_params* opt, const double initial_position[3],
                         uint8_t gait_cycles, fpe_global_footholds* response, fpe_global_footholds* centroid,
                         double* default_footholds, int32_t* n_default_rows, fpe_track_report* nominal_report,
                         fpe_track_report* centroid_report, fpe_global_footholds* opt_msg, fpe_track_report* opt_report,
                         fpe_opt_cycle* opt_cycles);

/* ---- several GPUs in ONE process (fpe_multi.cpp) ----------------------------------------------
 * north_star: "a batch of candidate body trajectories is the parallel axis and shards across the 8 GPUs of one
 * node".  A C++ host that owns every GPU of the node (the ROS node) creates one group: an engine per device, the
 * map replicated by fpe_multi_upload_map, and fpe_multi_plan splitting the pose batch into contiguous blocks (the
 * first B % n devices take one pose more), one resident host thread per device, results written into the caller's
 * arrays at their global positions.  (A group may list one device several times — independent engines — for the
 * host-buffer form; the RCCL form needs distinct devices.)  Same semantics, argument meaning and status codes as the single-device calls they
 * fan out to (fpe_upload_map, fpe_plan: the seam at cpp:863-909 for every pose of the batch).  The
 * one-process-per-GPU deployment (torch.distributed + RCCL all-gather of fpe_plan_out.selected) uses the
 * single-device entry points instead (bench.py, quadrupedal_foothold_planner_amd/dist.py). */
typedef struct fpe_multi* fpe_multi_handle;
int fpe_multi_create(const int32_t* device_ids, int32_t n_devices, fpe_multi_handle* out);
int fpe_multi_destroy(fpe_multi_handle h);
int fpe_multi_device_count(fpe_multi_handle h);
fpe_handle fpe_multi_engine(fpe_multi_handle h, int32_t k); /* engine of device k (for the single-device calls) */
const char* fpe_multi_last_error(fpe_multi_handle h);
int fpe_multi_upload_map(fpe_multi_handle h, const fpe_map_desc* desc, const float* traversability,
                         const float* elevation);
int fpe_multi_set_tuning(fpe_multi_handle h, const char* key, int32_t value);
int fpe_multi_plan(fpe_multi_handle h, const fpe_params* params, const fpe_pose* poses, int32_t B, int32_t n_cycles,
                   const fpe_plan_out* out);
/* Block of device k of a batch of B poses split over n_devices: poses [*first, *first + *count). */
int fpe_multi_shard_range(int32_t B, int32_t k, int32_t n_devices, int32_t* first, int32_t* count);

/* Device-resident form with the xGMI all-gather north_star names, for the C++ host that owns every GPU of the node
 * (replaces the 4 std::thread + join of cpp:863-909 for every pose of the batch, on all devices at once).  io[k]
 * describes device k: its block's poses and products in ITS memory (block = fpe_multi_shard_range(B, k, n)), its
 * stream (NULL: the group's own stream of that device, fpe_multi_stream), and — when record_kind is not
 * FPE_EXCHANGE_NONE — d_gathered: room for the records of the WHOLE batch in pose order, [B * n_cycles * 4] records of
 * the chosen kind.  The call queues, per device, the plan of its block and then ONE fused RCCL collective
 * (ncclGroupStart / ncclAllGather per device / ncclGroupEnd) on the same streams; when B % n != 0 the blocks travel PADDED
 * to ceil(B / n) poses through a staging buffer of the group (in-place all-gather) and n device-local copies put them at
 * their places: d_gathered on every device holds every block's records once its stream has passed the call.  Nothing
 * synchronises the host.  RCCL is loaded at the first gathering call (librccl.so.1); FPE_E_UNSUPPORTED without it. */
#define FPE_EXCHANGE_NONE 0
#define FPE_EXCHANGE_SELECTED 1 /* fpe_selected_foothold, 16 bytes: d_out.selected is the block's contribution */
#define FPE_EXCHANGE_PACKED 2   /* fpe_selected_packed, 8 bytes: d_out.selected_packed */
typedef struct fpe_multi_device_io {
    const fpe_pose* d_poses; /* device k's block of poses */
    fpe_plan_out d_out;      /* device k's products of its block (device pointers; any may be NULL) */
    void* d_gathered;        /* device k: records of the whole batch, or NULL with FPE_EXCHANGE_NONE */
    void* stream;            /* hipStream_t of device k, or NULL */
} fpe_multi_device_io;
int fpe_multi_plan_device(fpe_multi_handle h, const fpe_params* params, const fpe_multi_device_io* io, int32_t B, int32_t n_cycles,
                          int32_t record_kind);
void* fpe_multi_stream(fpe_multi_handle h, int32_t k); /* the group's own stream of device k (a hipStream_t) */
int fpe_multi_synchronize(fpe_multi_handle h);          /* waits for the group's own streams */

/* ---- host-side helpers (no GPU needed) -------------------------------------------------------- */
/* SpiralIterator visiting order as index offsets (di,dj) for rings 0..n_rings (generateRing walk,
 * consumed from the back).  Writes min(count, max_cells) entries of (di, dj, ring); returns count. */
int fpe_spiral_offsets(int32_t n_rings, int32_t* out_di_dj_ring, int32_t max_cells);
/* Half-width (cells) of the LDS tile a (searchRadius, footRadius, resolution) triple needs. */
int fpe_tile_halfwidth(float search_radius, float foot_radius, double resolution);
/* Algorithmic bytes per foothold used for the roofline:
 * 4*W^2 + 8*n_foot + 16 with W = 2*(floor(R/res+0.5)+floor(rf/res))+1. */
double fpe_algorithmic_bytes_per_foothold(float search_radius, float foot_radius, double resolution);

#ifdef __cplusplus
}
#endif
#endif /* FPE_H */
