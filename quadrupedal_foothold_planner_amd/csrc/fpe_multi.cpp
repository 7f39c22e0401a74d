// fpe_multi.cpp — several GPUs behind the C ABI (include/fpe.h, "multi-device" section).
//
// north_star: "Host code stays C++/ROS calling the kernels through a thin C-ABI shim; a batch of candidate body
// trajectories is the parallel axis and shards across the 8 GPUs of one node".  A C++ host (the ROS node) that owns
// ALL the GPUs of a node in ONE process uses this group handle: one engine per device, the map replicated on every
// device, the pose batch split into contiguous blocks (the same rule as quadrupedal_foothold_planner_amd/dist.py:
// the first B % n shards get one pose more), one host thread per device, results written straight into the caller's
// arrays at their global positions — with host buffers the "all-gather" is the shards' D2H copies landing side by
// side.  (One process PER GPU with RCCL is the other deployment: torch.distributed + fpe_plan_device, bench.py.)
// Built on the single-device entry points only; no kernel code here.
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fpe.h"

struct fpe_multi {
    std::vector<fpe_handle> engines;
    std::vector<int> devices;
    std::mutex mu;
    std::string err;
};

namespace {

thread_local std::string g_merr;

int mfail(fpe_multi* h, int code, const std::string& msg) {
    g_merr = msg;
    if (h) {
        std::lock_guard<std::mutex> lk(h->mu);
        h->err = msg;
    }
    return code;
}

// contiguous block split: the first (total % world) shards get one extra element (dist.shard_range)
void shard_range(long total, int rank, int world, long* lo, long* hi) {
    const long base = total / world, rem = total % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}

// Run fn(k) for k = 0..n-1 on one thread each; returns the first non-OK status and its message.
template <class F>
int for_each_device(fpe_multi* h, F fn) {
    const int n = static_cast<int>(h->engines.size());
    std::vector<int> rc(n, FPE_OK);
    std::vector<std::string> msg(n);
    std::vector<std::thread> th;
    for (int k = 1; k < n; ++k)
        th.emplace_back([&, k]() {
            rc[k] = fn(k);
            if (rc[k] != FPE_OK) msg[k] = fpe_last_error(h->engines[k]);  // thread-local text of THIS worker
        });
    rc[0] = fn(0);
    if (rc[0] != FPE_OK) msg[0] = fpe_last_error(h->engines[0]);
    for (auto& t : th) t.join();
    for (int k = 0; k < n; ++k)
        if (rc[k] != FPE_OK) return mfail(h, rc[k], "device " + std::to_string(h->devices[k]) + ": " + msg[k]);
    return FPE_OK;
}

}  // namespace

extern "C" {

int fpe_multi_create(const int32_t* device_ids, int32_t n_devices, fpe_multi_handle* out) {
    if (!out) return mfail(nullptr, FPE_E_INVALID_ARG, "null out handle");
    *out = nullptr;
    if (!device_ids || n_devices <= 0 || n_devices > 64) return mfail(nullptr, FPE_E_INVALID_ARG, "bad device list");
    fpe_multi* h = new (std::nothrow) fpe_multi();
    if (!h) return mfail(nullptr, FPE_E_NOMEM, "out of host memory");
    for (int k = 0; k < n_devices; ++k) {
        fpe_handle e = nullptr;
        const int rc = fpe_create(device_ids[k], &e);
        if (rc != FPE_OK) {
            const std::string msg = std::string("fpe_create(device ") + std::to_string(device_ids[k]) + "): " + fpe_last_error(nullptr);
            for (fpe_handle x : h->engines) fpe_destroy(x);
            delete h;
            return mfail(nullptr, rc, msg);
        }
        h->engines.push_back(e);
        h->devices.push_back(device_ids[k]);
    }
    *out = h;
    return FPE_OK;
}

int fpe_multi_destroy(fpe_multi_handle h) {
    if (!h) return FPE_OK;
    for (fpe_handle e : h->engines) fpe_destroy(e);
    delete h;
    return FPE_OK;
}

int fpe_multi_device_count(fpe_multi_handle h) { return h ? static_cast<int>(h->engines.size()) : 0; }

fpe_handle fpe_multi_engine(fpe_multi_handle h, int32_t k) {
    if (!h || k < 0 || k >= static_cast<int>(h->engines.size())) return nullptr;
    return h->engines[k];
}

const char* fpe_multi_last_error(fpe_multi_handle h) {
    if (!h) return g_merr.c_str();
    std::lock_guard<std::mutex> lk(h->mu);
    g_merr = h->err;
    return g_merr.c_str();
}

int fpe_multi_upload_map(fpe_multi_handle h, const fpe_map_desc* desc, const float* traversability, const float* elevation) {
    if (!h) return mfail(nullptr, FPE_E_INVALID_ARG, "null handle");
    return for_each_device(h, [&](int k) { return fpe_upload_map(h->engines[k], desc, traversability, elevation); });
}

int fpe_multi_set_tuning(fpe_multi_handle h, const char* key, int32_t value) {
    if (!h) return mfail(nullptr, FPE_E_INVALID_ARG, "null handle");
    for (fpe_handle e : h->engines) {
        const int rc = fpe_set_tuning(e, key, value);
        if (rc != FPE_OK) return mfail(h, rc, fpe_last_error(e));
    }
    return FPE_OK;
}

int fpe_multi_plan(fpe_multi_handle h, const fpe_params* params, const fpe_pose* poses, int32_t B, int32_t n_cycles,
                   const fpe_plan_out* out) {
    if (!h || !params || !poses || !out) return mfail(h, FPE_E_INVALID_ARG, "null argument");
    if (B <= 0 || n_cycles <= 0 || n_cycles > 255) return mfail(h, FPE_E_INVALID_ARG, "B and n_cycles must be in [1, ..] / [1, 255]");
    const int n = static_cast<int>(h->engines.size());
    return for_each_device(h, [&](int k) {
        long lo, hi;
        shard_range(B, k, n, &lo, &hi);
        if (hi <= lo) return static_cast<int>(FPE_OK);  // more devices than poses
        const size_t rec = static_cast<size_t>(lo) * n_cycles * 4;
        fpe_plan_out o;
        std::memset(&o, 0, sizeof(o));
        if (out->nominal) o.nominal = out->nominal + rec;
        if (out->centroid) o.centroid = out->centroid + rec;
        if (out->default_next) o.default_next = out->default_next + rec * 3;
        if (out->cycle_ok) o.cycle_ok = out->cycle_ok + static_cast<size_t>(lo) * n_cycles;
        if (out->stance) o.stance = out->stance + static_cast<size_t>(lo) * 12;
        if (out->selected) o.selected = out->selected + rec;
        if (out->pose_status) o.pose_status = out->pose_status + lo;
        return fpe_plan(h->engines[k], params, poses + lo, static_cast<int32_t>(hi - lo), n_cycles, &o);
    });
}

}  // extern "C"
