// fpe_engine.cpp — the C ABI of include/fpe.h over the gfx950 kernels (fpe_kernels.hip).
// Owns device memory: immutable map snapshots (swapped atomically, so a plan keeps the snapshot it
// started with — the reference races on gridmap_, cpp:506 vs cpp:818) and the spiral rank table.
// There is NO CPU compute path: without a usable GPU every compute entry point fails.
#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "fpe_host.hpp"

namespace fpe {
// fpe_kernels.hip
hipError_t launch_plan_chained(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const fpe_pose* d_poses,
                               int B, int nCycles, const fpe_plan_out& d_out, hipStream_t stream);
hipError_t launch_search_legs(const DevMap& m, const PlanConsts& pc, const SpiralLut& lut, const fpe_leg_query* d_q,
                              int n, fpe_foothold* d_out, hipStream_t stream);
hipError_t launch_canonicalise(const float* d_src, float* d_dst, int rows, int cols, int si, int sj, int srcRowMajor, hipStream_t stream,
                               uint32_t* d_planeWords = nullptr, float thrDefault = 0.0f, float thrCandidate = 0.0f);
hipError_t set_max_lds(size_t planBytes, size_t searchBytes);
size_t plan_lds_bytes(const PlanConsts& pc);
size_t search_lds_bytes(const PlanConsts& pc);
// bit-window path (fpe_bits.hip part of fpe_kernels.hip)
size_t bitmap_words(int rows, int cols, int* strideW, int* nw);
hipError_t launch_build_bitmap(const float* d_trav, int rows, int cols, float thrDefault, float thrCandidate,
                               uint32_t* d_words, hipStream_t stream);
bool bits_supported(const PlanConsts& pc, const MapGeom& g);
void describe_plan_kernel(const PlanConsts& pc, const MapGeom& g, char* buf, size_t n);
// producer filters (fpe_filters.hpp part of fpe_kernels.hip)
bool filters_supported(const FilterConsts& fc, const MapGeom& g);
bool filters_trav_only_ok(const FilterConsts& fc, const MapGeom& g);
hipError_t launch_filters(const MapGeom& g, const FilterConsts& fc, const float* d_elev, const FilterLayers& L, bool travOnly, hipStream_t stream);
hipError_t launch_plan_bits(const DevMap& m, const BitMap& bm, const PlanConsts& pc, const SpiralLut& lut,
                            const fpe_pose* d_poses, int B, int nCycles, const fpe_plan_out& d_out, hipStream_t stream);
// opt track (fpe_opt.hpp part of fpe_kernels.hip)
hipError_t launch_opt_track(const DevMap& m, const PlanConsts& pc, const OptConsts& oc, const fpe_pose* d_poses, int B, int nCycles,
                            const uint8_t* d_cycleOk, const fpe_opt_out& d_out, hipStream_t stream, uint32_t* doneFlag = nullptr, uint32_t doneValue = 0);
}  // namespace fpe

namespace {

thread_local std::string g_err;
thread_local fpe_service_gate g_gate = {255, FPE_GATE_NONE, 0, 0, {0, 0, 0, 0}, 0.0, 0.0};  // verdict of this thread's last fpe_plan_service* call
thread_local const void* g_gateEngine = nullptr;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int fail_hip(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return FPE_E_HIP;
}
#define FPE_HIP(call)                                     \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) return fail_hip(e_, #call); \
    } while (0)

// Layer buffers of retired snapshots are recycled instead of freed: a traversability map arrives
// at 10-20 Hz and hipMalloc/hipFree (the latter a device-wide sync) would dominate the upload.
// Lifetime without per-launch GPU work: a snapshot retired while device-API (asynchronous) launches may still
// read it hands its buffers over marked "dirty"; whoever takes a dirty buffer — an upload or a bit-plane build,
// 10-20 Hz paths — synchronises the device once before writing it.  Plans never pay for this: an event recorded
// after every launch would put a 3-4 us bubble between back-to-back plan kernels (measured).
struct BufferPool {
    struct Entry {
        size_t n;      // capacity in 4-byte units
        float* p;
        bool dirty;    // retired while asynchronous (device-API) work may still have been reading it
    };
    std::mutex mu;
    std::vector<Entry> free;
    // Best fit: the smallest pooled buffer that holds n units and is not more than twice as large (a map stream whose
    // size changes from message to message keeps recycling instead of falling back to hipMalloc / hipFree, which
    // synchronise the device).  *cap receives the buffer's real capacity: the caller hands THAT back to give().
    // cleanOnly: skip buffers that need a device synchronisation before reuse (the plan path never pays for one).
    float* take(size_t n, bool* dirty, size_t* cap, bool cleanOnly = false) {
        std::lock_guard<std::mutex> lk(mu);
        size_t best = free.size();
        for (size_t k = 0; k < free.size(); ++k) {
            if (free[k].n < n || free[k].n > 2 * n + 1024) continue;
            if (cleanOnly && free[k].dirty) continue;
            if (best == free.size() || free[k].n < free[best].n) best = k;
        }
        if (best == free.size()) {
            *dirty = false;
            *cap = 0;
            return nullptr;
        }
        float* p = free[best].p;
        *dirty = free[best].dirty;
        *cap = free[best].n;
        free.erase(free.begin() + static_cast<long>(best));
        return p;
    }
    void give(size_t n, float* p, bool dirty = false) {
        float* drop = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (free.size() >= 12) {  // two snapshots' worth (layers + bit planes) plus the upload staging layer
                drop = free.front().p;
                free.erase(free.begin());
            }
            free.push_back(Entry{n, p, dirty});
        }
        if (drop) (void)hipFree(drop);  // hipFree synchronises the device: safe for a dirty buffer too
    }
    ~BufferPool() {
        for (auto& e : free) (void)hipFree(e.p);
    }
};

// Per-call resources of the host-buffer entry points (stream, one device arena, one pinned host
// arena), recycled across calls: a plan_global_footholds call must not pay for hipMalloc /
// hipStreamCreate (3.6 ms per call before this pool, ~0.1 ms after).  Concurrent calls each take
// their own context.
struct CallCtx {
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;  // the opt track's chain of a small plan + opt call runs beside the plan kernel (plan_host)
    unsigned char* dev = nullptr;
    size_t devCap = 0;
    unsigned char* pinned = nullptr;
    size_t pinCap = 0;
    std::vector<hipEvent_t> events;  // one per staged result chunk in flight (created on demand, kept)
    // work may be queued on `stream` that reads / writes the arenas (and the caller's pinned arrays): set before the first
    // enqueue of a call, cleared once the call has synchronised.  A call that leaves early (a failed HIP call after its
    // kernels were queued) hands the context back with the flag set, and the lease waits for the stream before the next
    // call's memcpy can touch the arena the kernels are still reading.
    bool inFlight = false;
    uint32_t doneSeq = 0;  // sequence number of the completion word of polled one-pose calls (plan_host)
    ~CallCtx() {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        if (dev) (void)hipFree(dev);
        if (pinned) (void)hipHostFree(pinned);
        if (stream) (void)hipStreamDestroy(stream);
        if (side) (void)hipStreamDestroy(side);
    }
    hipError_t side_stream(hipStream_t* out) {
        if (!side) {
            // highest priority: the chain is the call's critical path — and a priority stream gets a hardware queue of its own class,
            // where a second default-priority stream may land on the SAME queue as `stream` (the runtime spreads a process's streams
            // over a few hardware queues) and run behind the plan kernel: measured in bench.py's process, many streams alive, as
            // 150 us per service call against 138 with the kernels one after the other and 105 in a process of its own
            int lo = 0, hi = 0;
            hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi);
            if (e != hipSuccess) {  // (no priorities on this runtime: a plain stream — correct, possibly behind the plan kernel)
                (void)hipGetLastError();
                side = nullptr;
                e = hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
                if (e != hipSuccess) return e;
            }
        }
        *out = side;
        return hipSuccess;
    }
    hipError_t event(size_t k, hipEvent_t* out) {
        while (events.size() <= k) {
            hipEvent_t e = nullptr;
            hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if (rc != hipSuccess) return rc;
            events.push_back(e);
        }
        *out = events[k];
        return hipSuccess;
    }
    hipError_t reserve(size_t bytes) {
        if (!stream) {
            hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
            if (e != hipSuccess) return e;
        }
        if (bytes > devCap) {
            if (dev) (void)hipFree(dev);
            dev = nullptr;
            devCap = 0;
            const size_t cap = std::max<size_t>(bytes + bytes / 4, 1 << 16);
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&dev), cap);
            if (e != hipSuccess) return e;
            devCap = cap;
        }
        if (bytes > pinCap) {
            if (pinned) (void)hipHostFree(pinned);
            pinned = nullptr;
            pinCap = 0;
            const size_t cap = std::max<size_t>(bytes + bytes / 4, 1 << 16);
            hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&pinned), cap, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            pinCap = cap;
        }
        return hipSuccess;
    }
};
struct CtxPool {
    std::mutex mu;
    std::vector<std::unique_ptr<CallCtx>> free;
    std::unique_ptr<CallCtx> take() {
        std::lock_guard<std::mutex> lk(mu);
        if (free.empty()) return std::unique_ptr<CallCtx>(new CallCtx());
        std::unique_ptr<CallCtx> c = std::move(free.back());
        free.pop_back();
        return c;
    }
    void give(std::unique_ptr<CallCtx> c) {
        std::lock_guard<std::mutex> lk(mu);
        if (free.size() < 8) free.push_back(std::move(c));
    }
};
struct CtxLease {  // returns the context to the pool on every exit path
    CtxPool& pool;
    std::unique_ptr<CallCtx> ctx;
    explicit CtxLease(CtxPool& p) : pool(p), ctx(p.take()) {}
    ~CtxLease() {
        if (ctx && ctx->inFlight && ctx->stream) {
            (void)hipStreamSynchronize(ctx->stream);  // best effort: an early exit after work was queued
            if (ctx->side) (void)hipStreamSynchronize(ctx->side);
            ctx->inFlight = false;
        }
        pool.give(std::move(ctx));
    }
};
inline size_t align256(size_t n) { return (n + 255) & ~static_cast<size_t>(255); }

// Copy-out of large results from the pinned arena into the caller's (pageable) arrays: a few persistent host threads,
// so that the copy of chunk k runs while chunk k+1 is still crossing PCIe (one thread moves ~25 GB/s, the link ~50).
class CopyPool {
  public:
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    struct Batch {
        std::mutex mu;
        std::condition_variable cv;
        int pending = 0;
    };
    void submit(Batch& b, void* dst, const void* src, size_t len) {
        {
            std::lock_guard<std::mutex> lk(b.mu);
            ++b.pending;
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (threads_.empty()) start();
            q_.push_back(Task{&b, dst, src, len});
        }
        cv_.notify_one();
    }
    static void wait(Batch& b) {
        std::unique_lock<std::mutex> lk(b.mu);
        b.cv.wait(lk, [&] { return b.pending == 0; });
    }
    static size_t width() {  // threads of the pool (start())
        const unsigned n = std::thread::hardware_concurrency();
        return n > 4 ? 3 : (n > 1 ? n - 1 : 1);
    }

  private:
    struct Task {
        Batch* b;
        void* dst;
        const void* src;
        size_t len;
    };
    void start() {  // under mu_
        const size_t n = width();
        for (size_t k = 0; k < n; ++k) threads_.emplace_back([this] { run(); });
    }
    void run() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                t = q_.front();
                q_.pop_front();
            }
            std::memcpy(t.dst, t.src, t.len);
            {
                std::lock_guard<std::mutex> lk(t.b->mu);
                if (--t.b->pending == 0) t.b->cv.notify_all();
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Task> q_;
    std::vector<std::thread> threads_;
    bool stop_ = false;
};

// Is `p` host memory the GPU can write by DMA (hipHostMalloc / hipHostRegister / fpe_host_alloc)?  Then results are
// copied from the device straight into it.
bool is_pinned_host(const void* p) {
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary malloc'ed pointer is reported as an error: not sticky
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// Bit planes of a snapshot for one (defaultFootholdThreshold, candidateFootholdThreshold) pair: built by the upload
// for the pairs the previous snapshot was planned with (the upload already owns the pool's synchronisation), lazily by
// the first plan for a pair never seen before.  Shared: a call keeps the set it launched with alive, so the snapshot
// can evict its least recently used set when a fifth pair arrives (no silent change of kernels).
struct MaskSet {
    float thrD = 0.0f, thrC = 0.0f;
    uint32_t* d_words = nullptr;
    size_t cap = 0;  // capacity in 4-byte units (what goes back to the pool)
    int strideW = 0, nw = 0;
    hipEvent_t ready = nullptr;
    std::atomic<bool> readyDone{false};
    std::atomic<bool> asyncUsed{false};
    std::atomic<unsigned long long> lastUse{0};
    std::shared_ptr<BufferPool> pool;
    ~MaskSet() {
        if (ready) {
            (void)hipEventSynchronize(ready);
            (void)hipEventDestroy(ready);
        }
        if (d_words) pool ? pool->give(cap, reinterpret_cast<float*>(d_words), asyncUsed.load(std::memory_order_acquire)) : (void)hipFree(d_words);
    }
};

struct MapSnapshot {
    fpe::MapGeom g;
    size_t n = 0;
    size_t capTrav = 0, capElev = 0;  // pooled capacities of the two layers
    float* d_trav = nullptr;
    float* d_elev = nullptr;
    std::shared_ptr<BufferPool> pool;
    // `ready` is recorded on the uploading stream after the copy / canonicalise work; a consumer stream waits on
    // it before its first kernel until the event has been seen complete once (then no GPU-side wait is queued
    // any more).  `asyncUsed`: some device-API launch read this snapshot without the host waiting for it.
    hipEvent_t ready = nullptr;
    std::atomic<bool> readyDone{false};
    std::atomic<bool> asyncUsed{false};
    std::mutex mu;
    std::vector<std::shared_ptr<MaskSet>> masks;
    unsigned long long useClock = 0;  // guarded by mu

    hipError_t wait_ready(hipStream_t s) {
        if (!ready || readyDone.load(std::memory_order_acquire)) return hipSuccess;
        if (hipEventQuery(ready) == hipSuccess) {
            readyDone.store(true, std::memory_order_release);
            return hipSuccess;
        }
        return hipStreamWaitEvent(s, ready, 0);
    }
    void note_async_use() { asyncUsed.store(true, std::memory_order_release); }
    ~MapSnapshot() {
        const bool dirty = asyncUsed.load(std::memory_order_acquire);
        if (ready) {
            (void)hipEventSynchronize(ready);  // the upload itself (normally long complete)
            (void)hipEventDestroy(ready);
        }
        for (auto& ms : masks)
            if (dirty) ms->asyncUsed.store(true, std::memory_order_release);
        masks.clear();
        if (d_trav) pool ? pool->give(capTrav, d_trav, dirty) : (void)hipFree(d_trav);
        if (d_elev) pool ? pool->give(capElev, d_elev, dirty) : (void)hipFree(d_elev);
    }
};

constexpr size_t kMaxLdsBytes = 160 * 1024;
constexpr size_t kLayerPadBytes = 128;  // >= kRowOverreadBytes of fpe_kernels.hip (96)
constexpr size_t kZeroCopyBytes = 64 * 1024;  // fpe_plan calls up to this size run on the pinned arena directly

}  // namespace

struct fpe_engine {
    int device = 0;
    std::mutex mu;
    std::shared_ptr<MapSnapshot> map;
    std::shared_ptr<BufferPool> pool = std::make_shared<BufferPool>();
    CtxPool ctxPool;
    CopyPool copyPool;
    int16_t* d_di = nullptr;
    int16_t* d_dj = nullptr;
    uint8_t* d_ring = nullptr;
    int32_t* d_ringStart = nullptr;
    uint32_t* d_packed = nullptr;
    uint32_t* d_fast16 = nullptr;
    int maxRing = 0;
    std::vector<int32_t> ringStart;   // host copy of the rank table's ring offsets
    float maxLegSearchRadius = 0.0f;  // fpe_set_max_leg_search_radius
    fpe::Tuning tuning;               // fpe_set_tuning; seeded from the environment once, in fpe_create
    // The traversability-only filter chain's step_height scratch: up to kFilterSlots buffers, each with the event recorded behind
    // the last chain that used it.  A chain takes, in this order: the buffer its own stream used last (chains of one stream run in
    // order: nothing to wait for — a 10-20 Hz producer on its own stream never waits and never synchronises), a buffer whose last
    // chain has completed (neither when it is more than four times too large: a single huge map must not pin memory for good), a fresh buffer while there are free slots, and only then the least recently used busy buffer, behind
    // a GPU-side hipStreamWaitEvent on its event.  No path synchronises the device or blocks the host (round 5 kept ONE buffer
    // keyed to one stream: two producers alternating streams paid a hipDeviceSynchronize per call).
    // A stream handed to fpe_traversability_device must outlive the chains queued on it (include/fpe.h).
    static constexpr int kFilterSlots = 4;
    struct FilterSlot {
        float* buf = nullptr;
        size_t cap = 0;
        hipEvent_t done = nullptr;
        hipStream_t last = nullptr;
        unsigned long long use = 0;
    };
    std::mutex filterMu;
    FilterSlot filterSlots[kFilterSlots];
    unsigned long long filterClock = 0;

    fpe::SpiralLut lut() const {
        return fpe::SpiralLut{d_di, d_dj, d_ring, d_ringStart, maxRing, ringStart.empty() ? 0 : ringStart[static_cast<size_t>(maxRing) + 1], d_packed, d_fast16};
    }
};

namespace {

// Device allocation in 4-byte units with the tail padding every pooled buffer carries (the row scan reads whole
// 16-byte groups past the end of a layer, fpe_kernels.hip::rows_issue), so any pooled buffer can serve any role.
// *cap: the capacity to hand back to the pool.  cleanOnly: never synchronise the device (the plan path).
hipError_t alloc_units(BufferPool& pool, size_t n, float** out, size_t* cap, bool cleanOnly = false) {
    bool dirty = false;
    *out = pool.take(n, &dirty, cap, cleanOnly);
    if (*out) return dirty ? hipDeviceSynchronize() : hipSuccess;  // see BufferPool: asynchronous readers may be in flight
    *cap = n;
    return hipMalloc(reinterpret_cast<void**>(out), n * sizeof(float) + kLayerPadBytes);
}

// What one launch needs besides the constants: the snapshot, and (bit-window path) its bit planes.
struct CallPlan {
    std::shared_ptr<MapSnapshot> snap;
    fpe::PlanConsts pc;
    size_t planLds = 0, searchLds = 0;
    bool useBits = false;
    fpe::BitMap bits{nullptr, 0, 0};
    std::shared_ptr<MaskSet> mask;  // the bit planes this call launches with (kept alive past an eviction)
};

// Bit planes of `snap` for a threshold pair: queued on `stream`, event recorded.  cleanOnly as in alloc_units.
// fill false: the buffer only — the caller's canonicalising kernel writes the planes while the layer passes through it
// (launch_canonicalise with d_planeWords) and then calls mask_ready.
int alloc_mask(MapSnapshot& snap, float thrD, float thrC, bool cleanOnly, std::shared_ptr<MaskSet>& out) {
    auto ms = std::make_shared<MaskSet>();
    ms->thrD = thrD;
    ms->thrC = thrC;
    ms->pool = snap.pool;
    const size_t need = fpe::bitmap_words(snap.g.rows, snap.g.cols, &ms->strideW, &ms->nw);
    float* buf = nullptr;
    FPE_HIP(alloc_units(*snap.pool, need, &buf, &ms->cap, cleanOnly));
    ms->d_words = reinterpret_cast<uint32_t*>(buf);
    out = std::move(ms);
    return FPE_OK;
}
int mask_ready(MaskSet& ms, hipStream_t stream) {
    hipError_t e = hipEventCreateWithFlags(&ms.ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ms.ready, stream);
    if (e != hipSuccess) {
        ms.asyncUsed.store(true);  // the build may be queued: hand the buffer back dirty
        return fail_hip(e, "bit-plane build");
    }
    return FPE_OK;
}
int build_mask(MapSnapshot& snap, float thrD, float thrC, hipStream_t stream, bool cleanOnly, std::shared_ptr<MaskSet>& out) {
    std::shared_ptr<MaskSet> ms;
    const int rc0 = alloc_mask(snap, thrD, thrC, cleanOnly, ms);
    if (rc0 != FPE_OK) return rc0;
    hipError_t e = fpe::launch_build_bitmap(snap.d_trav, snap.g.rows, snap.g.cols, thrD, thrC, ms->d_words, stream);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ms->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ms->ready, stream);
    if (e != hipSuccess) {
        ms->asyncUsed.store(true);  // the build may be queued: hand the buffer back dirty
        return fail_hip(e, "bit-plane build");
    }
    out = std::move(ms);
    return FPE_OK;
}

int prepare_call(fpe_engine* h, const fpe_params* params, float maxRadius, CallPlan& cp, hipStream_t stream, bool wantBits) {
    if (!h || !params) return fail(FPE_E_INVALID_ARG, "null handle or params");
    int rc = fpe::validate_params(*params);
    if (rc != FPE_OK) return fail(rc, "non-finite or negative parameter");
    fpe::Tuning tuning;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        cp.snap = h->map;
        maxRadius = std::max(maxRadius, std::max(params->searchRadius, h->maxLegSearchRadius));
        tuning = h->tuning;
    }
    if (!cp.snap) return fail(FPE_E_NO_MAP, "no map uploaded");
    MapSnapshot& snap = *cp.snap;
    fpe::PlanConsts& pc = cp.pc;
    fpe::derive_constants(*params, snap.g, maxRadius, tuning, pc);
    if (fpe::spiral_rings(maxRadius, snap.g.res) > h->maxRing)
        return fail(FPE_E_UNSUPPORTED, "search radius needs more spiral rings than the rank table holds");
    pc.defNRings = fpe::spiral_rings(params->searchRadius, snap.g.res);
    pc.defNCand = h->ringStart[static_cast<size_t>(std::min(pc.defNRings, h->maxRing)) + 1];
    cp.planLds = fpe::plan_lds_bytes(pc);
    cp.searchLds = fpe::search_lds_bytes(pc);
    if (cp.planLds > kMaxLdsBytes || cp.searchLds > kMaxLdsBytes)
        return fail(FPE_E_UNSUPPORTED, "search/foot radius too large for the 160 KiB LDS tile");
    FPE_HIP(hipSetDevice(h->device));
    FPE_HIP(snap.wait_ready(stream));
    cp.useBits = wantBits && fpe::bits_supported(pc, snap.g);
    if (cp.useBits) {
        std::lock_guard<std::mutex> lk(snap.mu);
        std::shared_ptr<MaskSet> found;
        for (auto& ms : snap.masks)
            if (std::memcmp(&ms->thrD, &pc.thrDefault, 4) == 0 && std::memcmp(&ms->thrC, &pc.thrCandidate, 4) == 0) found = ms;
        if (!found) {
            // a pair this snapshot has not been planned with (the upload pre-builds the pairs of the previous snapshot).
            // cleanOnly: a plan never synchronises the device for a recycled buffer — it allocates instead.
            rc = build_mask(snap, pc.thrDefault, pc.thrCandidate, stream, true, found);
            if (rc != FPE_OK) return rc;
            if (snap.masks.size() >= 4) {  // least recently used set out (calls in flight keep theirs alive)
                size_t lru = 0;
                for (size_t k = 1; k < snap.masks.size(); ++k)
                    if (snap.masks[k]->lastUse.load() < snap.masks[lru]->lastUse.load()) lru = k;
                if (snap.asyncUsed.load(std::memory_order_acquire)) snap.masks[lru]->asyncUsed.store(true);
                snap.masks.erase(snap.masks.begin() + static_cast<long>(lru));
            }
            snap.masks.push_back(found);
        }
        found->lastUse.store(++snap.useClock);
        cp.mask = found;
        cp.bits = fpe::BitMap{reinterpret_cast<const uint4*>(found->d_words), found->strideW, found->nw};
        if (!found->readyDone.load(std::memory_order_acquire)) {
            if (hipEventQuery(found->ready) == hipSuccess) found->readyDone.store(true, std::memory_order_release);
            else FPE_HIP(hipStreamWaitEvent(stream, found->ready, 0));
        }
    }
    return FPE_OK;
}

fpe::DevMap dev_map(const MapSnapshot& s) { return fpe::DevMap{s.g, s.d_trav, s.d_elev}; }

int check_desc(const fpe_map_desc* d) {
    if (!d) return fail(FPE_E_INVALID_ARG, "null map descriptor");
    if (d->rows <= 0 || d->cols <= 0 || d->rows > 32768 || d->cols > 32768)
        return fail(FPE_E_INVALID_ARG, "map size out of range");
    if (!(d->resolution > 0.0) || !std::isfinite(d->resolution)) return fail(FPE_E_INVALID_ARG, "bad resolution");
    if (!std::isfinite(d->position[0]) || !std::isfinite(d->position[1])) return fail(FPE_E_INVALID_ARG, "bad position");
    if (d->start_index[0] < 0 || d->start_index[0] >= d->rows || d->start_index[1] < 0 || d->start_index[1] >= d->cols)
        return fail(FPE_E_INVALID_ARG, "start index out of range");
    if (d->storage_order != 0 && d->storage_order != 1) return fail(FPE_E_INVALID_ARG, "bad storage order");
    return FPE_OK;
}

int upload_common(fpe_engine* h, const fpe_map_desc* desc, const float* trav, const float* elev, bool srcOnDevice,
                  hipStream_t stream) {
    if (!h || !trav || !elev) return fail(FPE_E_INVALID_ARG, "null argument");
    int rc = check_desc(desc);
    if (rc != FPE_OK) return rc;
    FPE_HIP(hipSetDevice(h->device));
    const size_t n = static_cast<size_t>(desc->rows) * desc->cols;
    auto snap = std::make_shared<MapSnapshot>();
    snap->g = fpe::make_geom(desc->rows, desc->cols, desc->resolution, desc->position[0], desc->position[1]);
    snap->n = n;
    snap->pool = h->pool;
    FPE_HIP(alloc_units(*h->pool, n, &snap->d_trav, &snap->capTrav));
    FPE_HIP(alloc_units(*h->pool, n, &snap->d_elev, &snap->capElev));
    const bool canonical = desc->storage_order == 1 && desc->start_index[0] == 0 && desc->start_index[1] == 0;
    const float* src[2] = {trav, elev};
    float* dst[2] = {snap->d_trav, snap->d_elev};
    // staging layer of a host upload in message layout: recycled like the snapshot layers (a 10-20 Hz map stream
    // must not hipMalloc/hipFree per message), and returned to the pool on every exit path
    struct Staging {
        std::shared_ptr<BufferPool> pool;
        size_t cap = 0;
        float* p = nullptr;
        ~Staging() {
            if (p) pool->give(cap, p);
        }
    } stagingGuard{h->pool};
    float*& staging = stagingGuard.p;
    if (!srcOnDevice && !canonical) {
        FPE_HIP(alloc_units(*h->pool, n, &staging, &stagingGuard.cap));
    }
    // Threshold pairs the CURRENT snapshot was planned with: their bit planes are built here, on the upload's stream — the
    // upload is the 10-20 Hz path that may take a recycled ("dirty") buffer behind a device synchronisation; the first plan on
    // the new map then finds its planes and pays nothing (include/fpe.h: "plans never pay for it").  The FIRST pair's planes
    // ride in the kernel that moves the traversability layer anyway (canonicalise_layer_kernel ballots every destination row it
    // holds; round 5) whenever the layer goes through a kernel — message layout, or a device source (then the kernel replaces
    // the device-to-device copy): no second pass over the layer.  Further pairs (rare) by build_bitmap_kernel.
    std::vector<std::pair<float, float>> pairs;
    {
        std::shared_ptr<MapSnapshot> cur;
        {
            std::lock_guard<std::mutex> lk(h->mu);
            cur = h->map;
        }
        if (cur) {
            std::lock_guard<std::mutex> lk(cur->mu);
            for (auto& ms : cur->masks) pairs.emplace_back(ms->thrD, ms->thrC);
        }
    }
    std::shared_ptr<MaskSet> folded;
    const bool travThroughKernel = !canonical || srcOnDevice;
    if (!pairs.empty() && travThroughKernel && desc->rows < (1 << 24) - 2 && desc->cols < (1 << 24) - 64) {
        rc = alloc_mask(*snap, pairs[0].first, pairs[0].second, false, folded);
        if (rc != FPE_OK) return rc;
        // the canonicalising kernel below writes this buffer asynchronously: should any later step of the upload fail, the buffer
        // goes back to the pool DIRTY (its next taker synchronises) — a dirty return costs a synchronisation, a clean one of a
        // buffer with work queued on it would be a race.  Cleared again once `ready` is recorded, below.
        folded->asyncUsed.store(true, std::memory_order_release);
    }
    for (int l = 0; l < 2; ++l) {
        uint32_t* planeWords = (l == 0 && folded) ? folded->d_words : nullptr;
        if (canonical && !planeWords) {
            FPE_HIP(hipMemcpyAsync(dst[l], src[l], n * sizeof(float),
                                   srcOnDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
        } else {
            const float* dsrc = src[l];
            if (!srcOnDevice) {
                FPE_HIP(hipMemcpyAsync(staging, src[l], n * sizeof(float), hipMemcpyHostToDevice, stream));
                dsrc = staging;
            }
            FPE_HIP(fpe::launch_canonicalise(dsrc, dst[l], desc->rows, desc->cols, desc->start_index[0], desc->start_index[1],
                                             desc->storage_order, stream, planeWords, planeWords ? folded->thrD : 0.0f,
                                             planeWords ? folded->thrC : 0.0f));
        }
        if (!srcOnDevice) FPE_HIP(hipStreamSynchronize(stream));  // host buffers may be freed on return
    }
    // consumers on other streams wait for this event before their first kernel (device-source uploads are
    // asynchronous; host-source uploads have been synchronised above, the event is then already complete)
    FPE_HIP(hipEventCreateWithFlags(&snap->ready, hipEventDisableTiming));
    FPE_HIP(hipEventRecord(snap->ready, stream));
    if (folded) {
        rc = mask_ready(*folded, stream);
        if (rc != FPE_OK) return rc;
        // from here on ~MaskSet waits for `ready` (recorded behind the kernel that wrote the planes) before it gives the buffer back
        folded->asyncUsed.store(false, std::memory_order_release);
        folded->lastUse.store(++snap->useClock);
        snap->masks.push_back(folded);
    }
    for (size_t k = folded ? 1 : 0; k < pairs.size(); ++k) {
        std::shared_ptr<MaskSet> ms;
        rc = build_mask(*snap, pairs[k].first, pairs[k].second, stream, false, ms);
        if (rc != FPE_OK) return rc;
        ms->lastUse.store(++snap->useClock);
        snap->masks.push_back(std::move(ms));
    }
    // A host-buffer upload is synchronous already (the caller's arrays may be freed on return): it also waits for the planes it
    // has just queued, so the snapshot it installs is COMPLETE and the first service call on the new map waits for nothing
    // (bench.py service_latency_us: first_call_after_a_map was 16 us above steady_map — the plane build and its event on the
    // call's critical path; the map stream is the 10-20 Hz path that can afford it).
    if (!srcOnDevice && !snap->masks.empty()) {
        FPE_HIP(hipStreamSynchronize(stream));
        for (auto& ms : snap->masks) ms->readyDone.store(true, std::memory_order_release);
    }
    // (a host upload has synchronised behind each layer's copy: the snapshot is known complete — its first consumer need not ask the event)
    if (!srcOnDevice) snap->readyDone.store(true, std::memory_order_release);
    std::shared_ptr<MapSnapshot> old;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        old = std::move(h->map);
        h->map = snap;  // readers holding the old snapshot keep it alive until they finish
    }
    old.reset();  // outside the engine lock: ~MapSnapshot may wait for the old snapshot's last asynchronous readers
    return FPE_OK;
}

}  // namespace

extern "C" {

int fpe_abi_version(void) { return FPE_ABI_VERSION; }
const char* fpe_version(void) { return "fpe 0.5.0 (gfx950, wave64; bit-window plan kernels on tiled planes: 8 lanes per leg and two poses per wavefront, one wavefront per pose for large windows; opt track with a build-defined optimiser)"; }

const char* fpe_last_error(fpe_handle) { return g_err.c_str(); }

int fpe_create(int device_id, fpe_handle* out) {
    if (!out) return fail(FPE_E_INVALID_ARG, "null out handle");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(FPE_E_NO_DEVICE, "no HIP device visible");
    if (device_id < 0 || device_id >= count) return fail(FPE_E_INVALID_ARG, "device id out of range");
    FPE_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    FPE_HIP(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(FPE_E_NO_DEVICE, std::string("device is not gfx950: ") + prop.gcnArchName);
    fpe_engine* h = new (std::nothrow) fpe_engine();
    if (!h) return fail(FPE_E_NOMEM, "out of host memory");
    h->device = device_id;
    // tuning defaults: the environment is read HERE, once per engine — never in the per-call path
    if (const char* v = std::getenv("FPE_PLAN_GROUP")) h->tuning.planGroup = std::atoi(v);
    if (std::getenv("FPE_LITERAL_DISCS")) h->tuning.literalDiscs = 1;
    if (std::getenv("FPE_NO_MID_VARIANT")) h->tuning.noMidVariant = 1;
    if (std::getenv("FPE_NO_BITS")) h->tuning.noBits = 1;
    fpe::SpiralTable t;
    fpe::build_spiral_table(fpe::kMaxRings, t);
    h->maxRing = t.maxRing;
    h->ringStart = t.ringStart;
    const size_t n = t.di.size();
    auto cleanup = [&]() { fpe_destroy(h); };
#define FPE_HIP_C(call)                       \
    do {                                      \
        hipError_t e_ = (call);               \
        if (e_ != hipSuccess) {               \
            cleanup();                        \
            return fail_hip(e_, #call);       \
        }                                     \
    } while (0)
    FPE_HIP_C(hipMalloc(reinterpret_cast<void**>(&h->d_di), n * sizeof(int16_t)));
    FPE_HIP_C(hipMalloc(reinterpret_cast<void**>(&h->d_dj), n * sizeof(int16_t)));
    FPE_HIP_C(hipMalloc(reinterpret_cast<void**>(&h->d_ring), n * sizeof(uint8_t)));
    FPE_HIP_C(hipMalloc(reinterpret_cast<void**>(&h->d_ringStart), t.ringStart.size() * sizeof(int32_t)));
    FPE_HIP_C(hipMemcpy(h->d_di, t.di.data(), n * sizeof(int16_t), hipMemcpyHostToDevice));
    FPE_HIP_C(hipMemcpy(h->d_dj, t.dj.data(), n * sizeof(int16_t), hipMemcpyHostToDevice));
    FPE_HIP_C(hipMemcpy(h->d_ring, t.ring.data(), n * sizeof(uint8_t), hipMemcpyHostToDevice));
    FPE_HIP_C(hipMemcpy(h->d_ringStart, t.ringStart.data(), t.ringStart.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    {
        std::vector<uint32_t> packed(((n + 31) / 32 + 1) * 32, 255u << 16);  // SpiralLut::packed
        for (size_t k = 0; k < n; ++k)
            packed[k] = (static_cast<uint32_t>(t.di[k]) & 0xFFu) | ((static_cast<uint32_t>(t.dj[k]) & 0xFFu) << 8) | (static_cast<uint32_t>(t.ring[k]) << 16);
        FPE_HIP_C(hipMalloc(reinterpret_cast<void**>(&h->d_packed), packed.size() * sizeof(uint32_t)));
        FPE_HIP_C(hipMemcpy(h->d_packed, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        // SpiralLut::fast16: the first sixteen ranks by row (they all lie within two cells of the centre: rings 0-2)
        uint32_t fast[fpe::kFast16Words] = {0};
        for (int r = 0; r < 5; ++r) fast[r] = 0x1FFFFFFu;  // five 5-bit fields, 31 = none
        unsigned long long tdi = 0ull, tdj = 0ull;
        bool fastOk = n >= 16;
        for (size_t q = 0; q < 16 && fastOk; ++q) {
            const int di = t.di[q], dj = t.dj[q];
            if (di < -2 || di > 2 || dj < -2 || dj > 2) {
                fastOk = false;
                break;
            }
            uint32_t& w = fast[di + 2];
            w = (w & ~(31u << (5 * (dj + 2)))) | (static_cast<uint32_t>(q) << (5 * (dj + 2)));
            tdi |= static_cast<unsigned long long>(di + 2) << (4 * q);
            tdj |= static_cast<unsigned long long>(dj + 2) << (4 * q);
        }
        if (!fastOk) {
            cleanup();
            return fail(FPE_E_HIP, "spiral table: the first sixteen ranks do not lie within two cells of the centre");
        }
        fast[6] = static_cast<uint32_t>(tdi);
        fast[7] = static_cast<uint32_t>(tdi >> 32);
        fast[8] = static_cast<uint32_t>(tdj);
        fast[9] = static_cast<uint32_t>(tdj >> 32);
        FPE_HIP_C(hipMalloc(reinterpret_cast<void**>(&h->d_fast16), sizeof(fast)));
        FPE_HIP_C(hipMemcpy(h->d_fast16, fast, sizeof(fast), hipMemcpyHostToDevice));
    }
    // every kernel may use the whole 160 KiB of LDS: set once per process and device, never lowered again (a
    // second engine on the same device sets the same value)
    FPE_HIP_C(fpe::set_max_lds(kMaxLdsBytes, kMaxLdsBytes));
#undef FPE_HIP_C
    *out = h;
    return FPE_OK;
}

int fpe_destroy(fpe_handle h) {
    if (!h) return FPE_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->d_di) (void)hipFree(h->d_di);
    if (h->d_dj) (void)hipFree(h->d_dj);
    if (h->d_ring) (void)hipFree(h->d_ring);
    if (h->d_ringStart) (void)hipFree(h->d_ringStart);
    if (h->d_packed) (void)hipFree(h->d_packed);
    if (h->d_fast16) (void)hipFree(h->d_fast16);
    for (auto& fs : h->filterSlots) {
        if (fs.buf) (void)hipFree(fs.buf);
        if (fs.done) (void)hipEventDestroy(fs.done);
    }
    h->map.reset();
    delete h;
    return FPE_OK;
}

int fpe_set_tuning(fpe_handle h, const char* key, int32_t value) {
    if (!h || !key) return fail(FPE_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(h->mu);
    const std::string k(key);
    if (k == "plan_group") h->tuning.planGroup = value;
    else if (k == "literal_discs") h->tuning.literalDiscs = value ? 1 : 0;
    else if (k == "no_mid_variant") h->tuning.noMidVariant = value ? 1 : 0;
    else if (k == "no_bits") h->tuning.noBits = value ? 1 : 0;
    else if (k == "service_opt_gate") {
        if (value < 0 || value > 2) return fail(FPE_E_INVALID_ARG, "service_opt_gate is 0 (exact gates only), 1 (advisory) or 2 (enforce)");
        h->tuning.serviceOptGate = value;
    } else if (k == "service_overlap") h->tuning.serviceOverlap = value ? 1 : 0;
    else if (k == "service_poll") h->tuning.servicePoll = value ? 1 : 0;
    else if (k == "service_cycle0_gate_only") h->tuning.serviceOptGate = value ? 0 : 2;  // (older name)
    else return fail(FPE_E_INVALID_ARG, "unknown tuning key: " + k);
    return FPE_OK;
}

int fpe_describe_plan(fpe_handle h, const fpe_params* params, char* buf, int32_t n) {
    if (!h || !params || !buf || n <= 0) return fail(FPE_E_INVALID_ARG, "null argument");
    std::shared_ptr<MapSnapshot> snap;
    fpe::Tuning tuning;
    float maxRadius;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        snap = h->map;
        tuning = h->tuning;
        maxRadius = std::max(params->searchRadius, h->maxLegSearchRadius);
    }
    if (!snap) return fail(FPE_E_NO_MAP, "no map uploaded");
    fpe::PlanConsts pc;
    fpe::derive_constants(*params, snap->g, maxRadius, tuning, pc);
    fpe::describe_plan_kernel(pc, snap->g, buf, static_cast<size_t>(n));
    return FPE_OK;
}

int fpe_set_max_leg_search_radius(fpe_handle h, float radius) {
    if (!h || !(radius >= 0.0f) || !std::isfinite(radius)) return fail(FPE_E_INVALID_ARG, "bad radius");
    std::lock_guard<std::mutex> lk(h->mu);
    h->maxLegSearchRadius = radius;
    return FPE_OK;
}

int fpe_upload_map(fpe_handle h, const fpe_map_desc* desc, const float* traversability, const float* elevation) {
    return upload_common(h, desc, traversability, elevation, false, nullptr);
}

int fpe_upload_map_device(fpe_handle h, const fpe_map_desc* desc, const float* d_traversability,
                          const float* d_elevation, void* stream) {
    return upload_common(h, desc, d_traversability, d_elevation, true, static_cast<hipStream_t>(stream));
}

int fpe_filter_params_defaults(fpe_filter_params* out) {
    if (!out) return fail(FPE_E_INVALID_ARG, "null out");
    std::memset(out, 0, sizeof(*out));
    out->normal_radius = 0.05;
    out->slope_critical = 1.0;
    out->step_critical = 0.12;
    out->step_first_radius = 0.08;
    out->step_second_radius = 0.08;
    out->step_critical_cells = 4;
    out->roughness_critical = 0.05;
    out->roughness_radius = 0.05;
    return FPE_OK;
}

namespace {
int filters_common(fpe_engine* h, const fpe_map_desc* desc, const fpe_filter_params* fp, const float* elev, bool onDevice,
                   float* trav, float* layers, hipStream_t stream) {
    if (!h || !fp || !elev || !trav) return fail(FPE_E_INVALID_ARG, "null argument");
    int rc = check_desc(desc);
    if (rc != FPE_OK) return rc;
    const double radii[4] = {fp->normal_radius, fp->step_first_radius, fp->step_second_radius, fp->roughness_radius};
    for (double r : radii)
        if (!(r > 0.0) || !std::isfinite(r)) return fail(FPE_E_INVALID_ARG, "filter radius must be positive and finite");
    if (!(fp->slope_critical > 0.0) || !(fp->step_critical > 0.0) || !(fp->roughness_critical > 0.0) || fp->step_critical_cells <= 0)
        return fail(FPE_E_INVALID_ARG, "filter critical values must be positive");
    const fpe::MapGeom g = fpe::make_geom(desc->rows, desc->cols, desc->resolution, desc->position[0], desc->position[1]);
    const fpe::FilterConsts fc{fp->normal_radius, fp->slope_critical, fp->step_critical, fp->step_first_radius, fp->step_second_radius,
                               fp->step_critical_cells, fp->roughness_critical, fp->roughness_radius};
    if (!fpe::filters_supported(fc, g)) return fail(FPE_E_UNSUPPORTED, "filter radius spans more cells than the stencil tables hold");
    FPE_HIP(hipSetDevice(h->device));
    const size_t n = static_cast<size_t>(desc->rows) * desc->cols;
    const bool canonical = desc->storage_order == 1 && desc->start_index[0] == 0 && desc->start_index[1] == 0;
    // pooled scratch: the canonical elevation (when the source is a host buffer or in message layout) and the eight layers
    struct Scratch {
        std::shared_ptr<BufferPool> pool;
        std::vector<std::pair<size_t, float*>> bufs;
        ~Scratch() {
            for (auto& b : bufs) pool->give(b.first, b.second, true);
        }
    } scratch{h->pool, {}};
    auto take = [&](size_t units, float** out) -> hipError_t {
        size_t cap = 0;
        hipError_t e = alloc_units(*h->pool, units, out, &cap);
        if (e == hipSuccess) scratch.bufs.emplace_back(cap, *out);
        return e;
    };
    const float* d_elev = elev;
    if (!onDevice || !canonical) {
        float* canon = nullptr;
        FPE_HIP(take(n, &canon));
        const float* dsrc = elev;
        if (!onDevice) {
            float* staging = canon;
            if (!canonical) FPE_HIP(take(n, &staging));
            FPE_HIP(hipMemcpyAsync(staging, elev, n * sizeof(float), hipMemcpyHostToDevice, stream));
            dsrc = staging;
        }
        if (!canonical)
            FPE_HIP(fpe::launch_canonicalise(dsrc, canon, desc->rows, desc->cols, desc->start_index[0], desc->start_index[1],
                                             desc->storage_order, stream));
        d_elev = canon;
    }
    float* d_layers = onDevice ? layers : nullptr;
    // Nobody asked for the intermediate layers (device caller without a layer buffer, host caller without `layers`): the
    // two-launch chain that stores step_height and traversability only, traversability straight into the caller's buffer
    // when it is a device buffer.
    // (not when the caller's traversability buffer IS — or overlaps — the elevation buffer: the fused launch writes traversability
    // tile by tile while its neighbours still read their halos; the layer-buffer path below writes scratch and copies afterwards)
    const bool inPlace = onDevice && trav < elev + n && elev < trav + n;
    const bool travOnly = !layers && !inPlace && fpe::filters_trav_only_ok(fc, g);
    if (travOnly) {
        float* d_trav = trav;
        if (!onDevice) FPE_HIP(take(n, &d_trav));
        {
            // the lock is held until the chain is QUEUED and its event recorded (see fpe_engine::filterSlots)
            std::lock_guard<std::mutex> lk(h->filterMu);
            fpe_engine::FilterSlot* slot = nullptr;
            bool wait = false;
            for (auto& fs : h->filterSlots)  // (1) this stream's own buffer
                if (fs.buf && fs.cap >= n && fs.cap <= 4 * n + 4096 && fs.last == stream && (!slot || fs.use > slot->use)) slot = &fs;
            // (the wait on the slot's event is queued in this case too: on the same in-order queue it is free, and a caller who destroyed
            // the stream with work pending and got the same handle value back for a new one is ordered behind the old chain — ADVICE r5)
            if (slot) wait = true;
            if (!slot)
                for (auto& fs : h->filterSlots)  // (2) an idle buffer
                    if (fs.buf && fs.cap >= n && fs.cap <= 4 * n + 4096 && hipEventQuery(fs.done) == hipSuccess) {
                        slot = &fs;
                        break;
                    }
            (void)hipGetLastError();  // hipErrorNotReady of the queries above is not an error
            if (!slot) {
                for (auto& fs : h->filterSlots)  // (3) a free slot, else (4) the least recently used one
                    if (!fs.buf) {
                        slot = &fs;
                        break;
                    }
                if (!slot) {
                    slot = &h->filterSlots[0];
                    for (auto& fs : h->filterSlots)
                        if (fs.use < slot->use) slot = &fs;
                    wait = true;
                }
                if (slot->buf && slot->cap < n) {  // too small (also reached by (4)): back to the pool, dirty — its last chain may still run
                    h->pool->give(slot->cap, slot->buf, true);
                    slot->buf = nullptr;
                    wait = false;
                }
                if (!slot->buf) FPE_HIP(alloc_units(*h->pool, n, &slot->buf, &slot->cap, true));  // cleanOnly: never a device synchronisation
                if (!slot->done) FPE_HIP(hipEventCreateWithFlags(&slot->done, hipEventDisableTiming));
            }
            if (wait) FPE_HIP(hipStreamWaitEvent(stream, slot->done, 0));
            const fpe::FilterLayers L{nullptr, nullptr, nullptr, nullptr, slot->buf, nullptr, nullptr, d_trav};
            FPE_HIP(fpe::launch_filters(g, fc, d_elev, L, true, stream));
            FPE_HIP(hipEventRecord(slot->done, stream));
            slot->last = stream;
            slot->use = ++h->filterClock;
        }
        if (!onDevice) {
            FPE_HIP(hipMemcpyAsync(trav, d_trav, n * sizeof(float), hipMemcpyDeviceToHost, stream));
            FPE_HIP(hipStreamSynchronize(stream));
        }
        return FPE_OK;
    }
    if (!d_layers) FPE_HIP(take(8 * n, &d_layers));
    const fpe::FilterLayers L{d_layers, d_layers + n, d_layers + 2 * n, d_layers + 3 * n, d_layers + 4 * n, d_layers + 5 * n,
                              d_layers + 6 * n, d_layers + 7 * n};
    FPE_HIP(fpe::launch_filters(g, fc, d_elev, L, false, stream));
    if (onDevice) {
        if (trav != L.trav) FPE_HIP(hipMemcpyAsync(trav, L.trav, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
        // pooled scratch is handed back marked dirty: the next taker synchronises the device before reuse (BufferPool)
    } else {
        FPE_HIP(hipMemcpyAsync(trav, L.trav, n * sizeof(float), hipMemcpyDeviceToHost, stream));
        if (layers) FPE_HIP(hipMemcpyAsync(layers, d_layers, 8 * n * sizeof(float), hipMemcpyDeviceToHost, stream));
        FPE_HIP(hipStreamSynchronize(stream));
    }
    return FPE_OK;
}
}  // namespace

int fpe_traversability(fpe_handle h, const fpe_map_desc* desc, const fpe_filter_params* fp, const float* elevation,
                       float* traversability, float* layers) {
    return filters_common(h, desc, fp, elevation, false, traversability, layers, nullptr);
}

int fpe_traversability_device(fpe_handle h, const fpe_map_desc* desc, const fpe_filter_params* fp, const float* d_elevation,
                              float* d_traversability, float* d_layers, void* stream) {
    return filters_common(h, desc, fp, d_elevation, true, d_traversability, d_layers, static_cast<hipStream_t>(stream));
}

int fpe_map_info(fpe_handle h, fpe_map_desc* out) {
    if (!h || !out) return fail(FPE_E_INVALID_ARG, "null argument");
    std::shared_ptr<MapSnapshot> snap;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        snap = h->map;
    }
    if (!snap) return fail(FPE_E_NO_MAP, "no map uploaded");
    out->rows = snap->g.rows;
    out->cols = snap->g.cols;
    out->resolution = snap->g.res;
    out->position[0] = snap->g.posX;
    out->position[1] = snap->g.posY;
    out->start_index[0] = out->start_index[1] = 0;
    out->storage_order = 1;
    return FPE_OK;
}

// One chained-plan launch on `stream`: the bit-window kernels when the snapshot's bit planes apply, else the direct ones.
static int launch_plan(fpe_engine* h, const CallPlan& cp, const fpe_pose* d_poses, int32_t B, int32_t n_cycles,
                       const fpe_plan_out& d_out, hipStream_t stream) {
    if (d_out.selected_packed && (cp.snap->g.rows > FPE_PACKED_MAX_CELLS || cp.snap->g.cols > FPE_PACKED_MAX_CELLS))
        return fail(FPE_E_UNSUPPORTED, "selected_packed holds biased 14-bit grid indices: the map has more than 15871 rows or columns");
    if (cp.useBits)
        FPE_HIP(fpe::launch_plan_bits(dev_map(*cp.snap), cp.bits, cp.pc, h->lut(), d_poses, B, n_cycles, d_out, stream));
    else
        FPE_HIP(fpe::launch_plan_chained(dev_map(*cp.snap), cp.pc, h->lut(), d_poses, B, n_cycles, d_out, stream));
    return FPE_OK;
}

int fpe_plan_device(fpe_handle h, const fpe_params* params, const fpe_pose* d_poses, int32_t B, int32_t n_cycles,
                    const fpe_plan_out* d_out, void* stream) {
    if (!d_poses || !d_out) return fail(FPE_E_INVALID_ARG, "null argument");
    if (B <= 0 || n_cycles <= 0 || n_cycles > 255) return fail(FPE_E_INVALID_ARG, "B and n_cycles must be in [1, ..] / [1, 255]");
    hipStream_t st = static_cast<hipStream_t>(stream);
    CallPlan cp;
    int rc = prepare_call(h, params, 0.0f, cp, st, true);
    if (rc != FPE_OK) return rc;
    rc = launch_plan(h, cp, d_poses, B, n_cycles, *d_out, st);
    if (rc != FPE_OK) return rc;
    cp.snap->note_async_use();  // asynchronous launch: the snapshot's buffers are recycled only behind a device sync
    if (cp.mask) cp.mask->asyncUsed.store(true, std::memory_order_release);
    return FPE_OK;
}

namespace {
// Constants of the opt track for this map, checked (fpe_host.cpp::derive_opt_constants).
int prepare_opt(const fpe_params* params, const fpe_opt_params* opt, const CallPlan& cp, float maxRadius, fpe::OptConsts& oc) {
    fpe_opt_params yaml;
    if (!opt) {
        fpe_opt_params_yaml(&yaml);
        opt = &yaml;
    }
    const int rc = fpe::derive_opt_constants(*params, *opt, cp.snap->g, cp.pc, oc);
    if (rc != FPE_OK) return fail(rc, "opt parameters: weights / scales / tolerance must be finite");
    // checkFootholdUseCentroidMethod on the gait-cycle submap keeps its blocked-row mask in 128 bits
    const double rows = 2.0 * static_cast<double>(std::max(maxRadius, params->searchRadius)) / cp.snap->g.res + 3.0;
    if (rows > 128.0) return fail(FPE_E_UNSUPPORTED, "opt track: the foot search rectangle spans more than 128 rows");
    return FPE_OK;
}

// The host-buffer form of the plan and / or the opt track: one device arena, the launches back to back on one
// stream, the results copied out.  `out` NULL: no plan products are returned (the plan still runs when the opt track
// needs its cycle flags and the caller gave none).
#ifdef FPE_HOST_TIMING  // measurement builds only: where a small host-buffer call spends its time on the HOST (profiles/probe_host_timing.py)
static double g_hostT[16];
static long g_hostN = 0;
struct HostTimer {
    std::chrono::steady_clock::time_point t[16];
    int last = -1;
    void mark(int k) { t[k] = std::chrono::steady_clock::now(); last = k; }
    void commit() {
        for (int k = 1; k <= last; ++k) g_hostT[k] += std::chrono::duration<double, std::micro>(t[k] - t[k - 1]).count();
        ++g_hostN;
    }
};
#define FPE_HT_MARK(k) ht.mark(k)
#else
#define FPE_HT_MARK(k) do { } while (0)
#endif
struct GateGeom {  // what the service call's host-side (lateral) gate needs from the call's snapshot and constants
    fpe::MapGeom g;
    double isosLen, isosWid, drift;
};
int plan_host(fpe_engine* h, const fpe_params* params, const fpe_opt_params* opt, const fpe_pose* poses, int32_t B, int32_t n_cycles,
              const fpe_plan_out* out, const uint8_t* cycleOkIn, const fpe_opt_out* oout, GateGeom* gateGeom = nullptr,
              bool* optDropped = nullptr) {
#ifdef FPE_HOST_TIMING
    HostTimer ht;
#endif
    FPE_HT_MARK(0);
    if (!poses || (!out && !oout)) return fail(FPE_E_INVALID_ARG, "null argument");
    if (B <= 0 || n_cycles <= 0 || n_cycles > 255) return fail(FPE_E_INVALID_ARG, "B and n_cycles must be in [1, ..] / [1, 255]");
    float maxRadius = 0.0f;
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(poses[b].position[k]) || std::fabs(poses[b].position[k]) > 1e6)
                return fail(FPE_E_INVALID_ARG, "non-finite pose");
        if (poses[b].gait != 0 && poses[b].gait != 1) return fail(FPE_E_INVALID_ARG, "unknown gait");
        for (int l = 0; l < 4; ++l) {
            if (std::isfinite(poses[b].leg_search_radius[l])) maxRadius = std::max(maxRadius, poses[b].leg_search_radius[l]);
            if (poses[b].leg_polygon_kind[l] != 0 && poses[b].leg_polygon_kind[l] != 1)
                return fail(FPE_E_INVALID_ARG, "unknown polygon kind");
        }
    }
    fpe_plan_out none;
    std::memset(&none, 0, sizeof(none));
    const fpe_plan_out& po = out ? *out : none;
    const bool runPlan = out != nullptr || (oout && !cycleOkIn);
    const bool needOkDev = po.cycle_ok != nullptr || oout != nullptr;  // device copy of the cycle flags
    // one device arena + one pinned arena:
    // [poses | nominal | centroid | default | cycle_ok | stance | selected | status | packed | opt footholds | opt cycles | gate]
    const size_t nRec = static_cast<size_t>(B) * n_cycles * 4;
    const size_t nCyc = static_cast<size_t>(B) * n_cycles;
    const size_t szPose = align256(static_cast<size_t>(B) * sizeof(fpe_pose));
    const size_t szNom = po.nominal ? align256(nRec * sizeof(fpe_foothold)) : 0;
    const size_t szCen = po.centroid ? align256(nRec * sizeof(fpe_centroid_foothold)) : 0;
    const size_t szDef = po.default_next ? align256(nRec * 3 * sizeof(double)) : 0;
    const size_t szOk = needOkDev ? align256(nCyc) : 0;
    const size_t szSt = po.stance ? align256(static_cast<size_t>(B) * 12 * sizeof(double)) : 0;
    const size_t szSel = po.selected ? align256(nRec * sizeof(fpe_selected_foothold)) : 0;
    const size_t szPs = po.pose_status ? align256(static_cast<size_t>(B)) : 0;
    const size_t szPk = po.selected_packed ? align256(nRec * sizeof(fpe_selected_packed)) : 0;
    const size_t szOf = (oout && oout->footholds) ? align256(nRec * sizeof(fpe_opt_foothold)) : 0;
    const size_t szOc = (oout && oout->cycles) ? align256(nCyc * sizeof(fpe_opt_cycle)) : 0;
    const size_t szOg = (oout && oout->gate_fail_cycle) ? align256(static_cast<size_t>(B)) : 0;
    const size_t szOr = (oout && oout->rows_after) ? align256(static_cast<size_t>(B) * 2 * sizeof(double)) : 0;
    // A small call that runs the plan AND the opt track (the service: one pose) launches the two kernels side by side: the opt
    // track's chain needs the nominal track's cycle flags only to decide whether a cycle's result is committed, and the flags are
    // all 1 unless a nominal search fails — the chain runs on a second stream on flags of 1 (szSpec) while the plan kernel produces
    // the real ones, and runs again, after it, in the rare call whose flags turn out otherwise (round 5: the plan kernel's ~18 us
    // off the service's latency).
    const bool specWanted = runPlan && oout != nullptr && !cycleOkIn && B <= 4;
    const size_t szSpec = specWanted ? align256(nCyc) : 0;
    const size_t szDone = specWanted ? 256 : 0;  // the chain's completion word (one-pose calls: polled by the host, below)
    const size_t total = szPose + szNom + szCen + szDef + szOk + szSt + szSel + szPs + szPk + szOf + szOc + szOg + szOr + szSpec + szDone;
    if (!h) return fail(FPE_E_INVALID_ARG, "null handle or params");
    CallPlan cp;  // declared before the lease: on an early exit the lease waits for the stream BEFORE the snapshot / bit planes are released
    CtxLease lease(h->ctxPool);
    CallCtx& cx = *lease.ctx;
    FPE_HIP(hipSetDevice(h->device));
    FPE_HIP(cx.reserve(total));
    cx.inFlight = true;  // (prepare_call may already queue a bit-plane build)
    FPE_HT_MARK(1);
    int rc = prepare_call(h, params, maxRadius, cp, cx.stream, runPlan);
    if (rc != FPE_OK) return rc;
    FPE_HT_MARK(2);
    if (gateGeom) *gateGeom = GateGeom{cp.snap->g, cp.pc.isosLen, cp.pc.isosWid, cp.pc.drift};
    fpe::OptConsts oc;
    if (oout) {
        rc = prepare_opt(params, opt, cp, maxRadius, oc);
        // optOptional: the opt track was asked for its gate verdict only (fpe_plan_service* under service_opt_gate 1 / 2) and
        // is not supported for this geometry: the plan's own products must not fail with it
        if (rc == FPE_E_UNSUPPORTED && optDropped && runPlan) {
            *optDropped = true;
            oout = nullptr;
        } else if (rc != FPE_OK) {
            return rc;
        }
    }
    unsigned char* dp = cx.dev;
    unsigned char* hp = cx.pinned;
    size_t off = szPose;
    const size_t oNom = off; off += szNom;
    const size_t oCen = off; off += szCen;
    const size_t oDef = off; off += szDef;
    const size_t oOk = off; off += szOk;
    const size_t oSt = off; off += szSt;
    const size_t oSel = off; off += szSel;
    const size_t oPs = off; off += szPs;
    const size_t oPk = off; off += szPk;
    const size_t oOf = off; off += szOf;
    const size_t oOc = off; off += szOc;
    const size_t oOg = off; off += szOg;
    const size_t oOr = off; off += szOr;
    const size_t oSpec = off; off += szSpec;
    const size_t oDone = off;
    std::memcpy(hp, poses, static_cast<size_t>(B) * sizeof(fpe_pose));
    if (oout && cycleOkIn) std::memcpy(hp + oOk, cycleOkIn, nCyc);
    // Small calls (the plan_global_footholds service: one pose) skip both DMA copies: the kernels read the
    // poses from, and write their few KB of results straight into, the pinned (coherent, device-mapped) host
    // arena — two copy-engine round trips (~10 us each) less on a call whose kernel runs ~25 us.
    const bool zeroCopy = total <= kZeroCopyBytes;
    bool speculate = specWanted && zeroCopy && oout != nullptr;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (!h->tuning.serviceOverlap) speculate = false;
    }
    const fpe_pose* dPoses = nullptr;  // where the kernels read the poses (null: the head of the device arena)
    if (zeroCopy) {
        void* arena = nullptr;
        FPE_HIP(hipHostGetDevicePointer(&arena, hp, 0));
        dp = static_cast<unsigned char*>(arena);
    } else {
#ifndef FPE_POSES_BY_DMA
        // the poses are read by the kernels straight from the pinned arena (device-mapped): 64 bytes per pose, once, at the start
        // of its workgroup — a DMA transfer ahead of the launch costs its fixed fifteen microseconds before anything runs
        void* mapped = nullptr;
        FPE_HIP(hipHostGetDevicePointer(&mapped, hp, 0));
        dPoses = static_cast<const fpe_pose*>(mapped);
#else
        FPE_HIP(hipMemcpyAsync(dp, hp, static_cast<size_t>(B) * sizeof(fpe_pose), hipMemcpyHostToDevice, cx.stream));
#endif
        if (oout && cycleOkIn) FPE_HIP(hipMemcpyAsync(dp + oOk, hp + oOk, nCyc, hipMemcpyHostToDevice, cx.stream));
    }
    fpe_opt_out od;
    std::memset(&od, 0, sizeof(od));
    // One-pose overlapped calls (the service): the chain's last instruction writes a completion word into the pinned arena and the
    // host POLLS it (tuning key service_poll, default 1) instead of waiting for the side stream's completion signal.
    bool pollDone = false;
    uint32_t doneValue = 0;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        pollDone = speculate && B == 1 && h->tuning.servicePoll != 0;
    }
    const auto queue_opt = [&]() -> int {
        if (!oout) return FPE_OK;
        if (oout->footholds) od.footholds = reinterpret_cast<fpe_opt_foothold*>(dp + oOf);
        if (oout->cycles) od.cycles = reinterpret_cast<fpe_opt_cycle*>(dp + oOc);
        if (oout->gate_fail_cycle) od.gate_fail_cycle = dp + oOg;
        if (oout->rows_after) od.rows_after = reinterpret_cast<double*>(dp + oOr);
        hipStream_t os = cx.stream;
        const unsigned char* okDev = dp + oOk;
        uint32_t* doneDev = nullptr;
        if (speculate && pollDone) {
            doneValue = ++cx.doneSeq ? cx.doneSeq : ++cx.doneSeq;  // (never 0, never the value the word holds from the last call)
            doneDev = reinterpret_cast<uint32_t*>(dp + oDone);
        }
        if (speculate) {
            std::memset(hp + oSpec, 1, nCyc);
            FPE_HIP(cx.side_stream(&os));
            // the chain reads both layers: the side stream waits for an asynchronous upload (fpe_upload_map_device) exactly as
            // prepare_call made cx.stream wait (the speculative chain used to start while the canonicalise / copy
            // kernels of the snapshot were still writing d_trav / d_elev)
#ifndef FPE_TEST_NO_SIDE_WAIT  // (test builds only: the suite checks that its upload-race test FAILS without this wait)
            FPE_HIP(cp.snap->wait_ready(os));
#endif
            okDev = dp + oSpec;
        }
        FPE_HIP(fpe::launch_opt_track(dev_map(*cp.snap), cp.pc, oc, dPoses ? dPoses : reinterpret_cast<const fpe_pose*>(dp), B, n_cycles, okDev, od, os, doneDev, doneValue));
        return FPE_OK;
    };
    // (the overlapped form queues the LONGER kernel first — the chain is the call's critical path; the plan kernel's launch then costs
    // it nothing)
    FPE_HT_MARK(3);
    if (speculate) {
        rc = queue_opt();
        if (rc != FPE_OK) return rc;
        // A polled call never enters a stream wait, and it is a stream wait (or query) that makes the runtime hand queued work to
        // the hardware at once: without it the chain was seen to END 12 us later than under hipStreamSynchronize (measured, round 6:
        // profiles/round6_service_latency.txt).  One non-blocking query per stream, right behind its launch.
        if (pollDone && hipStreamQuery(cx.side) != hipSuccess) (void)hipGetLastError();
    }
    FPE_HT_MARK(4);
    if (runPlan) {
        fpe_plan_out d;
        std::memset(&d, 0, sizeof(d));
        if (po.nominal) d.nominal = reinterpret_cast<fpe_foothold*>(dp + oNom);
        if (po.centroid) d.centroid = reinterpret_cast<fpe_centroid_foothold*>(dp + oCen);
        if (po.default_next) d.default_next = reinterpret_cast<double*>(dp + oDef);
        if (needOkDev) d.cycle_ok = dp + oOk;
        if (po.stance) d.stance = reinterpret_cast<double*>(dp + oSt);
        if (po.selected) d.selected = reinterpret_cast<fpe_selected_foothold*>(dp + oSel);
        if (po.pose_status) d.pose_status = dp + oPs;
        if (po.selected_packed) d.selected_packed = reinterpret_cast<fpe_selected_packed*>(dp + oPk);
        // every (pose, cycle, leg) record is written by the kernel (trot: all legs each cycle; walk: each
        // leg in its phase), so the buffers need no clearing
        rc = launch_plan(h, cp, dPoses ? dPoses : reinterpret_cast<const fpe_pose*>(dp), B, n_cycles, d, cx.stream);
        if (rc != FPE_OK) return rc;
    }
    if (!speculate) {
        rc = queue_opt();
        if (rc != FPE_OK) return rc;
    }
    if (pollDone && hipStreamQuery(cx.stream) != hipSuccess) (void)hipGetLastError();  // (see the side stream's query above)
    FPE_HT_MARK(5);
    // ---- results to the caller ----
    struct Seg {
        size_t off, len;
        void* dst;
    };
    Seg segs[13];
    int nSeg = 0;
    auto add = [&](size_t o, size_t len, void* dst) {
        if (dst && len) segs[nSeg++] = Seg{o, len, dst};
    };
    add(oNom, nRec * sizeof(fpe_foothold), po.nominal);
    add(oCen, nRec * sizeof(fpe_centroid_foothold), po.centroid);
    add(oDef, nRec * 3 * sizeof(double), po.default_next);
    add(oOk, nCyc, po.cycle_ok);
    add(oSt, static_cast<size_t>(B) * 12 * sizeof(double), po.stance);
    add(oSel, nRec * sizeof(fpe_selected_foothold), po.selected);
    add(oPs, static_cast<size_t>(B), po.pose_status);
    add(oPk, nRec * sizeof(fpe_selected_packed), po.selected_packed);
    if (oout && szOf) add(oOf, nRec * sizeof(fpe_opt_foothold), oout->footholds);
    if (oout && szOc) add(oOc, nCyc * sizeof(fpe_opt_cycle), oout->cycles);
    if (oout && szOg) add(oOg, static_cast<size_t>(B), oout->gate_fail_cycle);
    if (oout && szOr) add(oOr, static_cast<size_t>(B) * 2 * sizeof(double), oout->rows_after);
    if (zeroCopy) {  // the kernels wrote into the pinned arena itself
        bool polled = false;
        if (pollDone && doneValue != 0) {
            // The chain is the call's critical path (~90 us against the plan kernel's ~20): wait for ITS word first.  Bounded: after
            // 5 ms the ordinary synchronisation below takes over (a word that never arrives must not hang the caller).
            volatile uint32_t* word = reinterpret_cast<volatile uint32_t*>(hp + oDone);
            const auto t0 = std::chrono::steady_clock::now();
            int spins = 0;
            while (*word != doneValue) {
                __builtin_ia32_pause();
                if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
            }
            polled = *word == doneValue;
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        // (the plan kernel ended long before the chain; when the runtime has not yet seen that, wait for it the ordinary way)
        if (!(polled && hipStreamQuery(cx.stream) == hipSuccess)) {
            (void)hipGetLastError();
            FPE_HIP(hipStreamSynchronize(cx.stream));
        }
        FPE_HT_MARK(6);
        if (oout && speculate) {
            // polled: every product of the chain is in the arena (its stores precede the word); the stream's own completion is
            // looked at without waiting — the context's next call queues behind it in stream order
            if (!polled) FPE_HIP(hipStreamSynchronize(cx.side));
            else if (hipStreamQuery(cx.side) != hipSuccess) (void)hipGetLastError();
            FPE_HT_MARK(7);
            bool allCommitted = true;
            for (size_t c = 0; c < nCyc; ++c) allCommitted = allCommitted && hp[oOk + c] != 0;
            if (!allCommitted) {  // a nominal search failed in some cycle: the chain again, on the flags as they are
                FPE_HIP(fpe::launch_opt_track(dev_map(*cp.snap), cp.pc, oc, dPoses ? dPoses : reinterpret_cast<const fpe_pose*>(dp), B, n_cycles, dp + oOk, od,
                                              cx.stream));
                FPE_HIP(hipStreamSynchronize(cx.stream));
            }
        }
        cx.inFlight = false;
        for (int k = 0; k < nSeg; ++k) std::memcpy(segs[k].dst, hp + segs[k].off, segs[k].len);
#ifdef FPE_HOST_TIMING
        if (ht.last < 7) ht.mark(7);
        ht.mark(8);
        ht.commit();
#endif
        return FPE_OK;
    }
    // Only what the caller asked for crosses PCIe.  A product whose destination is pinned / registered host memory
    // (fpe_host_alloc) is written by DMA directly; the others go through the pinned arena in chunks, and the copy of a
    // chunk into the caller's array runs on the copy pool while the next chunks are still in flight.
    // Pinned destinations that lie behind one another exactly as the products do in the device arena (no padding in between:
    // FootholdPlanner.plan_outputs(pinned=True) carves them out of ONE fpe_host_alloc block in this order) leave in ONE copy:
    // a DMA transfer has a fixed cost of some ten microseconds, and seven of them in a row are a third of a 4 096-pose call.
    size_t staged = 0;
    bool pinnedSeg[13];
    for (int k = 0; k < nSeg; ++k) pinnedSeg[k] = is_pinned_host(segs[k].dst);
    for (int k = 0; k < nSeg; ++k) {
        if (!pinnedSeg[k]) {
            staged += segs[k].len;
            continue;
        }
        size_t len = segs[k].len;
        int last = k;
        while (last + 1 < nSeg && pinnedSeg[last + 1] && segs[last + 1].off == segs[last].off + segs[last].len &&
               static_cast<unsigned char*>(segs[last + 1].dst) == static_cast<unsigned char*>(segs[last].dst) + segs[last].len) {
            ++last;
            len += segs[last].len;
        }
        FPE_HIP(hipMemcpyAsync(segs[k].dst, dp + segs[k].off, len, hipMemcpyDeviceToHost, cx.stream));
        for (int q = k; q <= last; ++q) segs[q].dst = nullptr;
        k = last;
    }
    // A DMA transfer has a fixed cost of some ten microseconds: few, large pieces (a quarter of what is staged, at least
    // 1 MB), and the copy of a piece into the caller's array split over the pool's threads so that the last piece's copy — the
    // only one that is not hidden behind a later transfer — is short.
    constexpr size_t kTargetChunks = 4;
    const size_t chunk = std::max<size_t>(1u << 20, (staged / kTargetChunks + 4095) & ~static_cast<size_t>(4095));
    // DMA pieces are cut from RUNS of staged products that are neighbours in the arena (alignment padding rides along: both
    // arenas are the engine's own), not from single products: four transfers for seven products, not nine.
    struct Piece {
        size_t off, len;
    };
    std::vector<Piece> pieces;
    for (int k = 0; k < nSeg; ++k) {
        if (!segs[k].dst) continue;
        int last = k;
        while (last + 1 < nSeg && segs[last + 1].dst && segs[last + 1].off == align256(segs[last].off + segs[last].len)) ++last;
        const size_t end = segs[last].off + segs[last].len;
        for (size_t c = segs[k].off; c < end; c += chunk) pieces.push_back(Piece{c, std::min(chunk, end - c)});
        k = last;
    }
    for (size_t k = 0; k < pieces.size(); ++k) {
        hipEvent_t ev;
        FPE_HIP(cx.event(k, &ev));
        FPE_HIP(hipMemcpyAsync(hp + pieces[k].off, dp + pieces[k].off, pieces[k].len, hipMemcpyDeviceToHost, cx.stream));
        FPE_HIP(hipEventRecord(ev, cx.stream));
    }
    CopyPool::Batch batch;
    hipError_t evErr = hipSuccess;
    for (size_t k = 0; k < pieces.size(); ++k) {
        const hipError_t e = hipEventSynchronize(cx.events[k]);
        if (e != hipSuccess) {
            evErr = e;
            break;
        }
        // what the piece holds of each staged product, into the caller's array
        const size_t p0 = pieces[k].off, p1 = p0 + pieces[k].len;
        for (int q = 0; q < nSeg; ++q) {
            if (!segs[q].dst) continue;
            const size_t a0 = std::max(p0, segs[q].off), a1 = std::min(p1, segs[q].off + segs[q].len);
            if (a0 >= a1) continue;
            unsigned char* dst = static_cast<unsigned char*>(segs[q].dst) + (a0 - segs[q].off);
            const size_t len = a1 - a0;
            if (len >= (256u << 10)) {
                const size_t parts = std::min<size_t>(h->copyPool.width(), (len + (256u << 10) - 1) / (256u << 10));
                const size_t step = ((len + parts - 1) / parts + 63) & ~static_cast<size_t>(63);
                for (size_t c = 0; c < len; c += step) h->copyPool.submit(batch, dst + c, hp + a0 + c, std::min(step, len - c));
            } else {
                std::memcpy(dst, hp + a0, len);
            }
        }
    }
    CopyPool::wait(batch);
    if (evErr != hipSuccess) return fail_hip(evErr, "result copy");  // (the lease waits for the stream)
    FPE_HIP(hipStreamSynchronize(cx.stream));  // the direct (pinned-destination) copies
    cx.inFlight = false;
    return FPE_OK;
}
}  // namespace

int fpe_host_alloc(fpe_handle h, size_t bytes, void** out) {
    if (!h || !out || bytes == 0) return fail(FPE_E_INVALID_ARG, "null argument");
    *out = nullptr;
    FPE_HIP(hipSetDevice(h->device));
    FPE_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return FPE_OK;
}

int fpe_host_free(fpe_handle h, void* p) {
    if (!h) return fail(FPE_E_INVALID_ARG, "null handle");
    if (!p) return FPE_OK;
    FPE_HIP(hipSetDevice(h->device));
    FPE_HIP(hipHostFree(p));
    return FPE_OK;
}

int fpe_plan(fpe_handle h, const fpe_params* params, const fpe_pose* poses, int32_t B, int32_t n_cycles,
             const fpe_plan_out* out) {
    if (!out) return fail(FPE_E_INVALID_ARG, "null argument");
    return plan_host(h, params, nullptr, poses, B, n_cycles, out, nullptr, nullptr);
}

int fpe_plan_opt(fpe_handle h, const fpe_params* params, const fpe_opt_params* opt, const fpe_pose* poses, int32_t B,
                 int32_t n_cycles, const uint8_t* cycle_ok, const fpe_opt_out* out) {
    if (!out) return fail(FPE_E_INVALID_ARG, "null argument");
    return plan_host(h, params, opt, poses, B, n_cycles, nullptr, cycle_ok, out);
}

int fpe_plan_opt_device(fpe_handle h, const fpe_params* params, const fpe_opt_params* opt, const fpe_pose* d_poses, int32_t B,
                        int32_t n_cycles, const uint8_t* d_cycle_ok, const fpe_opt_out* d_out, void* stream) {
    if (!d_poses || !d_out || !d_cycle_ok) return fail(FPE_E_INVALID_ARG, "null argument");
    if (B <= 0 || n_cycles <= 0 || n_cycles > 255) return fail(FPE_E_INVALID_ARG, "B and n_cycles must be in [1, ..] / [1, 255]");
    hipStream_t st = static_cast<hipStream_t>(stream);
    CallPlan cp;
    int rc = prepare_call(h, params, 0.0f, cp, st, false);
    if (rc != FPE_OK) return rc;
    fpe::OptConsts oc;
    rc = prepare_opt(params, opt, cp, std::max(params->searchRadius, cp.pc.maxSearchRadius), oc);
    if (rc != FPE_OK) return rc;
    FPE_HIP(fpe::launch_opt_track(dev_map(*cp.snap), cp.pc, oc, d_poses, B, n_cycles, d_cycle_ok, *d_out, st));
    cp.snap->note_async_use();
    return FPE_OK;
}

int fpe_search_legs_device(fpe_handle h, const fpe_params* params, const fpe_leg_query* d_queries, int32_t n,
                           fpe_foothold* d_out, void* stream) {
    if (!d_queries || !d_out) return fail(FPE_E_INVALID_ARG, "null argument");
    if (n <= 0) return fail(FPE_E_INVALID_ARG, "n must be positive");
    hipStream_t st = static_cast<hipStream_t>(stream);
    CallPlan cp;
    int rc = prepare_call(h, params, 0.0f, cp, st, false);
    if (rc != FPE_OK) return rc;
    FPE_HIP(fpe::launch_search_legs(dev_map(*cp.snap), cp.pc, h->lut(), d_queries, n, d_out, st));
    cp.snap->note_async_use();
    return FPE_OK;
}

int fpe_search_legs(fpe_handle h, const fpe_params* params, const fpe_leg_query* queries, int32_t n, fpe_foothold* out) {
    if (!queries || !out) return fail(FPE_E_INVALID_ARG, "null argument");
    if (n <= 0) return fail(FPE_E_INVALID_ARG, "n must be positive");
    float maxRadius = 0.0f;
    for (int k = 0; k < n; ++k) {
        if (!std::isfinite(queries[k].cx) || !std::isfinite(queries[k].cy) || std::fabs(queries[k].cx) > 1e6 ||
            std::fabs(queries[k].cy) > 1e6)
            return fail(FPE_E_INVALID_ARG, "non-finite query centre");
        if (!(queries[k].search_radius >= 0.0f) || !std::isfinite(queries[k].search_radius))
            return fail(FPE_E_INVALID_ARG, "bad search radius");
        if (queries[k].n_vertices < 0 || queries[k].n_vertices > FPE_MAX_POLYGON_VERTICES)
            return fail(FPE_E_INVALID_ARG, "too many polygon vertices");
        maxRadius = std::max(maxRadius, queries[k].search_radius);
    }
    if (!h) return fail(FPE_E_INVALID_ARG, "null handle or params");
    const size_t szQ = align256(static_cast<size_t>(n) * sizeof(fpe_leg_query));
    const size_t szO = align256(static_cast<size_t>(n) * sizeof(fpe_foothold));
    CallPlan cp;  // (before the lease, as in plan_host)
    CtxLease lease(h->ctxPool);
    CallCtx& cx = *lease.ctx;
    FPE_HIP(hipSetDevice(h->device));
    FPE_HIP(cx.reserve(szQ + szO));
    cx.inFlight = true;
    int rc = prepare_call(h, params, maxRadius, cp, cx.stream, false);
    if (rc != FPE_OK) return rc;
    std::memcpy(cx.pinned, queries, static_cast<size_t>(n) * sizeof(fpe_leg_query));
    FPE_HIP(hipMemcpyAsync(cx.dev, cx.pinned, static_cast<size_t>(n) * sizeof(fpe_leg_query), hipMemcpyHostToDevice, cx.stream));
    FPE_HIP(fpe::launch_search_legs(dev_map(*cp.snap), cp.pc, h->lut(), reinterpret_cast<const fpe_leg_query*>(cx.dev), n,
                                    reinterpret_cast<fpe_foothold*>(cx.dev + szQ), cx.stream));
    FPE_HIP(hipMemcpyAsync(cx.pinned + szQ, cx.dev + szQ, static_cast<size_t>(n) * sizeof(fpe_foothold), hipMemcpyDeviceToHost, cx.stream));
    FPE_HIP(hipStreamSynchronize(cx.stream));
    cx.inFlight = false;
    std::memcpy(out, cx.pinned + szQ, static_cast<size_t>(n) * sizeof(fpe_foothold));
    return FPE_OK;
}

int fpe_last_service_gate(fpe_handle h, fpe_service_gate* out) {
    if (!h || !out) return fail(FPE_E_INVALID_ARG, "null argument");
    if (g_gateEngine != h) return fail(FPE_E_INVALID_ARG, "no fpe_plan_service* call was made with this engine on this thread");
    *out = g_gate;
    return FPE_OK;
}

int fpe_plan_service(fpe_handle h, const fpe_params* params, const double initial_position[3], uint8_t gait_cycles,
                     fpe_global_footholds* response) {
    return fpe_plan_service_ex(h, params, initial_position, gait_cycles, response, nullptr, nullptr, nullptr);
}

int fpe_plan_service_ex(fpe_handle h, const fpe_params* params, const double initial_position[3], uint8_t gait_cycles,
                        fpe_global_footholds* response, fpe_global_footholds* centroid, double* default_footholds,
                        int32_t* n_default_rows) {
    return fpe_plan_service_report(h, params, initial_position, gait_cycles, response, centroid, default_footholds,
                                   n_default_rows, nullptr, nullptr);
}

int fpe_plan_service_report(fpe_handle h, const fpe_params* params, const double initial_position[3], uint8_t gait_cycles,
                            fpe_global_footholds* response, fpe_global_footholds* centroid, double* default_footholds,
                            int32_t* n_default_rows, fpe_track_report* nominal_report, fpe_track_report* centroid_report) {
    return fpe_plan_service_opt(h, params, nullptr, initial_position, gait_cycles, response, centroid, default_footholds, n_default_rows,
                                nominal_report, centroid_report, nullptr, nullptr, nullptr);
}

int fpe_plan_service_opt(fpe_handle h, const fpe_params* params, const fpe_opt_params* opt, const double initial_position[3],
                         uint8_t gait_cycles, fpe_global_footholds* response, fpe_global_footholds* centroid,
                         double* default_footholds, int32_t* n_default_rows, fpe_track_report* nominal_report,
                         fpe_track_report* centroid_report, fpe_global_footholds* opt_msg, fpe_track_report* opt_report,
                         fpe_opt_cycle* opt_cycles) {
    g_gate = fpe_service_gate{255, FPE_GATE_NONE, 0, 0, {0, 0, 0, 0}, 0.0, 0.0};
    g_gateEngine = h;
    if (!initial_position || !response) return fail(FPE_E_INVALID_ARG, "null argument");
    if (default_footholds && !n_default_rows) return fail(FPE_E_INVALID_ARG, "n_default_rows is required with default_footholds");
    fpe_pose pose;
    std::memset(&pose, 0, sizeof(pose));
    pose.position[0] = initial_position[0];
    pose.position[1] = initial_position[1];
    pose.position[2] = initial_position[2];
    const int N = gait_cycles;
    double stance[12];
    std::vector<fpe_foothold> nominal(static_cast<size_t>(N) * 4);
    std::vector<fpe_centroid_foothold> cen(static_cast<size_t>(N) * 4);
    std::vector<fpe_opt_foothold> optf(static_cast<size_t>(N) * 4);
    std::vector<double> dflt(static_cast<size_t>(N) * 12);
    std::vector<uint8_t> ok(static_cast<size_t>(N));
    if (N == 0) {
        // the reference's loop body never runs (cpp:762); the stance comes from initialize()
        if (!h || !params) return fail(FPE_E_INVALID_ARG, "null handle or params");
        fpe::PlanConsts pc;
        fpe::derive_constants(*params, fpe::make_geom(1, 1, 1.0, 0.0, 0.0), params->searchRadius, fpe::Tuning(), pc);
        for (int l = 0; l < 4; ++l) {
            double sx = (l == 0 || l == 3) ? pc.LbHalf : -pc.LbHalf;
            double sy = (l <= 1) ? pc.WbHalfNeg : pc.WbHalfPos;
            double sz = 0;
            sx += pose.position[0];
            sy += pose.position[1];
            sz += pose.position[2];
            stance[l * 3] = sx; stance[l * 3 + 1] = sy; stance[l * 3 + 2] = sz;
        }
    } else {
        fpe_plan_out out;
        std::memset(&out, 0, sizeof(out));
        out.nominal = nominal.data();
        out.cycle_ok = ok.data();
        out.stance = stance;
        if (centroid || centroid_report) out.centroid = cen.data();
        if (default_footholds) out.default_next = dflt.data();
        // The handler's gate (cpp:920-934; include/fpe.h "service-shaped call"): cycle 0 by the plan kernels, the y side of
        // every cycle here on the host, the x side of cycles >= 1 by the opt track's chain — build-defined, run only when
        // an opt product is asked for or fpe_set_tuning("service_opt_gate", 1 | 2) wants its verdict.
        fpe_opt_out oout;
        std::memset(&oout, 0, sizeof(oout));
        uint8_t gateFail = 255;
        oout.gate_fail_cycle = &gateFail;
        // (the track's records only when a product is made of them: a call that wants the gate's verdict alone spares the chain
        // its heights — getFootholdMeanHeight on the gait-cycle submap, an elevation round trip per cycle — and its stores)
        oout.footholds = (opt_msg || opt_report || centroid_report) ? optf.data() : nullptr;
        oout.cycles = opt_cycles;
        double rowsAfter[2] = {0.0, 0.0};
        oout.rows_after = rowsAfter;
        int optGate = 2;
        if (h) {
            std::lock_guard<std::mutex> lk(h->mu);
            optGate = h->tuning.serviceOptGate;
        }
        const bool wantOptProduct = opt_msg || opt_report || opt_cycles || centroid_report;
        const bool runChain = wantOptProduct || optGate != 0;
        uint8_t poseStatus = 0;
        out.pose_status = &poseStatus;
        GateGeom gg;
        bool optDropped = false;
        int rc = plan_host(h, params, opt, &pose, 1, N, &out, nullptr, runChain ? &oout : nullptr, &gg, wantOptProduct ? nullptr : &optDropped);
        if (rc != FPE_OK) return rc;
        // exact verdicts: the first cycle (stance feet), then the lateral side of every cycle
        int lateral = 255;
        {
            double adjY = 0.0;  // ajustedPose_[1], cpp:759
            for (int c = 0; c < N; ++c) {
                const double py = pose.position[1] + adjY;  // cpp:2329
                // (x at the map's centre: the x side of getSubmap passes there, what is left is the y side)
                const bool okY = std::fabs(py) <= 1e6 && std::fabs(gg.g.posX) <= 1e6 && fpe::submap_info(gg.g, gg.g.posX, py, gg.isosLen, gg.isosWid).ok;
                if (!okY) {
                    lateral = c;
                    break;
                }
                adjY += gg.drift;  // cpp:1578
            }
        }
        const bool chainRan = runChain && !optDropped;
        fpe_service_gate gate = g_gate;  // (reset at entry)
        gate.chain_ran = chainRan ? 1 : 0;
        gate.lf_current_row = chainRan ? rowsAfter[0] : 0.0;
        gate.rh_current_row = chainRan ? rowsAfter[1] : 0.0;
        auto verdict = [&](int cycle, int kind) {
            gate.fail_cycle = static_cast<uint8_t>(cycle);
            gate.fail_kind = static_cast<uint8_t>(kind);
        };
        if (poseStatus & FPE_POSE_OPT_SUBMAP_FAILED) verdict(0, FPE_GATE_CYCLE0);
        else if (lateral != 255) verdict(lateral, FPE_GATE_LATERAL);  // (exact kinds first: the reference refuses in cycle `lateral` at the latest)
        else if (chainRan && gateFail != 255) verdict(gateFail, gateFail == 0 ? FPE_GATE_CYCLE0 : FPE_GATE_BUILD_DEFINED);
        const bool refuse = gate.fail_kind == FPE_GATE_CYCLE0 || gate.fail_kind == FPE_GATE_LATERAL ||
                            (gate.fail_kind == FPE_GATE_BUILD_DEFINED && optGate == 2);
        gate.returned_false = refuse ? 1 : 0;
        g_gate = gate;
        if (refuse) {
            // getGaitCycleSearchGridMap fails in cycle gate.fail_cycle: the reference's handler logs "Failed to get gait-cycle
            // search gridmap." and returns false (cpp:931-934); the ROS response is never assigned (cpp:1588 is not reached)
            std::memset(response, 0, sizeof(*response));
            if (centroid) std::memset(centroid, 0, sizeof(*centroid));
            if (n_default_rows) *n_default_rows = 0;
            if (nominal_report) std::memset(nominal_report, 0, sizeof(*nominal_report));
            if (centroid_report) std::memset(centroid_report, 0, sizeof(*centroid_report));
            if (opt_msg) std::memset(opt_msg, 0, sizeof(*opt_msg));
            if (opt_report) std::memset(opt_report, 0, sizeof(*opt_report));
            static const char* const kinds[] = {"", "first gait cycle", "lateral side", "x side, build-defined optimiser"};
            return fail(FPE_E_SERVICE_FALSE, "getGaitCycleSearchGridMap: getSubmap failed in gait cycle " + std::to_string(static_cast<int>(gate.fail_cycle)) +
                                                 " (" + kinds[gate.fail_kind] + "; cpp:931-934)");
        }
    }
    // The call answers although the opt track's chain stopped at its gate (modes 0 / 1: the verdict is advisory): the chain's
    // products end where it stopped and the reference would not have published them — its handler returns false at that gate,
    // whatever its optimiser (cpp:931-934).  They are handed back EMPTY, never as the truncated track of an aborted chain; the
    // kind and cycle are in fpe_last_service_gate.
    const bool chainAborted = g_gate.chain_ran && g_gate.fail_kind == FPE_GATE_BUILD_DEFINED;
    fpe::assemble_global_footholds(nominal.data(), ok.data(), stance, N, response);
    if (centroid) fpe::assemble_centroid_footholds(cen.data(), ok.data(), stance, N, centroid);
    if (opt_msg) {
        if (chainAborted) std::memset(opt_msg, 0, sizeof(*opt_msg));
        else fpe::assemble_opt_footholds(optf.data(), ok.data(), stance, N, opt_msg);
    }
    if (nominal_report || centroid_report || opt_report) {
        if (!params) return fail(FPE_E_INVALID_ARG, "null params");
        std::vector<double> xyz(static_cast<size_t>(N) * 12);
        if (nominal_report) {
            for (size_t k = 0; k < static_cast<size_t>(N) * 4; ++k) {
                xyz[k * 3] = nominal[k].x;
                xyz[k * 3 + 1] = nominal[k].y;
                xyz[k * 3 + 2] = static_cast<double>(nominal[k].z);
            }
            fpe::assemble_track_report(xyz.data(), ok.data(), stance, N, *params, nominal_report);
        }
        if (centroid_report || opt_report) {
            std::unique_ptr<fpe_track_report> optRep(new (std::nothrow) fpe_track_report);
            if (!optRep) return fail(FPE_E_NOMEM, "out of host memory");
            for (size_t k = 0; k < static_cast<size_t>(N) * 4; ++k) {
                xyz[k * 3] = optf[k].x;
                xyz[k * 3 + 1] = optf[k].y;
                xyz[k * 3 + 2] = static_cast<double>(optf[k].z);
            }
            fpe::assemble_track_report(xyz.data(), ok.data(), stance, N, *params, optRep.get());
            if (chainAborted) std::memset(optRep.get(), 0, sizeof(fpe_track_report));  // (no opt path, no opt KPIs; the centroid path is then the centroid track's own)
            if (opt_report) std::memcpy(opt_report, optRep.get(), sizeof(*opt_report));
            if (centroid_report) {
                for (size_t k = 0; k < static_cast<size_t>(N) * 4; ++k) {
                    xyz[k * 3] = cen[k].x;
                    xyz[k * 3 + 1] = cen[k].y;
                    xyz[k * 3 + 2] = static_cast<double>(cen[k].z);
                }
                fpe::assemble_track_report(xyz.data(), ok.data(), stance, N, *params, centroid_report);
                if (!chainAborted) fpe::interleave_centroid_path(centroid_report, *optRep);  // cpp:946: the opt track pushes onto the same path
            }
        }
    }
    if (default_footholds) {
        int rows = 0;
        std::memcpy(default_footholds, stance, 12 * sizeof(double));  // cpp:666-671
        rows = 1;
        for (int g = 0; g < N; ++g)
            if (ok[g]) {  // cpp:1344-1348: appended only for committed cycles
                std::memcpy(default_footholds + static_cast<size_t>(rows) * 12, dflt.data() + static_cast<size_t>(g) * 12,
                            12 * sizeof(double));
                ++rows;
            }
        *n_default_rows = rows;
    }
    return FPE_OK;
}

}  // extern "C"

#ifdef FPE_HOST_TIMING
// mean microseconds per zero-copy plan_host call between its marks: [1] lease + reserve, [2] prepare_call, [3] opt constants + pose
// copy + device pointers, [4] chain queued (overlapped form), [5] plan kernel queued, [6] main stream synchronised, [7] side stream
// synchronised, [8] results copied out; out[0] = calls
extern "C" int fpe_debug_host_timing(double* out, int reset) {
    out[0] = static_cast<double>(g_hostN);
    for (int k = 1; k < 16; ++k) out[k] = g_hostN ? g_hostT[k] / static_cast<double>(g_hostN) : 0.0;
    if (reset) {
        for (double& v : g_hostT) v = 0.0;
        g_hostN = 0;
    }
    return 0;
}
#endif
