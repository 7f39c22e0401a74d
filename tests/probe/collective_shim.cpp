// collective_shim.cpp — TEST INFRASTRUCTURE, not product code and not RCCL.
//
// RCCL refuses a communicator with one device listed twice, and this pool's GPU boxes have one device: the part of
// fpe_multi_plan_device that only exists for n > 1 (csrc/fpe_multi.cpp: slot offsets of the padded in-place all-gather, the
// per-rank compaction copies, the staging buffer's reuse event, the stream ordering between a rank's plan kernel and the
// collective) could never run there.  This library stands in for librccl.so.1 behind the engine's run-time binding
// (FPE_RCCL_LIB=<this file's .so>, read once by fpe_multi.cpp) and implements the five entry points the gather uses with
// device-local copies, with the ordering guarantees the real collective gives:
//   * a rank's contribution is read only after everything queued on THAT rank's stream before the call (its plan kernel);
//   * a rank's stream continues only after every rank has finished reading its send buffer and writing its receive buffer
//     (stricter than RCCL — an all-gather completes on a rank once that rank's buffers are done — but it keeps a later
//     overwrite of a send buffer from racing a neighbour's read exactly as the real collective's completion does).
// It exports `fpe_test_collective_shim`, the marker by which fpe_multi.cpp allows one device to stand for several ranks.
// What it tests is the ENGINE's offsets, slots and stream order; it says nothing about RCCL or xGMI.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

extern "C" {

typedef struct ShimComm* ncclComm_t;
typedef int ncclResult_t;    // 0 = ncclSuccess
typedef int ncclDataType_t;  // the engine only sends ncclChar (0): one byte per element

int fpe_test_collective_shim = 1;
}

namespace {

struct Group {
    int n = 0;
    int live = 0;
};
struct Op {
    ShimComm* comm;
    const void* send;
    void* recv;
    size_t bytes;
    int root;  // -1: all-gather; >= 0: broadcast from root
    hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

}  // namespace

struct ShimComm {
    Group* group;
    int rank;
    int device;
};

namespace {

// Executes the ops of one group call: one op per rank, in any order.
ncclResult_t run(std::vector<Op>& ops) {
    if (ops.empty()) return 0;
    const int n = ops[0].comm->group->n;
    if (static_cast<int>(ops.size()) != n) return 5;  // ncclInvalidUsage: every rank must take part
    std::vector<Op*> byRank(static_cast<size_t>(n), nullptr);
    for (Op& o : ops) {
        if (o.comm->group != ops[0].comm->group || byRank[static_cast<size_t>(o.comm->rank)] || o.bytes != ops[0].bytes || o.root != ops[0].root) return 5;
        byRank[static_cast<size_t>(o.comm->rank)] = &o;
    }
    std::vector<hipEvent_t> ready(static_cast<size_t>(n)), done(static_cast<size_t>(n));
    hipError_t e = hipSuccess;
    for (int r = 0; r < n && e == hipSuccess; ++r) {
        e = hipSetDevice(byRank[static_cast<size_t>(r)]->comm->device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ready[static_cast<size_t>(r)], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done[static_cast<size_t>(r)], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(ready[static_cast<size_t>(r)], byRank[static_cast<size_t>(r)]->stream);
    }
    for (int k = 0; k < n && e == hipSuccess; ++k) {
        Op& dst = *byRank[static_cast<size_t>(k)];
        for (int r = 0; r < n && e == hipSuccess; ++r) {
            if (dst.root >= 0 && r != dst.root) continue;
            const Op& src = *byRank[static_cast<size_t>(r)];
            unsigned char* to = static_cast<unsigned char*>(dst.recv) + (dst.root >= 0 ? 0 : static_cast<size_t>(r) * dst.bytes);
            e = hipStreamWaitEvent(dst.stream, ready[static_cast<size_t>(r)], 0);
            if (e == hipSuccess && to != src.send) e = hipMemcpyAsync(to, src.send, dst.bytes, hipMemcpyDeviceToDevice, dst.stream);
        }
        if (e == hipSuccess) e = hipEventRecord(done[static_cast<size_t>(k)], dst.stream);
    }
    for (int k = 0; k < n && e == hipSuccess; ++k)
        for (int r = 0; r < n && e == hipSuccess; ++r)
            if (r != k) e = hipStreamWaitEvent(byRank[static_cast<size_t>(k)]->stream, done[static_cast<size_t>(r)], 0);
    for (int r = 0; r < n; ++r) {  // (an event destroyed while waits on it are queued is released when they have run)
        if (ready[static_cast<size_t>(r)]) (void)hipEventDestroy(ready[static_cast<size_t>(r)]);
        if (done[static_cast<size_t>(r)]) (void)hipEventDestroy(done[static_cast<size_t>(r)]);
    }
    return e == hipSuccess ? 0 : 1;  // ncclUnhandledCudaError
}

ncclResult_t submit(const Op& op) {
    g_ops.push_back(op);
    if (g_depth > 0) return 0;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run(ops);
}

}  // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devices) {
    if (!comms || n <= 0) return 4;  // ncclInvalidArgument
    Group* g = new Group();
    g->n = g->live = n;
    for (int k = 0; k < n; ++k) comms[k] = new ShimComm{g, k, devices ? devices[k] : k};
    return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return 0;
    if (--c->group->live == 0) delete c->group;
    delete c;
    return 0;
}

ncclResult_t ncclGroupStart() {
    ++g_depth;
    return 0;
}

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return 5;
    if (--g_depth > 0) return 0;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run(ops);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t comm, hipStream_t stream) {
    if (!send || !recv || !comm) return 4;
    return submit(Op{comm, send, recv, count, -1, stream});
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t, int root, ncclComm_t comm, hipStream_t stream) {
    if (!send || !recv || !comm || root < 0 || root >= comm->group->n) return 4;
    return submit(Op{comm, send, recv, count, root, stream});
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case 0: return "no error";
        case 1: return "shim: a HIP call failed";
        case 4: return "shim: invalid argument";
        case 5: return "shim: invalid usage (every rank of the group must post one matching operation)";
        default: return "shim: error";
    }
}

}  // extern "C"
