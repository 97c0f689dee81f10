// RCCL entry points of the C ABI ("bya_allgather_kv / bya_alltoall_router wrappers over RCCL for
// (e)") -- the two exchanges of the sharded denoise step for a host that drives the kernels through include/bya.h
// directly (the Python module of this package issues the same exchanges through torch.distributed, whose ``nccl``
// backend IS RCCL, because torch does not hand out its ncclComm_t):
//   * exchange A, K/V form: every rank contributes its rows of K and V and receives all rows (one grouped call);
//   * exchange A head-parallel form / exchange B (router repartition): an uneven all-to-all of bf16 elements expressed
//     as grouped point-to-point sends and receives -- xGMI is point to point, every element crosses one link once.
// Both only ENQUEUE on ``stream`` (give them a stream of their own to overlap with compute).  RCCL is resolved at
// first use from the library already mapped into the process (dlopen RTLD_NOLOAD first: one RCCL per process), so
// libbya_hip.so itself has no link-time dependency on it and loads on a box without RCCL.
#include "bya_common.h"
#include "../../include/bya.h"
#include <dlfcn.h>

namespace {

// The handful of RCCL declarations the two wrappers need, restated here so that the library builds on a box without
// the RCCL headers (values are ABI constants of NCCL >= 2.10 / RCCL, rccl.h: ncclSuccess = 0, ncclBfloat16 = 9).
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclBfloat16 = 9;

struct Rccl {
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*group_start)() = nullptr;
    ncclResult_t (*group_end)() = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*comm_rank)(const ncclComm_t, int*) = nullptr;
    bool ok = false;
};

const Rccl& rccl() {
    static const Rccl r = [] {
        Rccl t;
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);          // the copy torch (or the host program) already mapped
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return t;
        t.all_gather = reinterpret_cast<decltype(t.all_gather)>(dlsym(h, "ncclAllGather"));
        t.send = reinterpret_cast<decltype(t.send)>(dlsym(h, "ncclSend"));
        t.recv = reinterpret_cast<decltype(t.recv)>(dlsym(h, "ncclRecv"));
        t.group_start = reinterpret_cast<decltype(t.group_start)>(dlsym(h, "ncclGroupStart"));
        t.group_end = reinterpret_cast<decltype(t.group_end)>(dlsym(h, "ncclGroupEnd"));
        t.comm_count = reinterpret_cast<decltype(t.comm_count)>(dlsym(h, "ncclCommCount"));
        t.comm_rank = reinterpret_cast<decltype(t.comm_rank)>(dlsym(h, "ncclCommUserRank"));
        t.ok = t.all_gather && t.send && t.recv && t.group_start && t.group_end && t.comm_count && t.comm_rank;
        return t;
    }();
    return r;
}

}  // namespace

extern "C" int bya_allgather_kv(const void* k_local, const void* v_local, void* k_full, void* v_full, int64_t rows_local,
                                int64_t row_elems, void* comm, hipStream_t stream) {
    if (!k_local || !v_local || !k_full || !v_full || !comm || rows_local <= 0 || row_elems <= 0) return BYA_ERR_SHAPE;
    const Rccl& r = rccl();
    if (!r.ok) return BYA_ERR_UNSUPPORTED;
    const size_t n = (size_t)rows_local * (size_t)row_elems;
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    if (r.group_start() != ncclSuccess) return BYA_ERR_LAUNCH;
    const ncclResult_t a = r.all_gather(k_local, k_full, n, ncclBfloat16, c, stream);
    const ncclResult_t b = r.all_gather(v_local, v_full, n, ncclBfloat16, c, stream);
    const ncclResult_t e = r.group_end();
    return (a == ncclSuccess && b == ncclSuccess && e == ncclSuccess) ? BYA_OK : BYA_ERR_LAUNCH;
}

extern "C" int bya_alltoall_router(const void* send, void* recv, const int64_t* send_counts, const int64_t* recv_counts,
                                   int32_t world, void* comm, hipStream_t stream) {
    if (!send || !recv || !send_counts || !recv_counts || !comm || world <= 0) return BYA_ERR_SHAPE;
    const Rccl& r = rccl();
    if (!r.ok) return BYA_ERR_UNSUPPORTED;
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    int n = 0;
    if (r.comm_count(c, &n) != ncclSuccess || n != world) return BYA_ERR_SHAPE;
    for (int peer = 0; peer < world; ++peer)                     // validated BEFORE the group opens: nothing half-issued
        if (send_counts[peer] < 0 || recv_counts[peer] < 0) return BYA_ERR_SHAPE;
    const bf16_t* s = static_cast<const bf16_t*>(send);
    bf16_t* d = static_cast<bf16_t*>(recv);
    if (r.group_start() != ncclSuccess) return BYA_ERR_LAUNCH;
    bool good = true;
    int64_t so = 0, ro = 0;
    for (int peer = 0; peer < world; ++peer) {                   // element ranges are the prefix sums of the counts
        if (send_counts[peer]) good &= r.send(s + so, (size_t)send_counts[peer], ncclBfloat16, peer, c, stream) == ncclSuccess;
        if (recv_counts[peer]) good &= r.recv(d + ro, (size_t)recv_counts[peer], ncclBfloat16, peer, c, stream) == ncclSuccess;
        so += send_counts[peer];
        ro += recv_counts[peer];
    }
    const ncclResult_t e = r.group_end();
    return (good && e == ncclSuccess) ? BYA_OK : BYA_ERR_LAUNCH;
}
