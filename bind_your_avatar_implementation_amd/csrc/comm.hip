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

// =====================================================================================================================
// P2P exchange engine: the exchanges of the sharded step as PUSH kernels over peer-mapped device memory.
//
// xGMI is point to point and a GPU can store straight into a peer's HBM: an exchange does not need a collective library,
// it needs (1) the peers' receive buffers mapped into this process (hipIpc handles, traded once at set-up by the host
// module), (2) a kernel that copies this rank's outgoing pieces to their final addresses on the peers, and (3) a flag per
// (channel, source rank) on every peer.  What that buys over RCCL's grouped send/recv on this stack:
//   * a whole exchange -- any scatter/gather list, e.g. the 3 x W column blocks of the packed q|k|v projection to their
//     places in W peers' q / k / v buffers -- is ONE ordinary kernel launch (~5 us) instead of one collective per tensor
//     (~20 us each, 360 of them per rank-step in round 3);
//   * ordinary kernels can be captured: the sharded step replays as a hipGraph (an RCCL collective under capture never
//     returns on this stack, profiles/r3_rccl_graph_probe.txt).  Sequence numbers therefore live in DEVICE memory (the
//     push kernel bumps its channel's send counter, the wait kernel its expect counter): a replay advances them itself.
// Ordering: every workgroup of the push kernel makes its stores visible system-wide (__threadfence_system) before it
// counts itself done; the last one publishes the new sequence number to every peer's flag (system-scope release store).
// The receiver runs bya_p2p_wait -- one wave spinning on its LOCAL flags -- as the next kernel on its stream; the
// consumers are later kernels on that stream, whose dispatch acquires at system scope (stale L2 lines of the receive
// buffer are dropped).  A peer may only overwrite a receive buffer after this rank has consumed it: the step's own data
// dependencies guarantee that for every exchange the engine issues (multi-GPU section, lists them).
// Bounded waits (~1 s) count a time-out in the channel's control words instead of hanging the GPU.
namespace {

constexpr int P2P_CHUNK = 64 * 1024;                 // bytes per workgroup iteration
constexpr int P2P_CTRL_WORDS = 64;                   // per channel: [0..31] flags by source rank, [32] sent, [33] expected,
constexpr int P2P_SENT = 32, P2P_EXPECT = 33, P2P_DONE = 34, P2P_TIMEOUTS = 35;      // [34] workgroups done, [35] time-outs

__global__ __launch_bounds__(256) void p2p_push_kernel(const bya_p2p_copy* __restrict__ copies, int n_copies, long long total_chunks,
                                                        unsigned* const* __restrict__ peer_ctrl, int world, int rank,
                                                        unsigned* __restrict__ ctrl) {
    const int tid = threadIdx.x;
    for (long long c = blockIdx.x; c < total_chunks; c += gridDim.x) {
        int i = 0;
        while (i + 1 < n_copies && copies[i + 1].chunk0 <= c) ++i;               // <= 100 entries: a linear scan
        const long long off = (c - copies[i].chunk0) * (long long)P2P_CHUNK;
        const long long left = copies[i].bytes - off;
        const int n = (int)(left < P2P_CHUNK ? left : P2P_CHUNK);
        const char* src = static_cast<const char*>(copies[i].src) + off;
        char* dst = static_cast<char*>(copies[i].dst) + off;
        if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
            const int n16 = n & ~15;
            for (int b = tid * 16; b < n16; b += 256 * 16)
                *reinterpret_cast<u32x4*>(dst + b) = *reinterpret_cast<const u32x4*>(src + b);
            for (int b = n16 + tid * 2; b < n; b += 256 * 2)            // (a piece ends on a bf16 element, not on 16 bytes)
                *reinterpret_cast<uint16_t*>(dst + b) = *reinterpret_cast<const uint16_t*>(src + b);
        } else {                                                        // small odd-sized pieces (the router's logits)
            for (int b = tid * 2; b < n; b += 256 * 2)
                *reinterpret_cast<uint16_t*>(dst + b) = *reinterpret_cast<const uint16_t*>(src + b);
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const unsigned done = __hip_atomic_fetch_add(ctrl + P2P_DONE, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(ctrl + P2P_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned seq = ctrl[P2P_SENT] + 1u;
            ctrl[P2P_SENT] = seq;
            __threadfence_system();
            for (int p = 0; p < world; ++p)
                __hip_atomic_store(peer_ctrl[p] + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(64) void p2p_wait_kernel(unsigned* __restrict__ ctrl, int world) {
    const int lane = threadIdx.x;
    const unsigned expect = ctrl[P2P_EXPECT] + 1u;
    if (lane < world) {
        int spins = 0;
        // (sequence numbers wrap after 4 G exchanges: compare by signed distance)
        while ((int)(__hip_atomic_load(ctrl + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - expect) < 0) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1 << 21)) {
                __hip_atomic_fetch_add(ctrl + P2P_TIMEOUTS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __builtin_amdgcn_s_barrier();
    if (lane == 0) ctrl[P2P_EXPECT] = expect;
}

}  // namespace

extern "C" int bya_p2p_push(const bya_p2p_copy* copies_dev, int32_t n_copies, int64_t total_chunks, void* const* peer_ctrl_dev,
                            int32_t world, int32_t rank, void* ctrl, hipStream_t stream) {
    if (!copies_dev || !peer_ctrl_dev || !ctrl || n_copies <= 0 || total_chunks <= 0) return BYA_ERR_SHAPE;
    if (world <= 0 || world > 32 || rank < 0 || rank >= world) return BYA_ERR_SHAPE;
    if (((uintptr_t)copies_dev | (uintptr_t)peer_ctrl_dev) & 7 || ((uintptr_t)ctrl & 3)) return BYA_ERR_ALIGN;
    // enough workgroups to keep every xGMI link and the local HBM busy, few enough to leave the CUs to the compute stream
    const long long want = total_chunks < 64 ? total_chunks : 64;
    BYA_LAUNCH(p2p_push_kernel, dim3((unsigned)want), dim3(256), 0, stream, copies_dev, n_copies, (long long)total_chunks,
               reinterpret_cast<unsigned* const*>(peer_ctrl_dev), world, rank, static_cast<unsigned*>(ctrl));
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

extern "C" int bya_p2p_wait(void* ctrl, int32_t world, hipStream_t stream) {
    if (!ctrl || world <= 0 || world > 32) return BYA_ERR_SHAPE;
    BYA_LAUNCH(p2p_wait_kernel, dim3(1), dim3(64), 0, stream, static_cast<unsigned*>(ctrl), world);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
