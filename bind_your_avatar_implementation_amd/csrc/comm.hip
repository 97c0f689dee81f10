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
#include <string.h>
#include "options.h"

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
//     returns on this stack, profiles/history/r3_rccl_graph_probe.txt).  Sequence numbers therefore live in DEVICE memory (the
//     push kernel bumps its channel's send counter, the wait kernel its expect counter): a replay advances them itself.
// Ordering: every workgroup of the push kernel makes its stores visible system-wide (__threadfence_system) before it
// counts itself done; the last one publishes the new sequence number to every peer's flag (system-scope release store).
// The receiver waits on its LOCAL flags (the control block is fine-grained memory when the host module could get it:
// bya_p2p_alloc) with P2P_WAIT_GROUPS workgroups, which the dispatcher deals over the XCDs: each polls with system-scope
// loads and ends with a system-scope acquire fence, so every XCD's L2 has dropped what it may hold of the receive buffers
// before a consumer -- a later kernel on the stream -- reads them (round 4 waited with ONE wave: one XCD).  A peer may only
// overwrite a receive buffer after this rank has consumed it: the step's own data dependencies guarantee that for every
// exchange the engine issues (multi-GPU section, lists them).
// A wait is bounded by WALL time (s_memrealtime; the limit is an argument of the launch, default 30 s): when it gives up it
// counts the event in the channel's word 35 AND in word 37 of the group's first control block, both STICKY.  A wait that
// finds the group's word set does not poll at all (round 5 bounded every wait separately: with ~280 exchanges per
// rank-step a dead peer stalled each STEP for hours before anything was reported); it counts itself as timed out and
// returns, so the remaining launches of the step -- and of the clip -- run through at full speed on garbage.
// bya_p2p_poison -- the last launch of a sharded step -- overwrites the step's output with NaN if any channel of the group
// carries a time-out, so a result made from a stale buffer cannot be used, and the host raises at the next check.
namespace {

constexpr int P2P_CHUNK = 64 * 1024;                 // bytes per workgroup iteration
constexpr int P2P_CTRL_WORDS = 64;                   // per channel: [0..31] flags by source rank, [32] sent, [33] expected,
constexpr int P2P_SENT = 32, P2P_EXPECT = 33, P2P_DONE = 34, P2P_TIMEOUTS = 35, P2P_WAITED = 36;   // [34] push workgroups done,
                                                                                                   // [35] time-outs, [36] wait workgroups done
constexpr int P2P_GROUP_TIMEOUTS = 37;               // in the group's FIRST control block only: time-outs of any channel
constexpr int P2P_MISMATCHES = 38;                   // [38] written by the host module: steps whose exchange checksum differed between the ranks
constexpr int P2P_WAIT_GROUPS = 16;                  // workgroups of a wait: two per XCD under round-robin dispatch
constexpr int P2P_THREADS = 256;                    // lanes of a push workgroup: 16 waves x 4 loads of 16 bytes in flight per lane (a 4-wave
                                                     // workgroup per CU kept 16 KB in flight: 11 GB/s per CU at HBM latency)
constexpr int P2P_MAX_GROUPS = 128;                  // workgroups of a push (BYA_P2P_GROUPS: probe switch, read per call)

constexpr long long P2P_DEFAULT_LIMIT_TICKS = 30ll * 100000000ll;        // s_memrealtime ticks (100 MHz): 30 s

// the copy loop of one workgroup.  A table entry is a 2-D piece: `rows` rows of `row_bytes` bytes, `src_pitch` / `dst_pitch`
// bytes apart (a contiguous piece is one row).  Its chunks: a row longer than P2P_CHUNK is cut into P2P_CHUNK-byte chunks
// (chunks_per_row > 1, one row per chunk); shorter rows are grouped, rows_per_chunk = P2P_CHUNK / row_bytes of them per chunk.
// Order: the host's tables list the pieces peer by peer, so chunk index ~ destination.  Walking the chunks in index order would
// have every workgroup store to the SAME peer at any moment -- one xGMI link busy, six idle (the links are point to point).
// The walk therefore goes down the columns of a [world x ceil(chunks / world)] arrangement of the chunk indices: consecutive
// iterations -- what the workgroups of the grid work on at the same time -- are about one peer's share apart, and rank r
// starts with the share of peer r + 1, so that the ranks do not all open on peer 0's ingress either.
__device__ __forceinline__ void p2p_copy_chunks(const bya_p2p_copy* __restrict__ copies, int n_copies, long long total_chunks, int world,
                                                int rank) {
    const int tid = threadIdx.x;
    const long long share = (total_chunks + world - 1) / world, walk = share * world;
    for (long long it = blockIdx.x; it < walk; it += gridDim.x) {
        const long long c = ((it + rank + 1) % world) * share + it / world;
        if (c >= total_chunks) continue;
        int i = 0;
        while (i + 1 < n_copies && copies[i + 1].chunk0 <= c) ++i;               // <= ~100 entries: a linear scan
        const bya_p2p_copy e = copies[i];
        if (e.row_bytes <= 0 || e.rows <= 0) continue;                           // (the place-holder entry of a rank with nothing to send)
        const long long ci = c - e.chunk0;
        long long row0, col0;
        int nrows, n;                                                            // rows in this chunk, bytes per row of it
        if (e.row_bytes > P2P_CHUNK) {
            const long long cpr = (e.row_bytes + P2P_CHUNK - 1) / P2P_CHUNK;
            row0 = ci / cpr;
            col0 = (ci - row0 * cpr) * (long long)P2P_CHUNK;
            nrows = 1;
            const long long left = e.row_bytes - col0;
            n = (int)(left < P2P_CHUNK ? left : P2P_CHUNK);
        } else {
            const long long rpc = P2P_CHUNK / e.row_bytes;
            row0 = ci * rpc;
            col0 = 0;
            const long long left = e.rows - row0;
            nrows = (int)(left < rpc ? left : rpc);
            n = (int)e.row_bytes;
        }
        const char* src = static_cast<const char*>(e.src) + row0 * e.src_pitch + col0;
        char* dst = static_cast<char*>(e.dst) + row0 * e.dst_pitch + col0;
        const bool wide = ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) && (nrows == 1 || ((e.src_pitch | e.dst_pitch | n) & 15) == 0);
        if (wide) {
            // four independent 16-byte loads in flight per lane before the first store (one load -> one store per iteration
            // moved 0.5 TB/s against LOCAL memory: latency-bound long before any link is)
            const int n16 = n & ~15, per_row = n16 >> 4, total = per_row * nrows;
            const float inv_row = 1.0f / (float)per_row;
            auto off = [&](int t, long long& so, long long& d_o) {
                if (nrows == 1) { so = d_o = (long long)t << 4; return; }
                int r = (int)((float)t * inv_row);                       // t < 4096: exact up to the correction below
                r -= (r * per_row > t);
                r += ((r + 1) * per_row <= t);
                const long long b = (long long)(t - r * per_row) << 4;
                so = (long long)r * e.src_pitch + b;
                d_o = (long long)r * e.dst_pitch + b;
            };
            int t = tid;
            for (; t + 3 * P2P_THREADS < total; t += 4 * P2P_THREADS) {
                long long s0, s1, s2, s3, d0, d1, d2, d3;
                off(t, s0, d0); off(t + P2P_THREADS, s1, d1); off(t + 2 * P2P_THREADS, s2, d2); off(t + 3 * P2P_THREADS, s3, d3);
                const u32x4 a = *reinterpret_cast<const u32x4*>(src + s0), b = *reinterpret_cast<const u32x4*>(src + s1);
                const u32x4 c2 = *reinterpret_cast<const u32x4*>(src + s2), d = *reinterpret_cast<const u32x4*>(src + s3);
                __builtin_nontemporal_store(a, reinterpret_cast<u32x4*>(dst + d0));
                __builtin_nontemporal_store(b, reinterpret_cast<u32x4*>(dst + d1));
                __builtin_nontemporal_store(c2, reinterpret_cast<u32x4*>(dst + d2));
                __builtin_nontemporal_store(d, reinterpret_cast<u32x4*>(dst + d3));
            }
            for (; t < total; t += P2P_THREADS) {
                long long s0, d0;
                off(t, s0, d0);
                *reinterpret_cast<u32x4*>(dst + d0) = *reinterpret_cast<const u32x4*>(src + s0);
            }
            for (int b = n16 + tid * 2; b < n; b += P2P_THREADS * 2)            // (nrows == 1 here: a piece ends on a bf16 element, not on 16 bytes)
                *reinterpret_cast<uint16_t*>(dst + b) = *reinterpret_cast<const uint16_t*>(src + b);
        } else {                                                        // small odd-sized pieces (the router's logits)
            const int per_row = n >> 1, total = per_row * nrows;
            for (int t = tid; t < total; t += P2P_THREADS) {
                const int r = nrows == 1 ? 0 : t / per_row, b = (t - r * per_row) << 1;
                *reinterpret_cast<uint16_t*>(dst + (long long)r * e.dst_pitch + b) = *reinterpret_cast<const uint16_t*>(src + (long long)r * e.src_pitch + b);
            }
        }
    }
}

// one workgroup's wait: lanes < world poll the local flags until all carry `expect` (signed distance: sequence numbers wrap
// after 4 G exchanges) or the wall-clock limit passes; then a system-scope acquire; the workgroup that finishes last
// advances the channel's expect counter.  `groups` = workgroups taking part.
__device__ __forceinline__ void p2p_wait_flags(unsigned* __restrict__ ctrl, unsigned* __restrict__ group_ctrl, int world, unsigned expect,
                                               long long limit_ticks, unsigned groups) {
    const int lane = threadIdx.x;
    if (lane < world) {
        unsigned* const gt = group_ctrl ? group_ctrl + P2P_GROUP_TIMEOUTS : ctrl + P2P_TIMEOUTS;     // (no group block: the channel's own word)
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        // a wait of this group has already given up: the peers are late or gone, this one would only add its full limit
        bool dead = __hip_atomic_load(gt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        while (!dead && (int)(__hip_atomic_load(ctrl + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - expect) < 0) {
            __builtin_amdgcn_s_sleep(16);
            dead = __hip_atomic_load(gt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                   (long long)(__builtin_amdgcn_s_memrealtime() - t0) > limit_ticks;
        }
        if (dead) {
            __hip_atomic_fetch_add(ctrl + P2P_TIMEOUTS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (group_ctrl) __hip_atomic_fetch_add(group_ctrl + P2P_GROUP_TIMEOUTS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // system scope: this CU's L1 and this XCD's L2 drop stale lines
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (ONE wave calls this: program order is all the ordering it needs)
    if (lane == 0) {
        const unsigned done = __hip_atomic_fetch_add(ctrl + P2P_WAITED, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == groups - 1) {
            __hip_atomic_store(ctrl + P2P_WAITED, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctrl + P2P_EXPECT, expect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// WAIT: the same workgroups then wait for the peers' pushes of this exchange (push + wait as one launch; every workgroup has
// finished its own copies before it starts to poll, and publishing never depends on a poller, so nothing can wait in a circle)
template <bool WAIT>
__global__ __launch_bounds__(P2P_THREADS) void p2p_push_kernel(const bya_p2p_copy* __restrict__ copies, int n_copies, long long total_chunks,
                                                        unsigned* const* __restrict__ peer_ctrl, int world, int rank,
                                                        unsigned* __restrict__ ctrl, unsigned* __restrict__ group_ctrl, long long limit_ticks) {
    const int tid = threadIdx.x;
    unsigned expect = 0;
    if (WAIT) expect = __hip_atomic_load(ctrl + P2P_EXPECT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;   // (before anybody can advance it)
    p2p_copy_chunks(copies, n_copies, total_chunks, world, rank);
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const unsigned done = __hip_atomic_fetch_add(ctrl + P2P_DONE, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(ctrl + P2P_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned seq = __hip_atomic_load(ctrl + P2P_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            __hip_atomic_store(ctrl + P2P_SENT, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            for (int p = 0; p < world; ++p)
                __hip_atomic_store(peer_ctrl[p] + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (WAIT && blockIdx.x < P2P_WAIT_GROUPS) {
        // the first P2P_WAIT_GROUPS workgroups (two per XCD under round-robin dispatch, like the wait kernel's) stay to wait;
        // with all 128 polling the exchange of an 8-rank step took 8 us longer than push + a separate wait launch
        __syncthreads();
        if (tid < 64) p2p_wait_flags(ctrl, group_ctrl, world, expect, limit_ticks, gridDim.x < P2P_WAIT_GROUPS ? gridDim.x : P2P_WAIT_GROUPS);
    }
}

__global__ __launch_bounds__(64) void p2p_wait_kernel(unsigned* __restrict__ ctrl, unsigned* __restrict__ group_ctrl, int world,
                                                      long long limit_ticks) {
    const unsigned expect = __hip_atomic_load(ctrl + P2P_EXPECT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    p2p_wait_flags(ctrl, group_ctrl, world, expect, limit_ticks, gridDim.x);
}

// out[0 .. n) := NaN (bf16) if any of the n_channels control blocks starting at ctrl_base carries a time-out (word 35) or a
// checksum mismatch (word 38)
__global__ __launch_bounds__(256) void p2p_poison_kernel(const unsigned* __restrict__ ctrl_base, int n_channels, uint16_t* __restrict__ out,
                                                          long long n) {
    int bad = 0;
    for (int c = threadIdx.x; c < n_channels; c += 256)
        bad |= __hip_atomic_load(ctrl_base + (size_t)c * P2P_CTRL_WORDS + P2P_TIMEOUTS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
               __hip_atomic_load(ctrl_base + (size_t)c * P2P_CTRL_WORDS + P2P_MISMATCHES, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (!__syncthreads_or(bad)) return;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = 0x7fc0u;
}

inline long long p2p_limit_ticks(int64_t ms) { return ms > 0 ? (long long)ms * 100000ll : P2P_DEFAULT_LIMIT_TICKS; }

inline long long p2p_max_groups() {
    const int v = bya_opt(BYA_OPT_P2P_GROUPS);
    return v >= P2P_WAIT_GROUPS && v <= 1024 ? v : P2P_MAX_GROUPS;
}

int p2p_check(const void* copies_dev, const void* peer_ctrl_dev, const void* ctrl, int32_t n_copies, int64_t total_chunks, int32_t world,
              int32_t rank) {
    if (!copies_dev || !peer_ctrl_dev || !ctrl || n_copies <= 0 || total_chunks <= 0) return BYA_ERR_SHAPE;
    if (world <= 0 || world > 32 || rank < 0 || rank >= world) return BYA_ERR_SHAPE;
    if (((uintptr_t)copies_dev | (uintptr_t)peer_ctrl_dev) & 7 || ((uintptr_t)ctrl & 3)) return BYA_ERR_ALIGN;
    return BYA_OK;
}

}  // namespace

extern "C" int bya_p2p_push(const bya_p2p_copy* copies_dev, int32_t n_copies, int64_t total_chunks, void* const* peer_ctrl_dev,
                            int32_t world, int32_t rank, void* ctrl, hipStream_t stream) {
    const int rc = p2p_check(copies_dev, peer_ctrl_dev, ctrl, n_copies, total_chunks, world, rank);
    if (rc != BYA_OK) return rc;
    // enough workgroups to keep every xGMI link and the local HBM busy, few enough to leave CUs to the compute stream
    const long long cap = p2p_max_groups(), want = total_chunks < cap ? total_chunks : cap;
    BYA_LAUNCH(p2p_push_kernel<false>, dim3((unsigned)want), dim3(P2P_THREADS), 0, stream, copies_dev, n_copies, (long long)total_chunks,
               reinterpret_cast<unsigned* const*>(peer_ctrl_dev), world, rank, static_cast<unsigned*>(ctrl), static_cast<unsigned*>(nullptr), 0ll);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

extern "C" int bya_p2p_wait(void* ctrl, int32_t world, void* group_ctrl, int64_t wait_limit_ms, hipStream_t stream) {
    if (!ctrl || world <= 0 || world > 32) return BYA_ERR_SHAPE;
    if (((uintptr_t)ctrl | (uintptr_t)group_ctrl) & 3) return BYA_ERR_ALIGN;
    BYA_LAUNCH(p2p_wait_kernel, dim3(P2P_WAIT_GROUPS), dim3(64), 0, stream, static_cast<unsigned*>(ctrl), static_cast<unsigned*>(group_ctrl),
               world, p2p_limit_ticks(wait_limit_ms));
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

extern "C" int bya_p2p_exchange(const bya_p2p_copy* copies_dev, int32_t n_copies, int64_t total_chunks, void* const* peer_ctrl_dev,
                                int32_t world, int32_t rank, void* ctrl, void* group_ctrl, int64_t wait_limit_ms, hipStream_t stream) {
    const int rc = p2p_check(copies_dev, peer_ctrl_dev, ctrl, n_copies, total_chunks, world, rank);
    if (rc != BYA_OK) return rc;
    if ((uintptr_t)group_ctrl & 3) return BYA_ERR_ALIGN;
    // at least P2P_WAIT_GROUPS workgroups, so that the acquire at the end of the wait reaches every XCD
    const long long cap = p2p_max_groups();
    long long want = total_chunks < cap ? total_chunks : cap;
    if (want < P2P_WAIT_GROUPS) want = P2P_WAIT_GROUPS;
    BYA_LAUNCH(p2p_push_kernel<true>, dim3((unsigned)want), dim3(P2P_THREADS), 0, stream, copies_dev, n_copies, (long long)total_chunks,
               reinterpret_cast<unsigned* const*>(peer_ctrl_dev), world, rank, static_cast<unsigned*>(ctrl), static_cast<unsigned*>(group_ctrl),
               p2p_limit_ticks(wait_limit_ms));
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

extern "C" int bya_p2p_poison(const void* ctrl_base, int32_t n_channels, void* out, int64_t n_elems, hipStream_t stream) {
    if (!ctrl_base || !out || n_channels <= 0 || n_elems <= 0) return BYA_ERR_SHAPE;
    if (((uintptr_t)ctrl_base & 3) || ((uintptr_t)out & 1)) return BYA_ERR_ALIGN;
    const long long blocks = (n_elems + 255) / 256;
    BYA_LAUNCH(p2p_poison_kernel, dim3((unsigned)(blocks < 128 ? blocks : 128)), dim3(256), 0, stream,
               static_cast<const unsigned*>(ctrl_base), n_channels, static_cast<uint16_t*>(out), (long long)n_elems);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

// ---- memory a peer may store into or that a running kernel polls: the three kinds of device memory, and their hipIpc handles --------
// kind 0: coarse-grained (hipMalloc: cached in the L2s; coherent at kernel boundaries), 1: fine-grained, 2: uncached.
// Set-up calls, not part of a step: they are the only entry points of the library that allocate.
extern "C" int bya_p2p_alloc(int64_t bytes, int32_t kind, void** out) {
    if (!out || bytes <= 0 || kind < 0 || kind > 2) return BYA_ERR_SHAPE;
    void* p = nullptr;
    const unsigned flags = kind == 0 ? hipDeviceMallocDefault : (kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
    if (hipExtMallocWithFlags(&p, (size_t)bytes, flags) != hipSuccess || !p) { (void)hipGetLastError(); return BYA_ERR_UNSUPPORTED; }
    if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return BYA_ERR_LAUNCH; }
    *out = p;
    return BYA_OK;
}

extern "C" int bya_p2p_free(void* ptr) {
    if (!ptr) return BYA_ERR_SHAPE;
    return hipFree(ptr) == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}

extern "C" int bya_p2p_ipc_export(void* ptr, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle travels as 64 bytes");
    if (!ptr || !handle64) return BYA_ERR_SHAPE;
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, ptr) != hipSuccess) { (void)hipGetLastError(); return BYA_ERR_UNSUPPORTED; }
    memcpy(handle64, &h, 64);
    return BYA_OK;
}

extern "C" int bya_p2p_ipc_import(const void* handle64, void** out) {
    if (!handle64 || !out) return BYA_ERR_SHAPE;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p) { (void)hipGetLastError(); return BYA_ERR_UNSUPPORTED; }
    *out = p;
    return BYA_OK;
}

extern "C" int bya_p2p_ipc_release(void* ptr) {
    if (!ptr) return BYA_ERR_SHAPE;
    return hipIpcCloseMemHandle(ptr) == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
