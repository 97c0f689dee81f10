// Self-contained reproducer kernels for the multi-process corruption seen in round 1:
//   * lds_hog      -- a "disturber": persistent 512-thread blocks that own `lds_bytes` of dynamic LDS and keep reading
//                     and writing it for `iters` passes (no global traffic besides one sink word);
//   * rmw_victim   -- a "victim": every 8-lane group reads 8 x 16 bytes, reduces across the group with DPP/shuffles,
//                     and writes 16 bytes per lane either IN PLACE or to a second buffer.
// Built into tools/timeslice/liblds_hog.so by tools/timeslice/repro.py; nothing in the product links it.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(512) void lds_hog(int* sink, int iters, int lds_words) {
    extern __shared__ __attribute__((aligned(16))) int s[];
    for (int i = threadIdx.x; i < lds_words; i += 512) s[i] = i ^ (int)blockIdx.x;
    __syncthreads();
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < lds_words; i += 512) { acc += s[i]; s[i] = acc ^ it; }
        __syncthreads();
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

__global__ __launch_bounds__(256) void rmw_victim(const float* __restrict__ in, float* __restrict__ out, long long nvec) {
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;      // one float4 per thread
    if (v >= nvec) return;
    float4 x = reinterpret_cast<const float4*>(in)[v];
    float s = x.x + x.y + x.z + x.w;
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);       // 8-lane group sum
    const float m = s * (1.0f / 32.0f);
    x.x = (x.x - m) * 1.5f + 0.25f; x.y = (x.y - m) * 1.5f + 0.25f;
    x.z = (x.z - m) * 1.5f + 0.25f; x.w = (x.w - m) * 1.5f + 0.25f;
    reinterpret_cast<float4*>(out)[v] = x;
}

// "broadcast-table" victim: the access pattern of the q/k-norm + RoPE kernel reduced to its loads.  A wave handles 8
// consecutive (row, head) pairs, 8 lanes each; all pairs of one row read the SAME 256-byte row of a table (8 lanes x 32
// bytes), i.e. every table address is requested by 8 lanes spread over the whole wave.  out = x * table (out of place).
__global__ __launch_bounds__(256) void bcast_victim(const float* __restrict__ x, const float* __restrict__ table,
                                                    float* __restrict__ out, long long npairs, int heads, int sc1) {
    const int lane = threadIdx.x & 63;
    const long long pair = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (lane >> 3);
    if (pair >= npairs) return;
    const long long row = pair / heads;
    const int d0 = (lane & 7) * 8;
    const float* t = table + row * 64 + d0;
    float4 a, b;
    if (sc1) {
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, 0x7fffffff, 0x00020000);
        const unsigned off = (unsigned)((row * 64 + d0) * 4);
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 ra = __builtin_amdgcn_raw_buffer_load_b128(rt, off, 0, 16), rb = __builtin_amdgcn_raw_buffer_load_b128(rt, off + 16, 0, 16);
        a = __builtin_bit_cast(float4, ra); b = __builtin_bit_cast(float4, rb);
    } else {
        a = reinterpret_cast<const float4*>(t)[0]; b = reinterpret_cast<const float4*>(t)[1];
    }
    const float4 x0 = reinterpret_cast<const float4*>(x + pair * 64 + d0)[0], x1 = reinterpret_cast<const float4*>(x + pair * 64 + d0)[1];
    float4 o0 = {x0.x * a.x, x0.y * a.y, x0.z * a.z, x0.w * a.w}, o1 = {x1.x * b.x, x1.y * b.y, x1.z * b.z, x1.w * b.w};
    reinterpret_cast<float4*>(out + pair * 64 + d0)[0] = o0;
    reinterpret_cast<float4*>(out + pair * 64 + d0)[1] = o1;
}

extern "C" int bcast_launch(const float* x, const float* table, float* out, long long npairs, int heads, int sc1,
                            hipStream_t stream) {
    hipLaunchKernelGGL(bcast_victim, dim3((unsigned)((npairs + 31) / 32)), dim3(256), 0, stream, x, table, out, npairs, heads, sc1);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int hog_launch(int* sink, int blocks, int iters, int lds_bytes, hipStream_t stream) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_hog), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds_bytes) != hipSuccess) return -1;
    hipLaunchKernelGGL(lds_hog, dim3(blocks), dim3(512), lds_bytes, stream, sink, iters, lds_bytes / 4);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int victim_launch(const float* in, float* out, long long nvec, hipStream_t stream) {
    hipLaunchKernelGGL(rmw_victim, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, stream, in, out, nvec);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
