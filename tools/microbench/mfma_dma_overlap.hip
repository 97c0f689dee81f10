// What does one LDS-DMA piece (buffer_load_dwordx4 ... lds: 64 lanes x 16 bytes, global -> LDS) cost a wave that is issuing MFMAs?
// One wave per SIMD (4 waves, one workgroup per CU, 256 workgroups), a loop of { 8 x v_mfma_f32_16x16x32_bf16 (16 cycles each) +
// k pieces }, the persistent GEMM's ratio being k = 1.  Variants of the piece: the shipped triple (s_mov m0, s_nop 0, buffer_load),
// without the s_nop, M0 written once per group of pieces, and a plain (register-destination) buffer_load for comparison.
// Data: a 4 MiB buffer per workgroup-range, L2-resident after the first pass.  Prints cycles per group of 8 MFMAs (bare: 128).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 raw_rsrc(const void* base, uint32_t bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

template <int K, int VAR>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, const char* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.5f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.25f + i * 1e-2f); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const i32x4 rs = raw_rsrc(src + (size_t)(blockIdx.x & 31) * (128 << 10), 128 << 10);
    const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 16384);
    uint32_t vo = lane * 16;
    i32x4 sink = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const uint32_t soff = (uint32_t)((it & 63) * 2048);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
            if (m < K) {
                if constexpr (VAR == 0)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds + m * 1024), "v"(vo), "s"(rs), "s"(soff + m * 1024) : "memory");
                else if constexpr (VAR == 1)
                    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds + m * 1024), "v"(vo), "s"(rs), "s"(soff + m * 1024) : "memory");
                else if constexpr (VAR == 2) {
                    if (m == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds) : "memory");
                    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(vo), "s"(rs), "s"(soff + m * 1024) : "memory");
                } else
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(sink) : "v"(vo), "s"(rs), "s"(soff + m * 1024) : "memory");
            }
        }
        if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)sink[0];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    s += smem[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(s != 12345.f); }
}

template <int K, int VAR>
double run(unsigned long long* d, const char* src, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<K, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((k<K, VAR>), dim3(256), dim3(256), 65536, 0, d, src, iters);
    hipLaunchKernelGGL((k<K, VAR>), dim3(256), dim3(256), 65536, 0, d, src, iters);
    unsigned long long h[2];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    return (double)h[0] / iters;
}

int main() {
    unsigned long long* d;
    char* src;
    hipMalloc(&d, 64);
    hipMalloc(&src, 8 << 20);
    hipMemset(src, 1, 8 << 20);
    const int iters = 20000;
    const double base = run<0, 0>(d, src, iters);
    printf("bare loop: %.2f ticks per 8 MFMAs (= 128 cycles)\n", base);
    const char* names[4] = {"s_mov m0 + s_nop 0 + buffer_load lds (shipped)", "s_mov m0 + buffer_load lds", "m0 once + buffer_load lds (same LDS target)",
                            "buffer_load_dwordx4 into registers"};
    double r[4][4] = {{run<1, 0>(d, src, iters), run<2, 0>(d, src, iters), run<4, 0>(d, src, iters), run<8, 0>(d, src, iters)},
                      {run<1, 1>(d, src, iters), run<2, 1>(d, src, iters), run<4, 1>(d, src, iters), run<8, 1>(d, src, iters)},
                      {run<1, 2>(d, src, iters), run<2, 2>(d, src, iters), run<4, 2>(d, src, iters), run<8, 2>(d, src, iters)},
                      {run<1, 3>(d, src, iters), run<2, 3>(d, src, iters), run<4, 3>(d, src, iters), run<8, 3>(d, src, iters)}};
    for (int v = 0; v < 4; ++v) {
        printf("%-52s", names[v]);
        const int ks[4] = {1, 2, 4, 8};
        for (int i = 0; i < 4; ++i) printf("  k=%d: %6.1f (+%4.1f per piece)", ks[i], r[v][i] / base * 128.0, (r[v][i] / base * 128.0 - 128.0) / ks[i]);
        printf("\n");
    }
    return 0;
}
