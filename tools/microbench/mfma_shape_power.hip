// Which bf16 MFMA shape gives more FLOP/s on a power-limited MI355X?  v_mfma_f32_16x16x32_bf16 (what gemm256p_kernel
// issues) reads 2 x 4 operand VGPRs per 16384 FLOP, v_mfma_f32_32x32x16_bf16 (what attn_joint_w4_kernel issues) the same
// 2 x 4 per 32768 FLOP: half the register-file reads per FLOP at the same FLOP per cycle.  On all-zero operands both run at the
// full clock; on gaussian operands the board clocks down until it fits its power limit, so the sustained rate measures energy
// per FLOP.  One wave per SIMD, one workgroup per CU, 256 accumulator AGPRs (a 128 x 128 wave tile per k-step, like the GEMM),
// nothing but MFMAs in the loop: 8 A and 8 B fragments held in VGPRs.
//   hipcc --offload-arch=gfx950 -O3 mfma_shape_power.hip -o mfma_shape_power && ./mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// SHAPE 0: 64 blocks of 16x16 (A_i x B_j, i, j < 8), k = 32 per sweep;  SHAPE 1: 16 blocks of 32x32 (i, j < 4), two k = 16
// halves per sweep (fragments 0..3 and 4..7): both sweeps are 128 x 128 x 32 = 1 MFLOP per wave
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* __restrict__ src, float* out, int iters) {
    bf16x8 a[8], b[8];
    const int lane = threadIdx.x, base = (blockIdx.x * 256 + lane) * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = src[base + i]; b[i] = src[base + 8 + i]; }
    if constexpr (SHAPE == 0) {
        f32x4 acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (s == 12345.678f) out[0] = s;
    } else {
        f32x16 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[4 * h + i]), "v"(b[4 * h + j]));
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        if (s == 12345.678f) out[0] = s;
    }
}

static unsigned short bf16_of(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main() {
    const int grid = 256, iters = 600000;
    const size_t n = (size_t)grid * 256 * 16 * 8;
    std::vector<unsigned short> h(n);
    unsigned long long st = 88172645463325252ull;
    auto uni = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    for (size_t i = 0; i < n; i += 2) {
        const double r = sqrt(-2.0 * log(uni() + 1e-300)), t = 6.283185307179586 * uni();
        h[i] = bf16_of((float)(r * cos(t)));
        h[i + 1] = bf16_of((float)(r * sin(t)));
    }
    bf16x8 *gauss, *zeros;
    float* out;
    hipMalloc(&gauss, n * 2); hipMalloc(&zeros, n * 2); hipMalloc(&out, 4);
    hipMemcpy(gauss, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemset(zeros, 0, n * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = (double)grid * 4 * iters * 2.0 * 128 * 128 * 32;
    printf("{");
    for (int rep = 0; rep < 3; ++rep)
        for (int data = 0; data < 2; ++data)
            for (int shape = 0; shape < 2; ++shape) {
                const bf16x8* src = data ? gauss : zeros;
                for (int w = 0; w < 2; ++w) {          // the second launch is the timed one (the first brings the board to its steady clock)
                    hipEventRecord(e0);
                    if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, src, out, iters);
                    else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, src, out, iters);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                }
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                printf("%s\"rep%d %s %s\": {\"ms\": %.2f, \"tflops\": %.0f}", (rep || data || shape) ? ", " : "", rep,
                       data ? "gaussian" : "zeros", shape ? "32x32x16" : "16x16x32", ms, flop / ms * 1e-9);
            }
    printf("}\n");
    return 0;
}
