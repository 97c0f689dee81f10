// Board calibration (round 6): a loop of nothing but the GEMM's matrix instruction, v_mfma_f32_16x16x32_bf16, on operands the
// caller supplies (gaussian bf16: the board clocks down until it fits its power limit, so the sustained rate is what THIS
// board gives the instruction on real data; on zeros every board runs at the full clock).  One wave per SIMD, one workgroup per
// CU, a 128 x 128 wave tile per sweep (8 A and 8 B fragments in VGPRs, 64 accumulators in AGPRs), no loads, no LDS, no
// barrier inside the loop.  bench.py runs it before the timed region and prints `board_calibration_tflops` next to its
// roofline fractions: the pool's boxes differ by +-3 % in what they sustain, and a line that carries
// the board's own ceiling can be compared across boxes.  (tools/microbench/mfma_shape_power.hip is the stand-alone form.)
#include "bya_common.h"
#include "../../include/bya.h"

namespace {

__global__ __launch_bounds__(256, 1) void mfma_calibration_kernel(const bf16x8* __restrict__ src, float* __restrict__ sink, int iters) {
    bf16x8 a[8], b[8];
    const int base = (blockIdx.x * 256 + threadIdx.x) * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = src[base + i]; b[i] = src[base + 8 + i]; }
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
    if (s == 12345.678f) sink[0] = s;          // (keeps the loop alive; never true in practice)
}

}  // namespace

extern "C" int bya_mfma_calibration(const void* operands, int64_t operand_bytes, void* sink, int32_t iters, hipStream_t stream) {
    if (!operands || !sink || iters <= 0) return BYA_ERR_SHAPE;
    if (operand_bytes < BYA_CALIBRATION_OPERAND_BYTES) return BYA_ERR_SHAPE;
    if (((uintptr_t)operands & 15) || ((uintptr_t)sink & 3)) return BYA_ERR_ALIGN;
    BYA_LAUNCH(mfma_calibration_kernel, dim3(256), dim3(256), 0, stream, static_cast<const bf16x8*>(operands), static_cast<float*>(sink), iters);
    return hipGetLastError() == hipSuccess ? BYA_OK : BYA_ERR_LAUNCH;
}
