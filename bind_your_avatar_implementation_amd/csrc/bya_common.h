// Shared device helpers for the gfx950 (CDNA4, MI355X) kernels of the Bind-Your-Avatar
// denoise-step engine.  Wave = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define BYA_OK 0
#define BYA_ERR_SHAPE (-1)
#define BYA_ERR_ALIGN (-2)
#define BYA_ERR_LAUNCH (-3)
#define BYA_ERR_UNSUPPORTED (-4)

// hipGetLastError() is sticky across unrelated runtime calls of the host process (torch's own event queries
// etc.); clear it before the launch so the status we return describes THIS launch only.
#define BYA_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: remember, per kernel, on which devices it has
// been raised (one bit per device; racing first calls both set it, which is harmless).
#include <atomic>
inline int bya_allow_big_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return BYA_ERR_LAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return BYA_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return BYA_ERR_LAUNCH;
    done.fetch_or(bit, std::memory_order_release);
    return BYA_OK;
}

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// 8 bf16 (one 16-byte vector) <-> 8 floats
__device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bflo(v[i]); f[2 * i + 1] = bfhi(v[i]); }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// GELU(tanh):  0.5 x (1 + tanh(u)) = x * sigmoid(2u),  u = k0 (x + k1 x^3).  One v_exp_f32 and one v_rcp_f32 (1 ulp
// each, far below the bf16 output step) instead of expf + an IEEE division: the FF1 epilogue evaluates it 218 M times
// per launch (128 values per lane of a 256x256 tile), where the division sequence alone was ~10 instructions a value.
__device__ __forceinline__ float gelu_tanh(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u2 = x * fmaf(x * x, 2.0f * k0 * k1, 2.0f * k0);                  // 2u
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * u2);            // exp(-2u): 0 .. inf, never NaN for finite x
    return x * __builtin_amdgcn_rcpf(1.0f + e);                                   // x / (1 + exp(-2u)); e = inf -> 0
}
// the round-1 form (expf + IEEE division), kept for A/B in tools/gemm_probe.py as activation code 6
__device__ __forceinline__ float gelu_tanh_ieee(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float u = k0 * (x + k1 * x * x * x);
    float e = __expf(2.0f * u);
    float t = 1.0f - 2.0f / (e + 1.0f);
    return 0.5f * x * (1.0f + t);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// Bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (round-robin dispatch),
// so give every XCD one contiguous chunk of the logical tile order (speed only, never correctness).
// fp32 -> OCP e4m3fn, round to nearest even, |v| <= 448, in integer arithmetic: byte for byte torch's float8_e4m3fn
// converter (checked on 2.5 M values incl. the subnormal range and signed zeros), independent of the conversion
// instruction's mode bits.  ~10 VALU ops per element under an HBM-bound pass: no measurable cost.
__device__ __forceinline__ uint32_t f32_to_e4m3(float v) {
    const uint32_t u = __float_as_uint(v), sign = (u >> 24) & 0x80u;
    uint32_t a = u & 0x7fffffffu;
    a = a > 0x43e00000u ? 0x43e00000u : a;                       // 448 (a product that rounded just above it)
    uint32_t r;
    if (a >= 0x3c800000u) {                                      // >= 2^-6: normal.  3 of 23 mantissa bits, rebias 127 -> 7
        r = ((a + 0x7ffffu + ((a >> 20) & 1u)) >> 20) - (120u << 3);
    } else {                                                     // subnormal: multiples of 2^-9; adding 2^14 rounds to them
        r = __float_as_uint(__uint_as_float(a) + 16384.0f) - 0x46800000u;
    }
    return sign | r;
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
