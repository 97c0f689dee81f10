// Does vector-ALU work of ONE wave hide under that wave's own MFMAs on gfx950?  One wave per SIMD (launch_bounds(256, 1), one
// workgroup per CU), a loop of { one v_mfma_f32_32x32x16_bf16 (8 passes = 32 cycles) + n independent VALU instructions }, n = 0..10,
// several kinds of VALU instruction; MFMA accumulators in AGPRs (first line: in VGPRs).
// Prints cycles per iteration (s_memtime, one wave).  hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void valu(float (&v)[10], unsigned (&w)[5], f32x2 (&p2)[5], int j) {
    if constexpr (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(v[(j + 5) % 10]));
    else if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %1" : "=v"(v[j]) : "v"(v[(j + 5) % 10]));
    else if constexpr (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[j % 5]) : "v"(v[j]), "v"(v[(j + 5) % 10]));
    else if constexpr (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2[j % 5]) : "v"(p2[(j + 2) % 5]));
    else if constexpr (KIND == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p2[j % 5]) : "v"(p2[(j + 2) % 5]), "v"(p2[(j + 3) % 5]));
    else if constexpr (KIND == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[j]) : "v"(v[(j + 5) % 10]), "v"(v[(j + 3) % 10]));
    else if constexpr (KIND == 6) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(v[j]) : "v"(w[j % 5]), "v"(w[(j + 2) % 5]));
    else if constexpr (KIND == 7) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(w[j % 5]) : "v"(v[j]), "v"(v[(j + 5) % 10]), "s"(0x07060302));
    else if constexpr (KIND == 8) asm volatile("v_exp_f16 %0, %1" : "=v"(v[j]) : "v"(v[(j + 5) % 10]));
    else if constexpr (KIND == 9) asm volatile("v_exp_legacy_f32 %0, %1" : "=v"(v[j]) : "v"(v[(j + 5) % 10]));
    else if constexpr (KIND == 10) asm volatile("v_ldexp_f32 %0, %1, %2" : "=v"(v[j]) : "v"(v[(j + 5) % 10]), "v"(w[j % 5]));
    else if constexpr (KIND == 11) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(v[(j + 5) % 10]));
    else if constexpr (KIND == 12) asm volatile("v_rcp_f32 %0, %1" : "=v"(v[j]) : "v"(v[(j + 5) % 10]));
    else asm volatile("v_lshl_add_u32 %0, %1, 23, %2" : "=v"(w[j % 5]) : "v"(w[(j + 1) % 5]), "v"(w[(j + 2) % 5]));
}

template <int N, int KIND, bool AGPR>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, int iters) {
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.5f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.25f + i * 1e-2f); }
    float v[10];
    for (int i = 0; i < 10; ++i) v[i] = 1.0f + i + threadIdx.x * 1e-3f;
    unsigned w[5] = {0, 0, 0, 0, 0};
    f32x2 p2[5];
    for (int i = 0; i < 5; ++i) p2[i] = f32x2{1.0f + i, 0.5f + threadIdx.x * 1e-3f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (AGPR) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
        } else {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            valu<KIND>(v, w, p2, j);
        }
        if constexpr (AGPR) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
        } else {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            valu<KIND>(v, w, p2, j);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 10; ++i) s += v[i];
    for (int i = 0; i < 5; ++i) s += (float)w[i] + p2[i][0] + p2[i][1];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(s != 12345.f); }
}

template <int N, int KIND, bool AGPR>
double run(unsigned long long* d, int iters) {
    hipLaunchKernelGGL((k<N, KIND, AGPR>), dim3(256), dim3(256), 0, 0, d, iters);
    hipLaunchKernelGGL((k<N, KIND, AGPR>), dim3(256), dim3(256), 0, 0, d, iters);
    unsigned long long h[2];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    return (double)h[0] / (2.0 * iters);        // s_memtime ticks (100 MHz) per MFMA + N VALU
}


// Mixed gaps: what the attention stream puts behind each MFMA (2 exp + 2 add + 1 convert) in several orders / splits.
template <int MIX>
__global__ __launch_bounds__(256, 1) void kmix(unsigned long long* out, int iters) {
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.5f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.25f + i * 1e-2f); }
    float v[10];
    for (int i = 0; i < 10; ++i) v[i] = 1.0f + i + threadIdx.x * 1e-3f;
    unsigned w[5] = {0, 0, 0, 0, 0};
    float s0 = 0.f, s1 = 0.f;
#define MF(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(a), "v"(b))
#define EXP(D, S) asm volatile("v_exp_f32 %0, %1" : "=v"(v[D]) : "v"(v[S]))
#define ADD(ACC, S) asm volatile("v_add_f32 %0, %0, %1" : "+v"(ACC) : "v"(v[S]))
#define CVT(D, A, B) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[D]) : "v"(v[A]), "v"(v[B]))
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MIX == 0) {            // the shipped order: exp, add, exp, add, cvt behind every MFMA
            MF(acc0); EXP(0, 5); ADD(s0, 2); EXP(1, 6); ADD(s1, 3); CVT(0, 2, 3);
            MF(acc1); EXP(2, 7); ADD(s0, 0); EXP(3, 8); ADD(s1, 1); CVT(1, 0, 1);
        } else if constexpr (MIX == 1) {     // exps first
            MF(acc0); EXP(0, 5); EXP(1, 6); ADD(s0, 2); ADD(s1, 3); CVT(0, 2, 3);
            MF(acc1); EXP(2, 7); EXP(3, 8); ADD(s0, 0); ADD(s1, 1); CVT(1, 0, 1);
        } else if constexpr (MIX == 2) {     // plain first
            MF(acc0); ADD(s0, 2); ADD(s1, 3); CVT(0, 2, 3); EXP(0, 5); EXP(1, 6);
            MF(acc1); ADD(s0, 0); ADD(s1, 1); CVT(1, 0, 1); EXP(2, 7); EXP(3, 8);
        } else if constexpr (MIX == 3) {     // exps only (2 per MFMA)
            MF(acc0); EXP(0, 5); EXP(1, 6);
            MF(acc1); EXP(2, 7); EXP(3, 8);
        } else if constexpr (MIX == 4) {     // plain only (2 add + 1 cvt per MFMA)
            MF(acc0); ADD(s0, 2); ADD(s1, 3); CVT(0, 2, 3);
            MF(acc1); ADD(s0, 0); ADD(s1, 1); CVT(1, 0, 1);
        } else if constexpr (MIX == 5) {     // uneven: all four exps behind one MFMA, all plain work behind the other
            MF(acc0); EXP(0, 5); EXP(1, 6); EXP(2, 7); EXP(3, 8);
            MF(acc1); ADD(s0, 2); ADD(s1, 3); CVT(0, 2, 3); ADD(s0, 0); ADD(s1, 1); CVT(1, 0, 1);
        } else if constexpr (MIX == 6) {     // one exp per MFMA + plain work (half the exps: what a head_dim of 128 would need)
            MF(acc0); EXP(0, 5); ADD(s0, 2); CVT(0, 2, 3);
            MF(acc1); EXP(2, 7); ADD(s1, 0); CVT(1, 0, 1);
        } else {                             // 2 exp + 1 add + 1 cvt (row sums elsewhere)
            MF(acc0); EXP(0, 5); EXP(1, 6); CVT(0, 2, 3);
            MF(acc1); EXP(2, 7); EXP(3, 8); CVT(1, 0, 1);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = s0 + s1;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 10; ++i) s += v[i];
    for (int i = 0; i < 5; ++i) s += (float)w[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(s != 12345.f); }
}

template <int MIX>
double runmix(unsigned long long* d, int iters) {
    hipLaunchKernelGGL((kmix<MIX>), dim3(256), dim3(256), 0, 0, d, iters);
    hipLaunchKernelGGL((kmix<MIX>), dim3(256), dim3(256), 0, 0, d, iters);
    unsigned long long h[2];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    return (double)h[0] / (2.0 * iters);
}

template <int KIND, bool AGPR>
void sweep(unsigned long long* d, const char* name) {
    const int iters = 20000;
    double r[11] = {run<0, KIND, AGPR>(d, iters), run<1, KIND, AGPR>(d, iters), run<2, KIND, AGPR>(d, iters), run<3, KIND, AGPR>(d, iters),
                    run<4, KIND, AGPR>(d, iters), run<5, KIND, AGPR>(d, iters), run<6, KIND, AGPR>(d, iters), run<7, KIND, AGPR>(d, iters),
                    run<8, KIND, AGPR>(d, iters), run<9, KIND, AGPR>(d, iters), run<10, KIND, AGPR>(d, iters)};
    printf("%-36s", name);
    for (int n = 0; n <= 10; ++n) printf(" n=%d:%6.2f", n, r[n] / r[0] * 32.0);     // in MFMA-normalised cycles (n = 0 is 32)
    printf("   (n=0: %.3f ticks)\n", r[0]);
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    sweep<0, false>(d, "v_add_f32 (accumulators in VGPRs)");
    sweep<0, true>(d, "v_add_f32");
    sweep<5, true>(d, "v_fma_f32");
    sweep<11, true>(d, "v_mov_b32");
    sweep<1, true>(d, "v_exp_f32");
    sweep<8, true>(d, "v_exp_f16");
    sweep<9, true>(d, "v_exp_legacy_f32");
    sweep<12, true>(d, "v_rcp_f32");
    sweep<2, true>(d, "v_cvt_pk_bf16_f32");
    sweep<7, true>(d, "v_perm_b32");
    sweep<3, true>(d, "v_pk_add_f32");
    sweep<4, true>(d, "v_pk_fma_f32");
    sweep<6, true>(d, "v_dot2_f32_bf16");
    sweep<10, true>(d, "v_ldexp_f32");
    sweep<13, true>(d, "v_lshl_add_u32");
    const double base = run<0, 0, true>(d, 20000);
    const char* names[8] = {"exp add exp add cvt (shipped)", "exp exp add add cvt", "add add cvt exp exp", "exp exp", "add add cvt",
                            "4 exp | 4 add 2 cvt (alternating)", "exp add cvt (one exp per MFMA)", "exp exp cvt"};
    double r[8] = {runmix<0>(d, 20000), runmix<1>(d, 20000), runmix<2>(d, 20000), runmix<3>(d, 20000), runmix<4>(d, 20000),
                   runmix<5>(d, 20000), runmix<6>(d, 20000), runmix<7>(d, 20000)};
    for (int i = 0; i < 8; ++i) printf("mixed gap: %-36s %6.2f cycles per MFMA\n", names[i], r[i] / base * 32.0);
    return 0;
}
