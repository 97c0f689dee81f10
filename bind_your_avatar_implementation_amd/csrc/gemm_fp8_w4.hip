// One-wave-per-SIMD form of the e4m3 GEMM: 256 x 256 tiles, 4 waves with 128 x 128 wave tiles -- 64 accumulator tiles in
// 256 AGPRs, 16 fragments of 32 bytes in 128 VGPRs, 0.5 LDS reads per MFMA (the 64 x 64 wave tiles of the default form
// need 1.0, which saturates the LDS port at the e4m3 rate).  Same template as gemm_fp8.hip; this translation unit is
// built WITHOUT -amdgpu-mfma-vgpr-form so that the accumulators live in AGPRs (build.py, AGPR_SOURCES).
#include "gemm_fp8_kernel.h"

int bya_launch_gemm_fp8_w4(const void* args, const float* sa, const float* sw, int batch, hipStream_t stream) {
    return launch_fp8<256, 256, 2, 2>(*static_cast<const GemmArgs*>(args), sa, sw, batch, stream);
}
