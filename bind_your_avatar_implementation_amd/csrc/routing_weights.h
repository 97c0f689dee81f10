// Routing weights of one token (shared by the masked combines of misc.hip and the fused kv-attention + mix kernel of
// attn.hip): the reference's G1 / G2 weights, models/transformer.py:821-822 (face) and :860-863,895-900 (audio).
#pragma once
#include "bya_common.h"

namespace {

// Routing weights of one token.  mode 0 (face): w[id] = r[id].  mode 1 (audio): av = af @ r (bf16 bmm output), then for
// two streams the reference's  w = 1 - av[[1, 0]]  (models/transformer.py:895-900); for more streams -- the reference
// hard-codes two -- the build-defined generalisation  w[a] = prod_{b != a} (1 - av[b])  evaluated as a chain of bf16 tensor
// ops like oracle/model.py::audio_weights ("not any other speaker's region"; identical bits for two streams).
template <int NID>
__device__ __forceinline__ void audio_weights_n(const bf16_t* __restrict__ af, const bf16_t* __restrict__ r, float (&w)[4]) {
    float rv[NID], om[NID];
#pragma unroll
    for (int i = 0; i < NID; ++i) rv[i] = bf2f(r[i]);
#pragma unroll
    for (int a = 0; a < NID; ++a) {
        float av = 0.f;
#pragma unroll
        for (int i = 0; i < NID; ++i) av = fmaf(bf2f(af[a * NID + i]), rv[i], av);
        om[a] = bf2f(f2bf(1.0f - bf2f(f2bf(av))));
    }
#pragma unroll
    for (int a = 0; a < NID; ++a) {
        float t = 1.0f;
#pragma unroll
        for (int bb = 0; bb < NID; ++bb)
            if (bb != a) t = bf2f(f2bf(t * om[bb]));
        w[a] = t;
    }
}

// w[0 .. n_id) of one token: mode 0 (face) = the logits themselves, mode 1 (audio) = the complemented, swapped mixture.
__device__ __forceinline__ void routing_weights_of(int mode, int n_id, const bf16_t* __restrict__ af, const bf16_t* r, float (&w)[4]) {
    if (mode == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = i < n_id ? bf2f(r[i]) : 0.f;
        return;
    }
    w[2] = w[3] = 0.f;
    switch (n_id) {            // unrolled per count: everything stays in registers (the two-stream form is the hot one)
        case 2: audio_weights_n<2>(af, r, w); break;
        case 3: audio_weights_n<3>(af, r, w); break;
        default: audio_weights_n<4>(af, r, w); break;
    }
}

}  // namespace
