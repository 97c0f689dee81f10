#!/usr/bin/env python3
"""Micro-benchmarks of the individual HIP kernels at the 49x480x720 shapes (HIP-event timing, random data)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rnd(*shape, std=1.0):
    return (torch.randn(*shape, device=dev) * std).to(torch.bfloat16)


def gemm_case(M, N, K, **kw):
    a, w, out = rnd(M, K), rnd(N, K, std=K ** -0.5), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    b = rnd(N)
    t = timeit(lambda: ops.gemm(a, w, out, bias=b, **kw))
    # cold: rotate over enough distinct operands (> 1 GB) that nothing is served from L2 / Infinity Cache
    nrot = max(2, int(1.2e9 // ((M * K + N * K + M * N) * 2)) + 1)
    As = [rnd(M, K) for _ in range(nrot)]
    Ws = [rnd(N, K, std=K ** -0.5) for _ in range(nrot)]
    Os = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(nrot)]
    it = [0]
    def cold():
        i = it[0] % nrot
        it[0] += 1
        ops.gemm(As[i], Ws[i], Os[i], bias=b, **kw)
    tc = timeit(cold, iters=2 * nrot, warm=nrot)
    print(f"gemm M={M} N={N} K={K} {kw}: hot {t*1e3:.3f} ms {2*M*N*K/t/1e12:.1f} TF/s | cold {tc*1e3:.3f} ms "
          f"{2*M*N*K/tc/1e12:.1f} TF/s", flush=True)


def attn_case(B, S, H, D=64):
    q, k, v = rnd(B, S, H * D), rnd(B, S, H * D), rnd(B, S, H * D)
    out = torch.empty_like(q)
    t = timeit(lambda: ops.self_attention(q, k, v, out, heads=H, head_dim=D), iters=5, warm=2)
    print(f"attn B={B} S={S} H={H} D={D}: {t*1e3:.3f} ms  {4*B*H*S*S*D/t/1e12:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    S = 17776
    gemm_case(S, 3072, 3072)
    gemm_case(S, 12288, 3072, act="gelu_tanh")
    gemm_case(S, 3072, 12288)
    gemm_case(17550, 2048, 3072)
    gemm_case(35100, 3072, 2048)
    gemm_case(35100, 512, 512)
    gemm_case(8192, 8192, 8192)
    attn_case(1, S, 48)
    attn_case(26, 1350, 8)
    x = rnd(1, S, 3072)
    y = torch.empty_like(x)
    w, b = rnd(3072), rnd(3072)
    mods = rnd(1, 6 * 3072)
    t = timeit(lambda: ops.layernorm(x, y, w, b, shift0=mods[:, 9216:], scale0=mods[:, 12288:], shift1=mods, scale1=mods[:, 3072:], split=226, mod_batch_stride=18432))
    print(f"adaln S={S}: {t*1e6:.1f} us  {2*x.numel()*2/t/1e12:.2f} TB/s")
    q, k = rnd(1, S, 3072), rnd(1, S, 3072)
    cos, sin = torch.randn(17550, 64, device=dev), torch.randn(17550, 64, device=dev)
    w64 = rnd(64)
    t = timeit(lambda: ops.qknorm_rope(q, k, w64, w64, w64, w64, cos, sin, heads=48, text_rows=226))
    print(f"qknorm_rope: {t*1e6:.1f} us  {4*q.numel()*2/t/1e12:.2f} TB/s")
    feat = rnd(1, 2, 17550, 3072)
    r = torch.rand(1, 17550, 2, device=dev).to(torch.bfloat16)
    t = timeit(lambda: ops.masked_combine(x[:, 226:], feat, r, None, "face"))
    print(f"combine: {t*1e6:.1f} us  {4*17550*3072*2/t/1e12:.2f} TB/s")
