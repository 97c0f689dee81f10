#!/usr/bin/env python3
"""Kernel efficiency at the PER-RANK shapes of the sharded step: there is one GPU per box, so
the N-GPU step cannot be run; what can be measured is every kernel of one rank's share at the shapes that rank sees
(sequence-parallel over W ranks: S_loc = 17776 / W rows, 48 / W heads with whole sequences, router partitions of
ceil(26 / W) (id, frame) pairs and ceil(1350 / W) locations), on one MI355X.  From the per-launch times and the launch
counts of a step it projects the per-rank COMPUTE time of an N-GPU step; the exchanges are priced from their bytes and
the xGMI link rate (MI355X_MICROARCH.md / task statement: 7 links x ~153 GB/s per GPU, point to point), fully exposed
except for the v exchange that runs under the q/k-norm kernel.  Gaussian operands, median of interleaved rounds.

  python tools/shard_shape_probe.py [--world 8] [--out gpurun_out/shard_shapes_w8.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bind_your_avatar_implementation_amd import ops

dev = torch.device("cuda:0")
S, TT, N, D, H, L = 17776, 226, 17550, 3072, 48, 42
PER_FRAME, T, NID = 1350, 13, 2
LINK_GBPS = 153.0


def rnd(*shape, std=1.0):
    return (torch.randn(*shape, device=dev) * std).to(torch.bfloat16)


def timeit(fn, iters=10, rounds=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e-3)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    W = a.world
    S_loc, N_loc, Hl = S // W, N // W, H // W
    pairs = NID * T
    nPA, nLB = -(-pairs // W), -(-PER_FRAME // W)
    rows = []

    def gemm(name, M, Nn, K, count, **kw):
        x, w, out = rnd(M, K), rnd(Nn, K, std=K ** -0.5), torch.empty(M, Nn, dtype=torch.bfloat16, device=dev)
        b = rnd(Nn)
        if kw.pop("res", False):
            kw["res"] = out
        t = timeit(lambda: ops.gemm(x, w, out, bias=b, **kw))
        rows.append(dict(kernel=f"gemm {name}", shape=f"{M}x{Nn}x{K}", us=t * 1e6, tflops=2.0 * M * Nn * K / t / 1e12,
                         per_step=count))

    gemm("qkv", S_loc, 3 * D, D, L)
    gemm("attn_out(+res)", S_loc, D, D, L, res=True)
    gemm("ff1(gelu)", S_loc, 4 * D, D, L, act="gelu_tanh")
    gemm("ff2(+res)", S_loc, D, 4 * D, L, res=True)
    gemm("audio_q", N_loc, D, D, L)
    gemm("audio_out(+res)", N_loc, D, D, L, res=True)
    gemm("perceiver_q", N_loc, 2048, D, L // 2)
    gemm("perceiver_out(+res)", N_loc, D, 2048, L // 2, res=True)
    gemm("router_q", N_loc, 2048, 2048, L // 2)

    # joint attention on whole sequences of this rank's heads (the block order is not XCD-aligned when Hl % 8 != 0)
    q, k, v = rnd(1, S, Hl * 64), rnd(1, S, Hl * 64) * 0.18, rnd(1, S, Hl * 64)
    o = torch.empty_like(q)
    t = timeit(lambda: ops.self_attention(q, k, v, o, heads=Hl, prescaled=True, score_bound=11.8), iters=5)
    rows.append(dict(kernel="joint attention", shape=f"S={S} heads={Hl}", us=t * 1e6,
                     tflops=4.0 * S * S * 64 * Hl / t / 1e12, per_step=L))
    # row-local elementwise kernels on S_loc rows
    x, y = rnd(1, S_loc, D), torch.empty(1, S_loc, D, dtype=torch.bfloat16, device=dev)
    w, b, mods = rnd(D), rnd(D), rnd(1, 6 * D)
    t = timeit(lambda: ops.layernorm(x, y, w, b, shift0=mods[:, 3 * D:], scale0=mods[:, 4 * D:], shift1=mods, scale1=mods[:, D:],
                                      split=TT, mod_batch_stride=6 * D))
    rows.append(dict(kernel="adaln layernorm", shape=f"{S_loc}x{D}", us=t * 1e6, gbps=2 * x.numel() * 2 / t / 1e9, per_step=2 * L + L + L // 2))
    qq, kk = rnd(1, S_loc, D), rnd(1, S_loc, D)
    cos, sin, w64 = torch.randn(S_loc, 64, device=dev), torch.randn(S_loc, 64, device=dev), rnd(64)
    t = timeit(lambda: ops.qknorm_rope(qq, kk, w64, w64, w64, w64, cos, sin, heads=H, text_rows=0))
    rows.append(dict(kernel="qknorm_rope", shape=f"{S_loc}x{D} x2", us=t * 1e6, gbps=4 * qq.numel() * 2 / t / 1e9, per_step=L))
    # router partitions: frame-major (spatial attention) and location-major (everything else)
    RA, RB = nPA * PER_FRAME, pairs * nLB
    for name, R in (("frame-major", RA), ("location-major", RB)):
        xr = rnd(R, 512)
        o3, o1 = torch.empty(R, 1536, dtype=torch.bfloat16, device=dev), torch.empty(R, 512, dtype=torch.bfloat16, device=dev)
        pk3 = ops.pack_rowgemm512(rnd(1536, 512) * 0.04, rnd(1536), rnd(512), rnd(512))
        pk1 = ops.pack_rowgemm512(rnd(512, 512) * 0.04, rnd(512))
        n3, n1 = (1, 1) if name == "frame-major" else (2, 3)          # launches per ST block in that partition
        t = timeit(lambda: ops.rowgemm512(xr, pk3, o3))
        rows.append(dict(kernel=f"rowgemm LN+qkv ({name})", shape=f"{R}x1536x512", us=t * 1e6, tflops=2.0 * R * 1536 * 512 / t / 1e12,
                         per_step=n3 * 4 * (L // 2)))
        t = timeit(lambda: ops.rowgemm512(xr, pk1, o1, res=o1))
        rows.append(dict(kernel=f"rowgemm out+res ({name})", shape=f"{R}x512x512", us=t * 1e6, tflops=2.0 * R * 512 * 512 / t / 1e12,
                         per_step=(n1 + (1 if name != "frame-major" else 0)) * 4 * (L // 2)))
    qkv = rnd(RA, 1536)
    ra = torch.empty(RA, 512, dtype=torch.bfloat16, device=dev)
    F = 512
    t = timeit(lambda: ops.attention(qkv, qkv[:, F:], qkv[:, 2 * F:], ra, head_dim=64, heads=8, nb1=nPA, nb2=1, Sq=PER_FRAME,
                                     Skv=PER_FRAME, q_strides=(PER_FRAME * 3 * F, 0, 3 * F), k_strides=(PER_FRAME * 3 * F, 0, 3 * F),
                                     v_strides=(PER_FRAME * 3 * F, 0, 3 * F), o_strides=(PER_FRAME * F, 0, F), scale=0.125))
    rows.append(dict(kernel="router spatial attention", shape=f"{nPA} pairs x 8 heads x {PER_FRAME}^2", us=t * 1e6,
                     tflops=4.0 * nPA * 8 * PER_FRAME ** 2 * 64 / t / 1e12, per_step=4 * (L // 2)))
    qkvb = rnd(RB, 1536)
    rb = torch.empty(RB, 512, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.attn_tiny(qkvb, qkvb[:, F:], qkvb[:, 2 * F:], rb, T, 8, NID, nLB, T * nLB, nLB, 3 * F, F, 0.125))
    rows.append(dict(kernel="router temporal attention", shape=f"{RB} rows", us=t * 1e6, per_step=4 * (L // 2)))
    t = timeit(lambda: ops.attn_tiny(qkvb, qkvb[:, F:], qkvb[:, 2 * F:], rb, NID, 8, 1, T * nLB, 0, T * nLB, 3 * F, F, 0.125))
    rows.append(dict(kernel="router multi-id attention", shape=f"{RB} rows", us=t * 1e6, per_step=4 * (L // 2)))

    compute_ms = sum(r["us"] * r["per_step"] for r in rows) / 1e3
    # exchanges (bytes RECEIVED per rank and step; every element crosses one link once, W - 1 links busy in parallel)
    a2a_attn = 4 * L * S_loc * D * 2 * (W - 1) / W                       # q, k, v in and o out, head-parallel
    a2a_router = 7 * (L // 2) * pairs * PER_FRAME * 512 * 2 / W * (W - 1) / W + (L // 2) * NID * N * 512 * 2 * (W - 1) / W
    comm_ms = (a2a_attn + a2a_router) / ((W - 1) * LINK_GBPS * 1e9) * 1e3
    launches = 4 * L + 9 * (L // 2)
    comm_latency_ms = launches * 0.02                                    # ~20 us per collective launch + sync
    res = {"world": W, "rows": rows, "projected_ms_per_step": {
        "compute_per_rank": round(compute_ms, 2), "exchange_bytes_over_links": round(comm_ms, 2),
        "exchange_launch_latency(20us each)": round(comm_latency_ms, 2),
        "total_if_nothing_overlaps": round(compute_ms + comm_ms + comm_latency_ms, 2)},
        "note": "per-rank kernels timed on ONE MI355X at the shapes of a W-rank sequence-parallel step; small kernels not "
                "listed (routed mixes, router scores / head, small-M linears, audio / perceiver cross-attention) add "
                "about (7.7 + 3.3 + 2.1 + 17) / W ms"}
    for r in rows:
        eff = f"{r['tflops']:7.0f} TFLOP/s" if "tflops" in r else (f"{r['gbps']:7.0f} GB/s" if "gbps" in r else " " * 14)
        print(f"{r['kernel']:34s} {r['shape']:28s} {r['us']:9.1f} us  {eff}  x{r['per_step']}")
    print(json.dumps(res["projected_ms_per_step"]))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
