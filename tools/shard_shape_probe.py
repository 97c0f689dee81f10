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
    rows.append(dict(kernel="qknorm_rope (q and k)", shape=f"{S_loc}x{D} x2", us=t * 1e6, gbps=4 * qq.numel() * 2 / t / 1e9, per_step=L))
    # the two cross-attentions with the masked combine in their epilogue (bya_attn_kv_mix): audio = one launch per (partial)
    # frame of this rank's rows, perceiver = one launch
    segs = -(-N_loc // PER_FRAME) + 1
    seg_rows = N_loc // segs
    qa, ka, va = rnd(seg_rows, D), rnd(NID, 32, D), rnd(NID, 32, D)
    rl = torch.sigmoid(torch.randn(seg_rows, NID, device=dev)).to(torch.bfloat16)
    afm = torch.eye(NID, device=dev, dtype=torch.bfloat16)
    za, wsum = torch.empty(seg_rows, D, dtype=torch.bfloat16, device=dev), torch.empty(seg_rows, dtype=torch.float32, device=dev)
    t = timeit(lambda: ops.attn_kv_mix(qa, ka, va, rl, afm, za, wsum, head_dim=64, heads=H, n_id=NID, n_grp=1, Sq=seg_rows, Skv=32,
                                       q_strides=(0, D), k_strides=(32 * D, 0, D), v_strides=(32 * D, 0, D), z_strides=(0, D), scale=0.125))
    rows.append(dict(kernel="audio cross-attention + mix", shape=f"{seg_rows} rows x {H} heads x 32 keys", us=t * 1e6, per_step=L * segs))
    qp_, kvp = rnd(N_loc, 2048), rnd(NID, 32, 4096)
    rl2 = torch.sigmoid(torch.randn(N_loc, NID, device=dev)).to(torch.bfloat16)
    zp = torch.empty(N_loc, 2048, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.attn_kv_mix(qp_, kvp, kvp[..., 2048:], rl2, None, zp, None, head_dim=128, heads=16, n_id=NID, n_grp=1, Sq=N_loc,
                                       Skv=32, q_strides=(0, 2048), k_strides=(32 * 4096, 0, 4096), v_strides=(32 * 4096, 0, 4096),
                                       z_strides=(0, 2048), scale=128 ** -0.5))
    rows.append(dict(kernel="perceiver cross-attention + mix", shape=f"{N_loc} rows x 16 heads x 32 keys", us=t * 1e6, per_step=L // 2))
    # router pre-stage on this rank's rows: LayerNorm(2048), scores + LN(512) + positions
    qn, qn2, g2, b2 = rnd(1, N_loc, 2048), rnd(1, N_loc, 2048), rnd(2048), rnd(2048)
    t = timeit(lambda: ops.layernorm(qn, qn2, g2, b2))
    rows.append(dict(kernel="router norm_q", shape=f"{N_loc}x2048", us=t * 1e6, per_step=L // 2))
    qr, krr = rnd(N_loc, 2048), rnd(NID, 32, 2048)
    rs = torch.empty(NID, N_loc, 512, dtype=torch.bfloat16, device=dev)
    g5, b5, pos5 = rnd(512), rnd(512), rnd(N_loc, 512)
    t = timeit(lambda: ops.router_scores(qr, krr, g5, b5, pos5, rs, NID, N_loc))
    rows.append(dict(kernel="router scores", shape=f"{N_loc} tokens", us=t * 1e6, per_step=L // 2))
    # router partitions: frame-major (spatial attention) and location-major (everything else)
    RA, RB = nPA * PER_FRAME, pairs * nLB
    for name, R in (("frame-major", RA), ("location-major", RB)):
        xr = rnd(R, 512)
        o3, o1 = torch.empty(R, 1536, dtype=torch.bfloat16, device=dev), torch.empty(R, 512, dtype=torch.bfloat16, device=dev)
        pk3 = ops.pack_rowgemm512(rnd(1536, 512) * 0.04, rnd(1536), rnd(512), rnd(512))
        pk1 = ops.pack_rowgemm512(rnd(512, 512) * 0.04, rnd(512))
        n3, n1 = (1, 1) if name == "frame-major" else (0, 3)          # launches per ST block in that partition
        if n3:
            t = timeit(lambda: ops.rowgemm512(xr, pk3, o3))
            rows.append(dict(kernel=f"rowgemm LN+qkv ({name})", shape=f"{R}x1536x512", us=t * 1e6, tflops=2.0 * R * 1536 * 512 / t / 1e12,
                             per_step=n3 * 4 * (L // 2)))
        else:
            # round 4: LayerNorm -> q|k|v -> temporal / multi-ID attention in ONE launch each (bya_router_group_attn)
            t = timeit(lambda: ops.router_group_attn(xr, pk3, o1, T, NID, nLB, T * nLB, nLB))
            rows.append(dict(kernel="fused LN+qkv+temporal attention", shape=f"{R} rows", us=t * 1e6, per_step=4 * (L // 2)))
            t = timeit(lambda: ops.router_group_attn(xr, pk3, o1, NID, 1, T * nLB, 0, T * nLB))
            rows.append(dict(kernel="fused LN+qkv+multi-id attention", shape=f"{R} rows", us=t * 1e6, per_step=4 * (L // 2)))
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
    xl = rnd(NID, T * nLB, 512)
    lg = torch.empty(T * nLB, NID, dtype=torch.bfloat16, device=dev)
    hw, hb = rnd(1, 512), rnd(1)
    t = timeit(lambda: ops.router_head(xl, hw, hb, lg, NID, T * nLB))
    rows.append(dict(kernel="router head", shape=f"{T * nLB} tokens", us=t * 1e6, per_step=L // 2))
    compute_ms = sum(r["us"] * r["per_step"] for r in rows) / 1e3
    # ---- exchanges on the P2P transport (round 4): ONE push kernel + one wait kernel per exchange.  Their fixed cost is
    # measured here (a 1-rank group: the push stores into this GPU's own buffer, so the time is launch + flag protocol,
    # not link time); the bytes are priced at the xGMI link rate, W - 1 links busy in parallel, every element crosses one
    # link once.  Per step: joint attention 2 per layer (packed q|k|v in, o out); router per routing layer 1 (tokens ->
    # pair owners) + 4 (frame-major -> location-major) + 3 (back) + 1 (logits) = 9; 1 output gather.
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29833")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
    from bind_your_avatar_implementation_amd.p2p import P2PGroup
    grp = P2PGroup(dist.group.WORLD, dev)
    rb_ = grp.symmetric("probe", (3 * W, 64), torch.bfloat16)
    srcs = [rnd(64) for _ in range(3 * W)]
    ch = grp.channel("probe", [(srcs[i], 0, "probe", i * 64) for i in range(3 * W)])
    t_x = timeit(lambda: ch.exchange(), iters=50)
    exchanges = 2 * L + 9 * (L // 2) + 1
    a2a_attn = 4 * L * S_loc * D * 2 * (W - 1) / W                       # q, k, v in and o out, head-parallel
    a2a_router = (7 * pairs * PER_FRAME * 512 * 2 / W + NID * N_loc * 512 * 2) * (L // 2) * (W - 1) / W
    comm_ms = (a2a_attn + a2a_router) / ((W - 1) * LINK_GBPS * 1e9) * 1e3
    comm_latency_ms = exchanges * t_x * 1e3
    res = {"world": W, "rows": rows, "exchange": {"per_step": exchanges, "push_plus_wait_us_measured_on_one_gpu": round(t_x * 1e6, 2),
                                                  "rccl_collectives_per_step": 0},
           "projected_ms_per_step": {
        "compute_per_rank": round(compute_ms, 2), "exchange_bytes_over_links": round(comm_ms, 2),
        "exchange_fixed_cost(push+wait kernels)": round(comm_latency_ms, 2),
        "total_if_nothing_overlaps": round(compute_ms + comm_ms + comm_latency_ms, 2)},
        "note": "per-rank kernels timed on ONE MI355X at the shapes of a W-rank sequence-parallel step; every kernel of the step is "
                "listed except the replicated step-invariant conditioning (face extractor, audio projector, their K/V: ~7 ms on "
                "every rank unless cached by precompute_conditioning) and the six small-M linears (0.6 ms)"}
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
