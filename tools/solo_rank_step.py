#!/usr/bin/env python3
"""ONE rank's step of a W-rank sequence-parallel denoise step, run alone on one MI355X.

There is one GPU per box, so the N-GPU step cannot be run.  tools/shard_shape_probe.py times every per-rank kernel in a
micro-benchmark loop and adds them up; this tool runs the REAL engine instead: the model is sharded as rank r of W
(``P2PGroup(solo=(r, W))``: every peer buffer is a second local allocation, a wait only expects this rank's own flag), so
the whole rank-step -- every kernel at its per-rank shape, every exchange launch, the host's launch sequence or a hipGraph
replay of it -- executes exactly as on a node, except that (1) pushes store into local HBM instead of a peer's, (2) nothing is
ever waited for, (3) the numbers it computes are wrong (the peers' rows never arrive).  The link time of the bytes a rank
sends is priced separately (7 links x 153 GB/s, nothing overlapped) and ADDED, although the local copy the push kernels make
here already costs time: the projection errs on the slow side there; it cannot see rank skew.

  python tools/solo_rank_step.py [--world 8] [--rank 1] [--steps 5] [--out gpurun_out/solo_w8.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

LINK_GBPS = 153.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=1, help="1 = a rank without text rows (2222 video rows at W = 8): the common kind")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=42)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--single", action="store_true", help="also time the unsharded step in this process (the 1-GPU reference)")
    ap.add_argument("--cached-conditioning", action="store_true", help="precompute_conditioning(): what the pipeline does")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from bench import MODEL_KW
    from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel, ops
    from bind_your_avatar_implementation_amd import parallel
    from bind_your_avatar_implementation_amd.p2p import P2PGroup
    from bind_your_avatar_implementation_amd.synth import synth_inputs
    dev = torch.device("cuda:0")
    model = BindyouravatarTransformer3DModel(**dict(MODEL_KW, num_layers=a.layers), device=dev).init_synthetic(seed=0, fast=True)
    inp = synth_inputs(batch=1, seed=0, device="cpu")
    inp = {k: (v.to(dev, torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else
               (v.to(dev) if torch.is_tensor(v) else v)) for k, v in inp.items()}
    inp["image_rotary_emb"] = tuple(t.to(dev, torch.float32) for t in inp["image_rotary_emb"])
    inp["id_cond"] = [t.to(dev, torch.bfloat16) for t in inp["id_cond"]]
    inp["id_vit_hidden"] = [[t.to(dev, torch.bfloat16) for t in l] for l in inp["id_vit_hidden"]]

    def step():
        return model(return_dict=False, denoise_step=0, **inp)[0]

    def timed(tag):
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        print(f"{tag}: {ms:.2f} ms per step", flush=True)
        return ms

    res = {"world": a.world, "rank": a.rank, "layers": a.layers, "steps": a.steps}
    if a.single:
        res["single_gpu_eager_ms"] = timed("unsharded, eager")
    W, r = a.world, a.rank
    model._seq_group, model._seq_world, model._seq_rank = "solo-probe", W, r          # (the group object is never dereferenced:
    model._seq_p2p = P2PGroup(device=dev, solo=(r, W))                                #  every exchange takes the P2P path)
    model._seq_transport = "p2p"
    model.invalidate_engine()
    if a.cached_conditioning:
        model.precompute_conditioning(inp["id_cond"], inp["id_vit_hidden"], inp["audio_embeds"], 13)
    parallel.COLLECTIVE_CALLS.clear()
    step()
    torch.cuda.synchronize()
    res["exchanges_per_step"] = parallel.COLLECTIVE_CALLS.get("p2p_exchange", 0)
    sh = next(p for k, p in model._engine._parts.items() if k[0] == "seq")
    res["rows_of_this_rank"] = {"S_loc": sh.S_loc, "text": sh.Tt_loc, "video": sh.N_loc}
    res["rank_step_eager_ms"] = timed(f"rank {r} of {W}, eager")
    # per-kernel events (eager): where the rank-step goes
    ops.enable_kernel_timers()
    step()
    torch.cuda.synchronize()
    kt = ops.collect_kernel_timers()
    res["kernel_ms_per_step"] = {k: round(sum(v) * 1e3, 3) for k, v in sorted(kt.items(), key=lambda kv: -sum(kv[1]))}
    res["launches_per_step"] = {k: len(v) for k, v in kt.items()}
    if not a.no_graph and not a.cached_conditioning:
        model.use_hip_graph = True
        res["rank_step_graph_replay_ms"] = timed(f"rank {r} of {W}, hipGraph replay")
        model.use_hip_graph = False
    # bytes this rank sends per step (every element crosses one link once): head-parallel q|k|v in and o out per layer, the
    # router's repartitions per routing layer, the output gather
    S, D, L, N, pairs, pf = 17776, 3072, a.layers, 17550, 26, 1350
    S_loc, N_loc = sh.S_loc, sh.N_loc
    attn = 4 * L * S_loc * D * 2 * (W - 1) / W
    router = (7 * pairs * pf * 512 * 2 / W + 2 * N_loc * 512 * 2) * (L // 2) * (W - 1) / W
    out_gather = N_loc * 64 * 2 * (W - 1)
    link_ms = (attn + router + out_gather) / ((W - 1) * LINK_GBPS * 1e9) * 1e3
    res["bytes_sent_per_step"] = {"joint_attention": attn, "router": router, "output": out_gather}
    res["link_ms_at_7x153GBps_nothing_overlapped"] = link_ms
    # (r6) exchange A with v first: a quarter of the joint attention's bytes (v out of q, k, v, o) travels on the side stream
    # underneath the q | k projection; hidden as far as that projection (two thirds of the packed one, timed per launch in the
    # event pass above) lasts longer than the v push's link time -- the model's only overlap
    eng = model._engine
    v_first = bool(getattr(eng, "sp_overlap_v", False)) and S_loc >= eng.SP_OVERLAP_V_MIN_ROWS
    hidden = 0.0
    if v_first:
        v_link_ms = (attn / 4) / ((W - 1) * LINK_GBPS * 1e9) * 1e3
        qk_gemm_ms = L * (2.0 * S_loc * 2 * D * D / 1.0e15) * 1e3              # q | k projection at ~1000 TFLOP/s (shard shapes)
        hidden = min(v_link_ms, qk_gemm_ms)
    res["exchange_a_v_first"] = {"active": v_first, "link_ms_hidden_under_the_qk_projection": hidden}
    link_ms -= hidden
    res["link_ms_exposed_in_the_model"] = link_ms
    best = min(res["rank_step_eager_ms"], res.get("rank_step_graph_replay_ms", 1e9))
    res["projected_rank_step_ms"] = {"eager": res["rank_step_eager_ms"] + link_ms,
                                     "graph_replay": (res["rank_step_graph_replay_ms"] + link_ms) if "rank_step_graph_replay_ms" in res else None,
                                     "note": "measured solo rank-step + EXPOSED link time of the bytes sent (all of it except the v third of exchange A when "
                                             "that travels underneath the q | k projection; the local copy the push kernels make here is already "
                                             "inside the measured step); no rank skew"}
    if "single_gpu_eager_ms" in res:
        res["projected_speedup_vs_single_gpu_eager"] = res["single_gpu_eager_ms"] / (best + link_ms)
    ops.check_gemm_workspace()
    print(json.dumps({k: v for k, v in res.items() if k not in ("kernel_ms_per_step", "launches_per_step")}))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
