#!/usr/bin/env python3
"""Root-cause experiment for the wrong elements seen when several PROCESSES share one MI355X (round-1 DESIGN.md
section 5; VERDICT "Root-cause the time-sliced corruption").

Two processes on cuda:0.  The child is a DISTURBER that keeps one kernel running back to back; the parent runs a
VICTIM kernel R times on the same input and compares every run bit for bit with the result it computed while the
disturber was idle.  Hypotheses separated by the matrix below:

  * in-place read-modify-write victims vs the same kernel writing to a second buffer   (VERDICT's hypothesis);
  * our kernels vs a plain torch op as victim;
  * disturbers by LDS footprint: a self-contained LDS hog at 64 ... 160 KiB per workgroup, our row GEMM at
    132 / 140 KiB, our 256x256 GEMM at 128 KiB, our attention at 32 KiB.

  python tools/timeslice/repro.py [--runs 40] [--out gpurun_out/timeslice_repro.json]
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
LIB = os.path.join(HERE, "liblds_hog.so")


NOSLP = os.path.join(HERE, "libnorm_noslp.so")


def build():
    src = os.path.join(HERE, "lds_hog.hip")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", LIB])
    # the product's norm.hip once more, compiled WITHOUT the SLP vectoriser: no v_pk_* (packed fp32) instructions at all
    nsrc = os.path.join(ROOT, "bind_your_avatar_implementation_amd", "csrc", "norm.hip")
    if not os.path.exists(NOSLP) or os.path.getmtime(NOSLP) < os.path.getmtime(nsrc):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                               "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1", nsrc, "-o", NOSLP])
    return LIB


def load_hog():
    import torch  # noqa: F401  (one HIP runtime per process: torch first)
    lib = ctypes.CDLL(LIB)
    lib.hog_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.victim_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
    lib.bcast_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_void_p]
    return lib


# *_sentinel: the same disturber, but EVERY buffer of the disturber process (plus a 3 GiB ballast allocated first, so that
# it covers the virtual addresses the victim process uses) holds the bit pattern of fp32 123.0 -- a victim output that
# shows the sentinel proves that bytes crossed from the other process.
DISTURBERS = ["none", "hog64", "hog96", "hog128", "hog132", "hog136", "hog144", "hog160", "rowgemm_n512_132k",
              "rowgemm_n1536_140k", "gemm256_128k", "attn_32k", "torch_matmul", "rowgemm_sentinel"]
VICTIMS = ["rmw_inplace", "rmw_out", "qknorm_rope_inplace", "qknorm_rope_noslp_inplace", "qknorm_rope_dbg1_inplace", "qknorm_rope_dbg2_inplace",
           "qknorm_rope_sc1_inplace", "bcast_table", "bcast_table_sc1",
           "qknorm_norope_inplace", "layernorm_out", "layernorm_inplace", "torch_mul_inplace",
           "torch_layernorm_out"]


def disturber_main(cmd_q, ack_q):
    import torch
    from bind_your_avatar_implementation_amd import ops
    dev = torch.device("cuda:0")
    hog = load_hog()
    ballast = torch.full((3 << 28,), 0x42F60000, dtype=torch.int32, device=dev)       # 3 GiB of fp32 123.0, allocated FIRST
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    bf = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
    x512, o1536, o512 = bf(35100, 512), torch.empty(35100, 1536, dtype=torch.bfloat16, device=dev), bf(35100, 512)
    pk1536 = ops.pack_rowgemm512(bf(1536, 512) * 0.04, bf(1536), bf(512), bf(512))
    pk512 = ops.pack_rowgemm512(bf(512, 512) * 0.04, bf(512), bf(512), bf(512))
    ga, gw, go = bf(8192, 3072), bf(3072, 3072) * 0.02, torch.empty(8192, 3072, dtype=torch.bfloat16, device=dev)
    aq, ak, av = bf(1, 4096, 3072), bf(1, 4096, 3072), bf(1, 4096, 3072)
    ao = torch.empty_like(aq)
    stream = torch.cuda.current_stream().cuda_stream
    sent = lambda *s: torch.full(s, 123.0, dtype=torch.bfloat16, device=dev)
    sx, so = sent(35100, 512), sent(35100, 1536)
    spk = ops.pack_rowgemm512(sent(1536, 512), sent(1536), sent(512), sent(512))
    for k in ("colsum", "cvec"):
        spk[k].fill_(123.0)
    print("disturber: ballast at", hex(ballast.data_ptr()), "x512 at", hex(x512.data_ptr()), flush=True)

    def once(name):
        if name.startswith("hog"):
            assert hog.hog_launch(sink.data_ptr(), 256, 40, int(name[3:]) * 1024, stream) == 0
        elif name == "rowgemm_n1536_140k":
            ops.rowgemm512(x512, pk1536, o1536)
        elif name == "rowgemm_sentinel":
            ops.rowgemm512(sx, spk, so)
            so.fill_(123.0)
        elif name == "rowgemm_n512_132k":
            ops.rowgemm512(x512, pk512, o512)
        elif name == "gemm256_128k":
            ops.gemm(ga, gw, go)
        elif name == "attn_32k":
            ops.self_attention(aq, ak, av, ao, heads=48)
        elif name == "torch_matmul":
            torch.matmul(ga, gw.t())
        else:
            time.sleep(0.001)

    cur = "none"
    while True:
        if not cmd_q.empty():
            cur = cmd_q.get()
            torch.cuda.synchronize()
            if cur == "quit":
                ack_q.put("bye")
                return
            for _ in range(4):
                once(cur)                       # the disturber is in flight before the victim starts
            ack_q.put(cur)
        for _ in range(8):
            once(cur)
        torch.cuda.synchronize()                # bounded queue depth: a stop request is seen within a few launches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=40)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "timeslice_repro.json"))
    ap.add_argument("--disturbers", default=",".join(DISTURBERS))
    ap.add_argument("--victims", default=",".join(VICTIMS))
    args = ap.parse_args()
    build()
    import torch
    import torch.multiprocessing as mp
    mp.set_start_method("spawn", force=True)
    cmd_q, ack_q = mp.Queue(), mp.Queue()
    child = mp.Process(target=disturber_main, args=(cmd_q, ack_q), daemon=True)
    child.start()

    from bind_your_avatar_implementation_amd import ops
    dev = torch.device("cuda:0")
    hog = load_hog()
    noslp = ctypes.CDLL(NOSLP)
    _vp, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    noslp.bya_qknorm_rope.argtypes = [_vp] * 8 + [_i32, _i32, _i32, _i64, _i64, _i32, _f32, _f32, _vp]
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(1)
    S, D = 17776, 3072
    pris_f = torch.randn(S * D // 2, device=dev, generator=g)                   # fp32, 109 MB
    pris_q = torch.randn(1, S, D, device=dev, generator=g).to(torch.bfloat16)
    pris_k = torch.randn(1, S, D, device=dev, generator=g).to(torch.bfloat16)
    w64 = (1 + 0.1 * torch.randn(64, device=dev, generator=g)).to(torch.bfloat16)
    b64 = (0.05 * torch.randn(64, device=dev, generator=g)).to(torch.bfloat16)
    wD = (1 + 0.1 * torch.randn(D, device=dev, generator=g)).to(torch.bfloat16)
    bD = (0.05 * torch.randn(D, device=dev, generator=g)).to(torch.bfloat16)
    cos, sin = torch.randn(S - 226, 64, device=dev, generator=g), torch.randn(S - 226, 64, device=dev, generator=g)
    npairs = pris_f.numel() // 64
    table = torch.randn((npairs + 47) // 48, 64, device=dev, generator=g)          # one 256-byte row per 48 pairs
    print("victim: cos at", hex(cos.data_ptr()), "table at", hex(table.data_ptr()), "pris_q at", hex(pris_q.data_ptr()), flush=True)
    wf, wq, wk = torch.empty_like(pris_f), torch.empty_like(pris_q), torch.empty_like(pris_k)
    of, oq = torch.empty_like(pris_f), torch.empty_like(pris_q)

    def victim(name):
        """-> tuple of result tensors (views of the work buffers)"""
        if name == "rmw_inplace":
            wf.copy_(pris_f)
            assert hog.victim_launch(wf.data_ptr(), wf.data_ptr(), wf.numel() // 4, stream) == 0
            return (wf,)
        if name == "rmw_out":
            wf.copy_(pris_f)
            assert hog.victim_launch(wf.data_ptr(), of.data_ptr(), wf.numel() // 4, stream) == 0
            return (of,)
        if name == "qknorm_rope_inplace":
            wq.copy_(pris_q)
            wk.copy_(pris_k)
            ops.qknorm_rope(wq, wk, w64, b64, w64, b64, cos, sin, heads=48, text_rows=226, k_scale=0.18)
            return (wq, wk)
        if name == "qknorm_rope_noslp_inplace":          # same source, built without packed-fp32 (v_pk_*) instructions
            wq.copy_(pris_q)
            wk.copy_(pris_k)
            rc = noslp.bya_qknorm_rope(wq.data_ptr(), wk.data_ptr(), w64.data_ptr(), b64.data_ptr(), w64.data_ptr(),
                                       b64.data_ptr(), cos.data_ptr(), sin.data_ptr(), 1, S, 48, D, 0, 226,
                                       ctypes.c_float(1e-6), ctypes.c_float(0.18), stream)
            assert rc == 0, rc
            return (wq, wk)
        if name in ("qknorm_rope_dbg1_inplace", "qknorm_rope_dbg2_inplace"):
            # experiment builds of the kernel: 1 = table loads drained before anything else runs, 2 = 32-bit index maths
            os.environ["BYA_QKNORM_DBG"] = name[15]
            try:
                wq.copy_(pris_q)
                wk.copy_(pris_k)
                ops.qknorm_rope(wq, wk, w64, b64, w64, b64, cos, sin, heads=48, text_rows=226, k_scale=0.18)
            finally:
                os.environ.pop("BYA_QKNORM_DBG", None)
            return (wq, wk)
        if name == "qknorm_rope_sc1_inplace":           # same kernel, cos / sin read past the vector L1
            os.environ["BYA_QKNORM_TABLE_SC1"] = "1"
            try:
                wq.copy_(pris_q)
                wk.copy_(pris_k)
                ops.qknorm_rope(wq, wk, w64, b64, w64, b64, cos, sin, heads=48, text_rows=226, k_scale=0.18)
            finally:
                os.environ.pop("BYA_QKNORM_TABLE_SC1", None)
            return (wq, wk)
        if name in ("bcast_table", "bcast_table_sc1"):   # self-contained: the load pattern alone, out of place
            assert hog.bcast_launch(pris_f.data_ptr(), table.data_ptr(), of.data_ptr(), npairs, 48,
                                    int(name.endswith("sc1")), stream) == 0
            return (of[:npairs * 64],)
        if name == "qknorm_norope_inplace":             # same kernel, no cos / sin reads (every row is a "text" row)
            wq.copy_(pris_q)
            wk.copy_(pris_k)
            ops.qknorm_rope(wq, wk, w64, b64, w64, b64, None, None, heads=48, text_rows=S, k_scale=0.18)
            return (wq, wk)
        if name == "layernorm_out":
            wq.copy_(pris_q)
            ops.layernorm(wq, oq, wD, bD)
            return (oq,)
        if name == "layernorm_inplace":
            wq.copy_(pris_q)
            ops.layernorm(wq, wq, wD, bD)
            return (wq,)
        if name == "torch_mul_inplace":
            wf.copy_(pris_f)
            wf.mul_(1.5).add_(0.25)
            return (wf,)
        if name == "torch_layernorm_out":
            wq.copy_(pris_q)
            return (torch.nn.functional.layer_norm(wq, (D,), wD, bD),)
        raise KeyError(name)

    victims = args.victims.split(",")
    if "qknorm_rope_inplace" in victims and "qknorm_norope_inplace" not in victims:
        victims.append("qknorm_norope_inplace")
    golden = {}
    for v in victims:                      # disturber idle
        outs = victim(v)
        torch.cuda.synchronize()
        golden[v] = tuple(o.clone() for o in outs)
        for _ in range(5):                 # the victim alone must be deterministic
            outs = victim(v)
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(outs, golden[v])), f"{v} is not deterministic on its own"

    # what a corrupted (row, head) group of the in-place q/k-norm looks like: the untouched input (the write never
    # happened), the kernel applied TWICE (a replayed read-modify-write), or something else
    def classify_qk(outs):
        info = {"groups": 0, "equal_input": 0, "equal_applied_twice": 0, "equal_norm_without_rope": 0, "all_zero": 0,
                "other": 0, "examples": []}
        nr = golden.get("qknorm_norope_inplace")
        for name, o, g1, x, g2 in (("q", outs[0], golden["qknorm_rope_inplace"][0], pris_q, twice[0]),
                                   ("k", outs[1], golden["qknorm_rope_inplace"][1], pris_k, twice[1])):
            og, gg, xg, tg = (t.view(-1, 64) for t in (o, g1, x, g2))
            bad = (og != gg).any(dim=1).nonzero().flatten()
            if bad.numel() == 0:
                continue
            eq_in = (og[bad] == xg[bad]).all(dim=1)
            eq_tw = (og[bad] == tg[bad]).all(dim=1) & ~eq_in
            eq_nr = torch.zeros_like(eq_in)
            if nr is not None:          # the q/k LayerNorm was applied but the RoPE branch was not taken
                eq_nr = (og[bad] == nr[0 if name == "q" else 1].view(-1, 64)[bad]).all(dim=1) & ~eq_in & ~eq_tw
            info["groups"] += int(bad.numel())
            info["equal_input"] += int(eq_in.sum())
            info["equal_applied_twice"] += int(eq_tw.sum())
            info["equal_norm_without_rope"] += int(eq_nr.sum())
            info["all_zero"] += int((og[bad] == 0).all(dim=1).sum())
            info["other"] += int((~eq_in & ~eq_tw & ~eq_nr).sum())
            b = bad.cpu().tolist()
            runs, start = [], b[0]
            for a, c in zip(b, b[1:] + [None]):
                if c != a + 1:
                    runs.append((start, a - start + 1))
                    start = c
            info["examples"].append({"tensor": name, "bad_groups": len(b), "first_runs(group,len)": runs[:12],
                                     "rows": sorted({g // 48 for g in b})[:12]})
        return info

    twice = None
    if "qknorm_rope_inplace" in victims:
        wq.copy_(pris_q)
        wk.copy_(pris_k)
        for _ in range(2):
            ops.qknorm_rope(wq, wk, w64, b64, w64, b64, cos, sin, heads=48, text_rows=226, k_scale=0.18)
        torch.cuda.synchronize()
        twice = (wq.clone(), wk.clone())

    results = {}
    for d in args.disturbers.split(","):
        cmd_q.put(d)
        assert ack_q.get(timeout=120) == d
        row = {}
        for v in victims:
            bad_runs, bad_elems, detail = 0, 0, None
            t0 = time.time()
            for _ in range(args.runs):
                outs = victim(v)
                torch.cuda.synchronize()
                n = sum(int((a != b).sum()) for a, b in zip(outs, golden[v]))
                bad_runs += n > 0
                bad_elems += n
                if n > 0 and detail is None:
                    detail = classify_qk(outs) if v == "qknorm_rope_inplace" else {}
                    a, b = outs[0].reshape(-1), golden[v][0].reshape(-1)
                    idx = (a != b).nonzero().flatten()
                    detail["wrong_values_sample"] = [float(x) for x in a[idx[:16]].float().cpu()]
                    detail["golden_values_sample"] = [float(x) for x in b[idx[:16]].float().cpu()]
                    detail["max_abs_wrong"] = float(a[idx].float().abs().max())
                    detail["wrong_lane_groups_mod64"] = sorted({int(i) % 512 // 64 for i in idx[:4096].cpu()}) \
                        if v.startswith("bcast") else None
            row[v] = {"bad_runs": bad_runs, "runs": args.runs, "wrong_elements": bad_elems,
                      "ms_per_run": round((time.time() - t0) / args.runs * 1e3, 2)}
            if detail is not None:
                row[v]["first_bad_run"] = detail
                print("   ", json.dumps(detail)[:600], flush=True)
        results[d] = row
        print(f"{d:20s} " + "  ".join(f"{v}:{row[v]['bad_runs']}/{args.runs}" for v in victims), flush=True)
    cmd_q.put("quit")
    try:
        ack_q.get(timeout=60)
    except Exception:
        pass
    child.join(timeout=30)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump({"runs": args.runs, "table": results,
                   "note": "bad_runs = victim runs whose output differed bit-wise from the run with an idle disturber; "
                           "two processes share cuda:0"}, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
