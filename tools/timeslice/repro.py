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


def build():
    src = os.path.join(HERE, "lds_hog.hip")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", LIB])
    return LIB


def load_hog():
    import torch  # noqa: F401  (one HIP runtime per process: torch first)
    lib = ctypes.CDLL(LIB)
    lib.hog_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.victim_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
    return lib


DISTURBERS = ["none", "hog64", "hog96", "hog128", "hog132", "hog136", "hog144", "hog160", "rowgemm_n512_132k",
              "rowgemm_n1536_140k", "gemm256_128k", "attn_32k", "torch_matmul"]
VICTIMS = ["rmw_inplace", "rmw_out", "qknorm_rope_inplace", "layernorm_out", "layernorm_inplace", "torch_mul_inplace",
           "torch_layernorm_out"]


def disturber_main(cmd_q, ack_q):
    import torch
    from bind_your_avatar_implementation_amd import ops
    dev = torch.device("cuda:0")
    hog = load_hog()
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    bf = lambda *s: (torch.randn(*s, device=dev)).to(torch.bfloat16)
    x512, o1536, o512 = bf(35100, 512), torch.empty(35100, 1536, dtype=torch.bfloat16, device=dev), bf(35100, 512)
    pk1536 = ops.pack_rowgemm512(bf(1536, 512) * 0.04, bf(1536), bf(512), bf(512))
    pk512 = ops.pack_rowgemm512(bf(512, 512) * 0.04, bf(512), bf(512), bf(512))
    ga, gw, go = bf(8192, 3072), bf(3072, 3072) * 0.02, torch.empty(8192, 3072, dtype=torch.bfloat16, device=dev)
    aq, ak, av = bf(1, 4096, 3072), bf(1, 4096, 3072), bf(1, 4096, 3072)
    ao = torch.empty_like(aq)
    stream = torch.cuda.current_stream().cuda_stream

    def once(name):
        if name.startswith("hog"):
            assert hog.hog_launch(sink.data_ptr(), 256, 40, int(name[3:]) * 1024, stream) == 0
        elif name == "rowgemm_n1536_140k":
            ops.rowgemm512(x512, pk1536, o1536)
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
    golden = {}
    for v in victims:                      # disturber idle
        outs = victim(v)
        torch.cuda.synchronize()
        golden[v] = tuple(o.clone() for o in outs)
        for _ in range(5):                 # the victim alone must be deterministic
            outs = victim(v)
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(outs, golden[v])), f"{v} is not deterministic on its own"

    table = {}
    for d in args.disturbers.split(","):
        cmd_q.put(d)
        assert ack_q.get(timeout=120) == d
        row = {}
        for v in victims:
            bad_runs, bad_elems = 0, 0
            t0 = time.time()
            for _ in range(args.runs):
                outs = victim(v)
                torch.cuda.synchronize()
                n = sum(int((a != b).sum()) for a, b in zip(outs, golden[v]))
                bad_runs += n > 0
                bad_elems += n
            row[v] = {"bad_runs": bad_runs, "runs": args.runs, "wrong_elements": bad_elems,
                      "ms_per_run": round((time.time() - t0) / args.runs * 1e3, 2)}
        table[d] = row
        print(f"{d:20s} " + "  ".join(f"{v}:{row[v]['bad_runs']}/{args.runs}" for v in victims), flush=True)
    cmd_q.put("quit")
    try:
        ack_q.get(timeout=60)
    except Exception:
        pass
    child.join(timeout=30)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump({"runs": args.runs, "table": table,
                   "note": "bad_runs = victim runs whose output differed bit-wise from the run with an idle disturber; "
                           "two processes share cuda:0"}, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
