#!/usr/bin/env python3
"""What bounds the joint attention: the product kernel against ABLATION builds of the same source (csrc/attn.hip with
-DBYA_ATTN_ABLATE=<mask>: pieces of the hot loop left out, results meaningless, time and clock real) on normalised gaussian
q / k / v, with board power and shader clock sampled while each arm loops.

  mask bits: 1 v_exp -> one FMA | 2 no K fragment reads from LDS | 4 half of the V fragment reads | 8 no K/V staging after
             the first tile (no global / LDS-DMA traffic) | 16 no row-sum adds
  schedule variants (same work, must give the same bits as the product build, mask 0 = two 32-row query blocks per wave
             sharing every K / V fragment, 256 rows per workgroup, two waves per SIMD, three K/V stages with the per-tile
             rendezvous behind the QK^T MFMAs): one_block = the kernel of round 1 (32 rows per wave, four waves per SIMD),
             one_block_kprefetch / _occ3 / _ring3 = its variants (K reads up front, three waves per SIMD, three stages),
             qb2_ring2 / qb2_ring4 = the product kernel with two / four stages, qb2_kprefetch = with the next tile's K fragments
             requested right behind the rendezvous
  (An ablation must not let the compiler drop MFMAs: the first version of mask 2 fed both score chains the same operands,
  hipcc merged them, and the "gain" was a quarter of the matrix work missing -- check the instruction counts in the ISA.)

Build (no GPU needed):  python tools/attn_ablate.py --build        -> tools/ablate/libattn_<mask>.so
Run on the GPU box:     python tools/attn_ablate.py [--out gpurun_out/attn_ablate.json]
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
MASKS = [0, 1, 16, 17, 2, 4, 6, 8, 14, 31]
# schedule variants of the SAME work (results must be bit-identical to the product build): name -> extra flags
ONE = ["-DBYA_ATTN_QB2=0", "-DBYA_ATTN_QB2_RING3=0"]          # the one-query-block kernel (round 1 .. mid round 2)
VARIANTS = {"one_block": ONE, "one_block_kprefetch": ONE + ["-DBYA_ATTN_KPREFETCH=1"], "one_block_occ3": ONE + ["-DBYA_ATTN_OCC=3"],
            "one_block_ring3": ONE + ["-DBYA_ATTN_RING=3", "-DBYA_ATTN_OCC=3"],
            "qb2_ring2": ["-DBYA_ATTN_QB2=1", "-DBYA_ATTN_QB2_RING3=0"],
            "qb2_ring4": ["-DBYA_ATTN_QB2=1", "-DBYA_ATTN_QB2_RING3=4"], "qb2_kprefetch": ["-DBYA_ATTN_QB2_KPF=1"]}
S, H, D = 17776, 48, 64


def build():
    src = os.path.join(ROOT, "bind_your_avatar_implementation_amd", "csrc", "attn.hip")
    os.makedirs(os.path.join(HERE, "ablate"), exist_ok=True)
    for m in MASKS:
        out = os.path.join(HERE, "ablate", f"libattn_{m}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
               "-amdgpu-mfma-vgpr-form=1", f"-DBYA_ATTN_ABLATE={m}", "-I" + os.path.join(ROOT, "include"), src, "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        print("built", out)
    for name, flags in VARIANTS.items():
        out = os.path.join(HERE, "ablate", f"libattn_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
               "-amdgpu-mfma-vgpr-form=1", *flags, "-I" + os.path.join(ROOT, "include"), src, "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        print("built", out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--variants-only", action="store_true", help="product build + the schedule variants, no ablations")
    a = ap.parse_args()
    if a.build:
        return build()
    import torch
    from gemm_probe import Smi, median, time_once
    from bind_your_avatar_implementation_amd import _hip
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    nrm = lambda t: (t / t.view(1, S, H, D).float().norm(dim=-1, keepdim=True).repeat_interleave(D, -1).view(1, S, H * D) * 8).to(torch.bfloat16)
    q = nrm(torch.randn(1, S, H * D, device=dev, generator=g))
    k = (nrm(torch.randn(1, S, H * D, device=dev, generator=g)).float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
    v = torch.randn(1, S, H * D, device=dev, generator=g).to(torch.bfloat16)
    o = torch.empty_like(q)
    d = _hip.AttnDesc()
    d.head_dim, d.heads, d.nb1, d.nb2, d.Sq, d.Skv = D, H, 1, 1, S, S
    for n in ("q", "k", "v", "o"):
        setattr(d, n + "_s1", 0); setattr(d, n + "_s2", 0); setattr(d, n + "_row", H * D)
    d.scale, d.scores_prescaled, d.score_bound = 0.125, 1, 11.8
    flop = 4.0 * S * S * D * H
    res = {}
    ref_out = None
    for m in ([0] + list(VARIANTS) + [0] + list(VARIANTS) if a.variants_only else MASKS + list(VARIANTS)):
        lib = ctypes.CDLL(os.path.join(HERE, "ablate", f"libattn_{m}.so"))
        lib.bya_attn_fwd.restype = ctypes.c_int32
        lib.bya_attn_fwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.POINTER(_hip.AttnDesc), ctypes.c_void_p]
        st = torch.cuda.current_stream().cuda_stream
        fn = lambda: lib.bya_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), ctypes.byref(d), st)
        assert fn() == 0
        torch.cuda.synchronize()
        if m == 0 and ref_out is None:
            ref_out = o.clone()
        same = bool(torch.equal(o, ref_out)) if (m == 0 or m in VARIANTS) else None
        t1 = time_once(fn, 3)
        iters = max(5, int(a.seconds / t1))
        smi = Smi(); smi.start()
        ts = [time_once(fn, iters // 3 + 1) for _ in range(3)]
        smi.stop_flag = True; smi.join()
        t, sm = median(ts), smi.summary()
        key = str(m) if str(m) not in res else str(m) + "#2"
        res[key] = dict(mask=m, ms=t * 1e3, equivalent_tflops=flop / t / 1e12, same_bits_as_product=same, **sm)
        print(f"{str(m):>15s}: {t * 1e3:7.3f} ms  {flop / t / 1e12:7.0f} 'TFLOP/s'  same bits: {same}  {sm}")
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
