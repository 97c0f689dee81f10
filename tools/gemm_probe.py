#!/usr/bin/env python3
"""GEMM evidence probe (profiles/README.md "GEMM probe"): bya_gemm_bf16 variants next to hipBLASLt
(torch.nn.functional.linear, reference point only -- never used by the engine) on the DiT shapes of
BASELINE configs[1], INTERLEAVED in one process (cdna_hip_programming.md section 5.4 rule 24: different boxes
differ by up to 12 %, only same-process A/B counts), on gaussian AND on all-zero operands (rule 25 / DVFS
give-back: zeros run at a higher clock), with board power / clock sampled from rocm-smi while each arm loops.

  python tools/gemm_probe.py [--quick] [--out gpurun_out/gemm_probe.json] [--variants default,w4,...]
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from bind_your_avatar_implementation_amd import ops
from bind_your_avatar_implementation_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")
S = 17776
SHAPES = {            # name: (M, N, K, epilogue kwargs of the engine's call site)
    "qkv": (S, 9216, 3072, dict(split=True)),
    "ff1": (S, 12288, 3072, dict(act="gelu_tanh")),
    "ff2": (S, 3072, 12288, dict(gate_res=True)),
    "attn_out": (S, 3072, 3072, dict(gate_res=True)),
    "audio_q": (17550, 3072, 3072, dict()),
    "perc_q": (17550, 2048, 3072, dict(nobias=True)),
    "sq8192": (8192, 8192, 8192, dict()),
    # the Embedding Router's 512-wide Linears (35100 rows = 2 ids x 17550 tokens): today's row-stationary kernel
    # (bya_rowgemm512) against the persistent tile kernel forced onto the same shape (BYA_GEMM_TILE=4)
    "router_qkv": (35100, 1536, 512, dict(router=True)),
    "router_out": (35100, 512, 512, dict(router=True, res=True)),
}


class Smi(threading.Thread):
    """Samples `rocm-smi --showpower --showclocks --json` (whatever the box reports) while a timing loop runs."""

    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop_flag = [], False

    def run(self):
        while not self.stop_flag:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True,
                                     text=True, timeout=5).stdout
                self.samples.append(json.loads(out))
            except Exception as e:                    # noqa: BLE001  (evidence only: record and go on)
                self.samples.append({"error": str(e)[:80]})
            time.sleep(0.05)

    def summary(self):
        pw, sclk = [], []
        for s in self.samples:
            for card in (s.values() if isinstance(s, dict) else []):
                if not isinstance(card, dict):
                    continue
                for k, v in card.items():
                    kl = k.lower()
                    try:
                        if "power" in kl and "w" in kl:
                            pw.append(float(str(v).split()[0]))
                        elif "sclk" in kl and "mhz" in str(v).lower():
                            sclk.append(float(str(v).lower().replace("(", "").replace("mhz)", "").replace("mhz", "").split()[-1]))
                    except Exception:                  # noqa: BLE001
                        pass
        avg = lambda a: round(sum(a) / len(a), 1) if a else None
        return {"n": len(self.samples), "power_w_avg": avg(pw), "power_w_max": max(pw) if pw else None,
                "sclk_mhz_avg": avg(sclk), "sclk_mhz_min": min(sclk) if sclk else None}


RAW = {}


def median(a):
    a = sorted(a)
    return a[len(a) // 2]


def time_once(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def make_case(name, zeros):
    M, N, K, ep = SHAPES[name]
    mk = (lambda *s, std=1.0: torch.zeros(*s, dtype=torch.bfloat16, device=dev)) if zeros else \
        (lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(torch.bfloat16))
    a, w, b = mk(M, K), mk(N, K, std=K ** -0.5), mk(N)
    kw = {}
    if ep.get("split"):
        out = torch.empty(3, M, N // 3, dtype=torch.bfloat16, device=dev)
        kw = dict(split=(N // 3, M * (N // 3)))
        target = out[0]
    else:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        target = out
    if ep.get("act"):
        kw["act"] = ep["act"]
    if ep.get("gate_res"):
        gate = mk(1, 2 * N)
        res = mk(M, N)
        kw.update(res=res, gate0=gate[:, :N], gate1=gate[:, N:], gate_split=226, gate_batch_stride=2 * N)
    bias = None if ep.get("nobias") else b
    return M, N, K, a, w, bias, target, kw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="one round, few iterations (for rocprofv3 passes)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--variants", default="v4,w8")
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--data", default="gaussian,zeros")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--fp8", action="store_true", help="add the e4m3 GEMM (bya_gemm_fp8), alone and with its row quantiser")
    args = ap.parse_args()
    variants = args.variants.split(",")
    rounds, iters = (1, 3) if args.quick else (args.rounds, 10)
    res = {}
    for data in args.data.split(","):
        for name in args.shapes.split(","):
            M, N, K, a, w, bias, out, kw = make_case(name, data == "zeros")
            fl = 2.0 * M * N * K / 1e12
            arms = {"vendor_plain": lambda: F.linear(a, w, bias)}
            if SHAPES[name][3].get("router"):
                res_t = out if SHAPES[name][3].get("res") else None
                pack = ops.pack_rowgemm512(w, bias)

                def rowgemm():
                    ops.rowgemm512(a, pack, out, res=res_t)

                def tiled(variant="v4"):
                    os.environ["BYA_GEMM_TILE"], os.environ["BYA_GEMM_VARIANT"] = "4", variant
                    _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
                    try:
                        ops.gemm(a, w, out, bias=bias, res=res_t)
                    finally:
                        os.environ.pop("BYA_GEMM_TILE", None)
                        _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
                arms["bya_rowgemm512"] = rowgemm
                arms["bya_v4_forced_big_tiles"] = tiled
                kw = None
            for v in (variants if kw is not None else []):
                def run(v=v, kw=kw):
                    os.environ["BYA_GEMM_VARIANT"] = v
                    _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
                    ops.gemm(a, w, out, bias=bias, **kw)
                arms[f"bya_{v}"] = run
                if kw and v == variants[0]:
                    def plain(v=v):
                        os.environ["BYA_GEMM_VARIANT"] = v
                        _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
                        ops.gemm(a, w, out if out.shape[-1] == N else torch.empty(M, N, dtype=torch.bfloat16, device=dev),
                                 bias=bias)
                    if "split" not in kw:
                        arms[f"bya_{v}_plain_epilogue"] = plain
                    if kw.get("act") == "gelu_tanh":
                        def ieee(v=v):
                            os.environ["BYA_GEMM_VARIANT"] = v
                            _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
                            ops.gemm(a, w, out, bias=bias, act="gelu_tanh_ieee")
                        arms[f"bya_{v}_gelu_ieee_div"] = ieee
            if kw is not None and args.fp8 and K % 128 == 0:
                a8, sa = ops.quantize_rows_fp8(a)
                w8, sw = ops.quantize_rows_fp8(w)
                kw8 = {k: v for k, v in kw.items()}

                def fp8_gemm(kw8=kw8):
                    ops.gemm_fp8(a8, sa, w8, sw, out, bias=bias, **kw8)

                def fp8_with_quant(kw8=kw8):
                    ops.quantize_rows_fp8(a, q=a8, scale=sa)
                    ops.gemm_fp8(a8, sa, w8, sw, out, bias=bias, **kw8)
                arms["bya_fp8_gemm_only"] = fp8_gemm
                arms["bya_fp8_quantise+gemm"] = fp8_with_quant
            times = {k: [] for k in arms}
            est = {k: time_once(fn, 2) for k, fn in arms.items()}               # warm-up + rough duration
            smi = {}
            for r in range(rounds):
                for k, fn in arms.items():
                    mon = Smi() if (r == 0 and not args.quick) else None
                    if mon:
                        mon.start()
                        time_once(fn, max(20, int(1.5 / est[k])))       # ~1.5 s under the sampler
                        mon.stop_flag = True
                        mon.join()
                        smi[k] = mon.summary()
                        RAW.setdefault("first_sample", mon.samples[0] if mon.samples else None)
                    times[k].append(time_once(fn, iters))
            row = {k: {"ms_median": round(median(t) * 1e3, 4), "ms_min": round(min(t) * 1e3, 4),
                       "tflops_median": round(fl / median(t), 1), "tflops_best": round(fl / min(t), 1),
                       "smi": smi.get(k)} for k, t in times.items()}
            res[f"{data}:{name}"] = {"M": M, "N": N, "K": K, "arms": row}
            print(f"{data:8s} {name:9s} " + "  ".join(f"{k}={v['tflops_median']:.0f}" for k, v in row.items()), flush=True)
            del a, w, out
            torch.cuda.empty_cache()
    os.environ.pop("BYA_GEMM_VARIANT", None)
    _hip.apply_env_options()      # (the library reads no environment: hand the change to its option table)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump({"device": torch.cuda.get_device_name(0), "rounds": rounds, "iters": iters,
                       "smi_raw_first_sample": RAW.get("first_sample"),
                       "note": "TFLOP/s = 2MNK / time; vendor_plain = torch F.linear (hipBLASLt) with bias, no fused "
                               "epilogue; bya_* = bya_gemm_bf16 with the engine's epilogue for that call site; "
                               "*_plain_epilogue = same kernel, bias only", "results": res}, f, indent=1)


if __name__ == "__main__":
    main()
