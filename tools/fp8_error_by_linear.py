"""Which Linears carry the e4m3 error?  The 42-layer model of tests/test_forward_gpu.py::test_depth_42_layers_vs_golden (reference
geometry, name-keyed synthetic weights, golden = the reference in fp32) run with ONE kind of Linear in e4m3 at a time, with
all of them, and with all but one: output and mid-depth error against the fp32 reference next to the bf16 engine's.
usage: python tools/fp8_error_by_linear.py [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel
from bind_your_avatar_implementation_amd.engine import FP8_LINEARS
from bind_your_avatar_implementation_amd.synth import synth_inputs

GOLD = os.path.join(ROOT, "tests", "golden")
dev = torch.device("cuda:0")
fx = np.load(os.path.join(GOLD, "ref_forward_depth_L42_seed0.npz"))
meta = json.load(open(os.path.join(GOLD, "ref_state_dict_keys.json")))
model = BindyouravatarTransformer3DModel(num_layers=42, **meta["model_kw"], device=dev)
model.init_synthetic(seed=0, fast=False)
def to_dev(inp, dtype=torch.bfloat16):
    def cv(t):
        if torch.is_tensor(t):
            return t.to(dev, dtype) if t.dtype.is_floating_point else t.to(dev)
        if isinstance(t, (list, tuple)):
            return type(t)(cv(u) for u in t)
        return t
    out = {k: cv(v) for k, v in inp.items()}
    out["image_rotary_emb"] = tuple(t.to(dev, torch.float32) for t in inp["image_rotary_emb"])
    return out


gi = to_dev(synth_inputs(batch=1, seed=0))
ref = torch.from_numpy(fx["output_f16"].astype(np.float32))
step = int(fx["tap_step"])
rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())


def run(linears):
    if linears is None:
        model.enable_fp8_weights(False)
    elif linears == "default":
        model.enable_fp8_weights(True)
    else:
        model.enable_fp8_weights(True, linears=linears)
    model(**gi)
    taps = {}
    e = model._engine
    out = e.step(gi["hidden_states"], gi["encoder_hidden_states"], gi["timestep"], gi["image_rotary_emb"], gi["id_cond"],
                 gi["id_vit_hidden"], gi["audio_embeds"], gi["af_matrix"], gi.get("routing_logits_forcing"), taps=taps)
    mid = {i: rel(taps[f"block{i}"][:, 226:].reshape(-1)[::step], torch.from_numpy(fx[f"block{i}.strided"])) for i in (11, 23, 41)}
    return {"output": rel(out, ref), **{f"block{i}": v for i, v in mid.items()}}


result = {"bf16_oracle_output": float(fx["bf16_err_output"]), "cases": {}}
cases = [("bf16 engine", None), ("all six", FP8_LINEARS), ("default", None if False else "default"), ("dit four + aq", ("qkv", "out", "ff1", "ff2", "aq"))] + [(f"only {k}", (k,)) for k in FP8_LINEARS] + \
        [(f"all but {k}", tuple(x for x in FP8_LINEARS if x != k)) for k in ("qkv", "out", "ff1", "ff2")]
for name, lin in cases:
    r = run(lin)
    result["cases"][name] = r
    print(f"{name:14s} " + "  ".join(f"{k} {v:.3e}" for k, v in r.items()), flush=True)
base = result["cases"]["bf16 engine"]["output"]
for name, r in result["cases"].items():
    r["output_vs_bf16_engine"] = r["output"] / base
if len(sys.argv) > 1:
    json.dump(result, open(sys.argv[1], "w"), indent=1)
