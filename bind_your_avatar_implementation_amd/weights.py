"""Checkpoint ingestion for the MI355X engine: the on-disk formats either side of
``BindyouravatarTransformer3DModel`` as the reference reads them.

* ``from_pretrained_cus`` -- reference models/transformer.py:1024-1093: ``config.json`` + ``diffusion_pytorch_model``
  ``.bin`` / ``.safetensors`` / any number of ``*.safetensors`` shards, loaded NON-strictly (shape mismatches are skipped
  with a message), ``patch_embed.proj.weight`` zero-padded (or cut) along its input channels when the checkpoint was
  trained with fewer (more) latent channels.  Shards are streamed tensor by tensor straight into the parameters' HBM
  (``safetensors.safe_open``): the 17 GB state dict is never materialised on the host.
* LoRA -- reference util/utils.py:1027-1048 + infer.py:279: rank-128 adapters on ``attn1.to_q`` / ``attn1.to_k``
  (``lora_alpha`` 128), folded with ``lora_scale = 1 / rank``:  W <- W + lora_scale * (alpha / r) * B @ A.  The fold is
  done once at load time in fp32 on the GPU and rounded to bf16 once; the engine never sees adapters.
"""
import glob
import json
import os
import re

import torch

WEIGHTS_NAME = "diffusion_pytorch_model.bin"          # diffusers.utils.WEIGHTS_NAME
_CONFIG_SKIP = ("_class_name", "_diffusers_version", "_name_or_path")


def _iter_checkpoint(path):
    """Yield (key, tensor-loader) for every tensor of the checkpoint directory, in the reference's precedence:
    the single ``.bin``, else the single ``.safetensors``, else every ``*.safetensors`` shard."""
    model_file = os.path.join(path, WEIGHTS_NAME)
    st = model_file.replace(".bin", ".safetensors")
    if os.path.exists(model_file):
        sd = torch.load(model_file, map_location="cpu")
        for k, v in sd.items():
            yield k, (lambda v=v, **kw: v)
        return
    files = [st] if os.path.exists(st) else sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise RuntimeError(f"no diffusion_pytorch_model.bin / *.safetensors under '{path}'")
    from safetensors import safe_open
    for f in files:
        with safe_open(f, framework="pt", device="cpu") as fh:
            for k in fh.keys():
                yield k, (lambda k=k, fh=fh, **kw: fh.get_tensor(k))


def _fit_patch_proj(dst, src):
    """reference :1060-1071 (4-D conv weight): zero-pad or cut the input-channel axis to the model's."""
    out = torch.zeros(dst.shape, dtype=src.dtype)
    c = min(dst.shape[1], src.shape[1])
    out[:, :c] = src[:, :c]
    return out


def load_checkpoint_dir(model, path, verbose=True):
    """Non-strict load of a checkpoint directory into ``model`` (any device).  -> (missing, unexpected, skipped)."""
    own = model.state_dict()
    seen, unexpected, skipped = set(), [], []
    with torch.no_grad():
        for key, get in _iter_checkpoint(path):
            if key not in own:
                unexpected.append(key)
                continue
            t = get()
            if key == "patch_embed.proj.weight" and t.dim() == 4 and tuple(t.shape) != tuple(own[key].shape):
                t = _fit_patch_proj(own[key], t)
            if tuple(t.shape) != tuple(own[key].shape):
                skipped.append(key)
                if verbose:
                    print(key, "Size don't match, skip")
                continue
            own[key].copy_(t.to(own[key].dtype))            # host -> HBM, one tensor at a time
            seen.add(key)
    missing = [k for k in own if k not in seen]
    if verbose:
        print(f"### missing keys: {len(missing)}; \n### unexpected keys: {len(unexpected)};")
    model.invalidate_engine()
    return missing, unexpected, skipped


def from_pretrained_cus(cls, pretrained_model_path, subfolder=None, config_path=None, transformer_additional_kwargs={},
                        device=None, dtype=torch.bfloat16, verbose=True):
    """Same call as the reference classmethod (+ ``device`` / ``dtype``)."""
    if subfolder:
        config_path = config_path or pretrained_model_path
        config_file = os.path.join(config_path, subfolder, "config.json")
        pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
    else:
        config_file = os.path.join(config_path or pretrained_model_path, "config.json")
    if not os.path.isfile(config_file):
        raise RuntimeError(f"Configuration file '{config_file}' does not exist")
    with open(config_file, "r") as f:
        config = json.load(f)
    import inspect
    accepted = set(inspect.signature(cls.__init__).parameters)
    kw = {k: v for k, v in config.items() if k not in _CONFIG_SKIP and k in accepted}
    kw.update(transformer_additional_kwargs)
    model = cls(**kw, device=device, dtype=dtype)
    load_checkpoint_dir(model, pretrained_model_path, verbose)
    return model


# ---- LoRA ----------------------------------------------------------------------------------------------------------
_LORA_KEY = re.compile(r"^(?:base_model\.model\.|transformer\.module\.|transformer\.)?(.*)\.lora_([AB])(?:\.default)?\.weight$")


def read_lora(path_or_state):
    """-> {module path: {"A": [r, in], "B": [out, r]}} from a LoRA safetensors file / state dict with the key
    spellings the reference normalises (``transformer.module.`` / ``transformer.`` / ``base_model.model.`` prefixes,
    ``lora_A[.default].weight``)."""
    if isinstance(path_or_state, (str, os.PathLike)):
        from safetensors.torch import load_file
        state = load_file(path_or_state)
    else:
        state = path_or_state
    out = {}
    for k, v in state.items():
        m = _LORA_KEY.match(k)
        if m:
            out.setdefault(m.group(1), {})[m.group(2)] = v
    return out


def fold_lora(model, lora, lora_scale, lora_alpha=128, target_modules=("attn1.to_q", "attn1.to_k")):
    """W <- W + lora_scale * (lora_alpha / r) * B @ A on every targeted Linear (reference: LoraConfig(r, lora_alpha=128,
    target_modules=[attn1.to_k, attn1.to_q]) then ``pipe.fuse_lora(lora_scale=1/r)``).  fp32 on the weight's device,
    one rounding to the weight dtype.  Returns the number of folded modules."""
    mods = dict(model.named_modules())
    n = 0
    with torch.no_grad():
        for name, ab in lora.items():
            if not any(name.endswith(t) for t in target_modules):
                continue
            if name not in mods or "A" not in ab or "B" not in ab:
                raise KeyError(f"LoRA factors for '{name}' do not match a module of the model")
            w = mods[name].weight
            A, B = ab["A"].to(w.device, torch.float32), ab["B"].to(w.device, torch.float32)
            r = A.shape[0]
            w.copy_((w.float() + (lora_scale * lora_alpha / r) * (B @ A)).to(w.dtype))
            n += 1
    model.invalidate_engine()
    return n
