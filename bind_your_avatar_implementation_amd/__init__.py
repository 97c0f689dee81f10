"""MI355X-native (gfx950) denoise-step engine for Bind-Your-Avatar.

Drop-in for the reference's per-step hot path: ``BindyouravatarTransformer3DModel.forward``
(reference models/transformer.py:615-964) behind the same constructor / state-dict / call surface, with the
step itself running on hand-written HIP kernels loaded from ``libbya_hip.so`` (C ABI: ``include/bya.h``).
"""
__all__ = ["BindyouravatarTransformer3DModel", "BindyouravatarPipeline", "BindyouravatarVAE", "ops", "build"]


def __getattr__(name):  # lazy: importing the package must not need torch.cuda or the built library
    if name == "BindyouravatarTransformer3DModel":
        from .transformer import BindyouravatarTransformer3DModel
        return BindyouravatarTransformer3DModel
    if name == "BindyouravatarPipeline":
        from .pipeline import BindyouravatarPipeline
        return BindyouravatarPipeline
    if name == "BindyouravatarVAE":
        from .vae import BindyouravatarVAE
        return BindyouravatarVAE
    raise AttributeError(name)
