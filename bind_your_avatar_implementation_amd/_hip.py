"""ctypes binding of ``libbya_hip.so`` (C ABI declared in ``include/bya.h``).

Loading fails LOUDLY: there is no CPU / PyTorch fallback for the product path.
"""
import ctypes
import os

from .build import LIB_PATH as _BUILT_LIB

# tools/ A/B runs may point at a side build of the same library (never set in production)
LIB_PATH = os.environ.get("BYA_HIP_LIB") or _BUILT_LIB

_c = ctypes
_vp, _i32, _i64, _f32 = _c.c_void_p, _c.c_int32, _c.c_int64, _c.c_float


class GemmDesc(_c.Structure):
    _fields_ = [("M", _i32), ("N", _i32), ("K", _i32), ("batch", _i32),
                ("lda", _i32), ("ldw", _i32), ("ldc", _i32), ("ldres", _i32),
                ("a_batch_stride", _i64), ("c_batch_stride", _i64), ("res_batch_stride", _i64),
                ("gate_batch_stride", _i64), ("gate_split", _i32), ("act", _i32),
                ("n_split", _i32), ("c_split_stride", _i64), ("bias_rowscale", _vp), ("alpha", _f32)]


class QkNormDesc(_c.Structure):
    _fields_ = [("qw", _vp), ("qb", _vp), ("kw", _vp), ("kb", _vp), ("cos", _vp), ("sin", _vp),
                ("text_rows", _i32), ("width", _i32), ("eps", _f32), ("k_scale", _f32)]


class AttnDesc(_c.Structure):
    _fields_ = [("head_dim", _i32), ("heads", _i32), ("nb1", _i32), ("nb2", _i32), ("Sq", _i32), ("Skv", _i32),
                ("q_s1", _i64), ("q_s2", _i64), ("q_row", _i64),
                ("k_s1", _i64), ("k_s2", _i64), ("k_row", _i64),
                ("v_s1", _i64), ("v_s2", _i64), ("v_row", _i64),
                ("o_s1", _i64), ("o_s2", _i64), ("o_row", _i64),
                ("scale", _f32), ("scores_prescaled", _i32), ("score_bound", _f32),
                ("bound_dev", _vp), ("bound_slots", _i32), ("bound_heads", _i32), ("bound_bh0", _i32), ("fallback_flags", _vp)]


class AttnMixDesc(_c.Structure):
    _fields_ = [("head_dim", _i32), ("heads", _i32), ("n_id", _i32), ("n_grp", _i32), ("Sq", _i32), ("Skv", _i32),
                ("q_grp", _i64), ("q_row", _i64), ("k_id", _i64), ("k_grp", _i64), ("k_row", _i64),
                ("v_id", _i64), ("v_grp", _i64), ("v_row", _i64), ("z_grp", _i64), ("z_row", _i64), ("scale", _f32)]


class SchedCoef(_c.Structure):
    _fields_ = [("guidance", _f32), ("sqrt_alpha", _f32), ("sqrt_beta", _f32), ("k_sample", _f32),
                ("k_denoised", _f32), ("k_noise", _f32), ("k_cur", _f32), ("k_old", _f32)]


# name -> argtypes (all return int32 status); mirrors include/bya.h one-to-one
SIGNATURES = {
    "bya_abi_version": [],
    "bya_set_option": [_i32, _i32],
    "bya_get_option": [_i32, _c.POINTER(_i32)],
    "bya_mfma_calibration": [_vp, _i64, _vp, _i32, _vp],
    "bya_gemm_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.POINTER(GemmDesc), _vp],
    "bya_gemm_skinny_bf16": [_vp, _vp, _vp, _vp, _vp, _c.POINTER(GemmDesc), _vp],
    "bya_gemm_qkv_norm_rope": [_vp, _vp, _vp, _vp, _c.POINTER(GemmDesc), _c.POINTER(QkNormDesc), _vp],
    "bya_set_gemm_workspace": [_vp, _i64],
    "bya_gemm_workspace_bytes": [_c.POINTER(_i64)],
    "bya_gemm_workspace_status": [_c.POINTER(_i32), _vp],
    "bya_quantize_rows_fp8": [_vp, _vp, _vp, _i32, _i32, _i64, _i64, _vp],
    "bya_gemm_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.POINTER(GemmDesc), _vp],
    "bya_layernorm_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64,
                          _f32, _vp],
    "bya_linear_small_m": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "bya_timestep_features": [_vp, _vp, _i32, _i32, _i32, _f32, _vp],
    "bya_layernorm": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64,
                      _f32, _vp],
    "bya_qknorm_rope": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i32, _f32, _f32, _vp, _i32, _vp],
    "bya_attn_fwd": [_vp, _vp, _vp, _vp, _c.POINTER(AttnDesc), _vp],
    "bya_attn_variant": [_c.POINTER(AttnDesc)],
    "bya_set_attn_workspace": [_vp, _i64],
    "bya_attn_workspace_bytes": [_c.POINTER(_i64)],
    "bya_attn_workspace_status": [_c.POINTER(_i32), _vp],
    "bya_attn_kv_mix": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.POINTER(AttnMixDesc), _vp],
    "bya_attn_tiny": [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp],
    "bya_router_scores": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _f32, _vp],
    "bya_router_head": [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp],
    "bya_forcing_max_over_frames": [_vp, _vp, _i32, _i64, _i32, _vp],
    "bya_masked_combine": [_vp, _vp, _vp, _vp, _i32, _f32, _i32, _i32, _i64, _i32, _i64, _i64, _i64, _vp],
    "bya_routed_mix": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _i64, _vp],
    "bya_patchify": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "bya_unpatchify": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "bya_act_add": [_vp, _vp, _vp, _i64, _i32, _vp],
    "bya_rowgemm512": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp],
    "bya_router_group_attn": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f32, _f32, _vp],
    "bya_router_mlp_fused": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp],
    "bya_router_group_attn_out": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f32, _f32,
                                  _i32, _vp],
    "bya_masks_to_routing_logits": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "bya_vae_patches": [_vp, _vp, _vp] + [_i32] * 14 + [_vp],
    "bya_vae_groupnorm_stats": [_vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "bya_vae_norm_act": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _i32, _i32, _i32, _i32, _i32,
                         _i32, _i64, _i32, _vp],
    "bya_vae_conv3d": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _vp],
    "bya_vae_upsample_pad": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "bya_allgather_kv": [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp],
    "bya_alltoall_router": [_vp, _vp, _vp, _vp, _i32, _vp, _vp],
    "bya_p2p_push": [_vp, _i32, _i64, _vp, _i32, _i32, _vp, _vp],
    "bya_p2p_wait": [_vp, _i32, _vp, _i64, _vp],
    "bya_p2p_exchange": [_vp, _i32, _i64, _vp, _i32, _i32, _vp, _vp, _i64, _vp],
    "bya_p2p_poison": [_vp, _i32, _vp, _i64, _vp],
    "bya_p2p_alloc": [_i64, _i32, _c.POINTER(_vp)],
    "bya_p2p_free": [_vp],
    "bya_p2p_ipc_export": [_vp, _vp],
    "bya_p2p_ipc_import": [_vp, _c.POINTER(_vp)],
    "bya_p2p_ipc_release": [_vp],
    "bya_cfg_scheduler_step": [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _c.POINTER(SchedCoef), _vp],
}

# bya_option keys / BYA_REF_* bits of include/bya.h
OPTIONS = {"gemm_splitk": 0, "gemm_splitk_min": 1, "gemm_tile": 2, "gemm_variant": 3, "attn_streamk": 4, "fp8_kernel": 5,
           "p2p_groups": 6, "reference_forms": 7}
REFERENCE_FORMS = {"rowgemm_chunked": 1, "kv_mix_generic": 2, "ln_generic": 4, "router_scores_wave": 8, "attn_narrow_store": 16}
# environment variable -> (option, parser): read ONCE, when the library is loaded (the C entry points never call getenv)
ENV_OPTIONS = {
    "BYA_GEMM_SPLITK": ("gemm_splitk", int),
    "BYA_GEMM_SPLITK_MIN": ("gemm_splitk_min", int),
    "BYA_GEMM_TILE": ("gemm_tile", int),
    "BYA_GEMM_VARIANT": ("gemm_variant", lambda v: {"w8": 1, "no128": 2}.get(v, 0)),
    "BYA_ATTN_STREAMK": ("attn_streamk", int),
    "BYA_FP8_KERNEL": ("fp8_kernel", lambda v: 1 if v.startswith("1") else 0),
    "BYA_P2P_GROUPS": ("p2p_groups", int),
}

ERRORS = {-1: "BYA_ERR_SHAPE", -2: "BYA_ERR_ALIGN", -3: "BYA_ERR_LAUNCH", -4: "BYA_ERR_UNSUPPORTED"}

_lib = None


def load():
    """Load the HIP library; raises if it is not built (run ``python -m ...build`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    # ONE HIP runtime per process: torch must map its (bundled) libamdhip64 first so that this library binds to
    # the same runtime that owns torch's device pointers and streams.  Loading in the other order gives two
    # runtimes and hipErrorNoDevice at the first launch.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the Bind-Your-Avatar MI355X engine has no fallback path. "
            "Build it with `python -m bind_your_avatar_implementation_amd.build`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export what bya.h declares
        fn.argtypes = argtypes
        fn.restype = _i32
    _lib = lib
    apply_env_options()
    return lib


def set_option(name, value):
    """bya_set_option by name (``OPTIONS``); raises on an unknown name or a value outside the option's range."""
    check(load().bya_set_option(OPTIONS[name], int(value)), f"bya_set_option({name}={value})")


def get_option(name):
    v = _i32(0)
    check(load().bya_get_option(OPTIONS[name], _c.byref(v)), f"bya_get_option({name})")
    return v.value


OPTION_DEFAULTS = {"gemm_splitk": 0, "gemm_splitk_min": 0, "gemm_tile": -1, "gemm_variant": 0, "attn_streamk": 1, "fp8_kernel": 0,
                   "p2p_groups": 0, "reference_forms": 0}


def apply_env_options():
    """Hand the BYA_* tuning variables of the environment to the library's option table; an option whose variable is unset
    goes back to its default.  Runs when the library is loaded; call it again after changing one of the variables in a live
    process (tools/ do)."""
    for var, (name, parse) in ENV_OPTIONS.items():
        v = os.environ.get(var)
        set_option(name, parse(v) if v not in (None, "") else OPTION_DEFAULTS[name])


class ByaError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise ByaError(f"{what} failed: {ERRORS.get(rc, rc)}")
