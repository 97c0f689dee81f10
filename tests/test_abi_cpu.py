"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/bya.h declares
(no compute is launched: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from bind_your_avatar_implementation_amd.build import build_hip_library
    return build_hip_library()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "bya.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(bya_\w+)\s*\(", src)))


def test_header_declares_the_documented_entry_points():
    syms = declared_symbols()
    for must in ["bya_gemm_bf16", "bya_gemm_skinny_bf16", "bya_attn_fwd", "bya_layernorm", "bya_qknorm_rope", "bya_masked_combine",
                 "bya_router_scores", "bya_router_head", "bya_forcing_max_over_frames", "bya_patchify",
                 "bya_unpatchify", "bya_linear_small_m", "bya_timestep_features", "bya_attn_tiny", "bya_act_add",
                 "bya_abi_version"]:
        assert must in syms


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/bya.h but not exported"
    assert lib.bya_abi_version() >= 1


def test_python_binding_table_matches_header(lib_path):
    from bind_your_avatar_implementation_amd import _hip
    assert sorted(_hip.SIGNATURES) == declared_symbols()
    lib = _hip.load()
    # argument validation happens before any launch: NULL pointers / bad shapes are rejected on a CPU-only box too
    d = _hip.GemmDesc()
    assert lib.bya_gemm_bf16(None, None, None, None, None, None, None, ctypes.byref(d), None) == -1
    a = _hip.AttnDesc()
    assert lib.bya_attn_fwd(None, None, None, None, ctypes.byref(a), None) == -1
    # the softmax-variant query mirrors bya_attn_fwd's kernel choice (host-side, launches nothing)
    a.head_dim, a.scores_prescaled, a.score_bound = 64, 1, 11.8
    assert lib.bya_attn_variant(ctypes.byref(a)) == 4          # static bound on the one-wave-per-SIMD kernel
    a.score_bound = 60.0                                        # P = exp2(s) without an offset: usable up to 90
    assert lib.bya_attn_variant(ctypes.byref(a)) == 4
    a.score_bound = 100.0
    assert lib.bya_attn_variant(ctypes.byref(a)) == 1
    dummy = (ctypes.c_float * 4)()
    a.bound_dev, a.fallback_flags = ctypes.addressof(dummy), ctypes.addressof(dummy)      # data-dependent bound: static kernel +
    assert lib.bya_attn_variant(ctypes.byref(a)) == 5                                      # per-head running-maximum fallback
    a.bound_dev, a.fallback_flags = None, None
    a.scores_prescaled = 0
    assert lib.bya_attn_variant(ctypes.byref(a)) == 0
    a.head_dim = 128
    assert lib.bya_attn_variant(ctypes.byref(a)) == 3
    # the RCCL entry points validate their arguments before touching a communicator
    assert lib.bya_allgather_kv(None, None, None, None, 1, 1, None, None) == -1
    cnt = (ctypes.c_int64 * 2)(1, 1)
    assert lib.bya_alltoall_router(None, None, cnt, cnt, 2, None, None) == -1


def test_struct_layout_matches_header():
    """ctypes mirrors of bya_gemm_desc / bya_attn_desc: field order and sizes as in the header."""
    from bind_your_avatar_implementation_amd import _hip
    src = open(os.path.join(ROOT, "include", "bya.h")).read()
    for cname, cls in (("bya_gemm_desc", _hip.GemmDesc), ("bya_attn_desc", _hip.AttnDesc)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *more = decl.split(",")
            typ, name = re.match(r"(.*?)(\w+)$", first.strip(), flags=re.S).groups()
            fields += [(name, typ.strip())] + [(n.strip(), typ.strip()) for n in more]
        assert [f[0] for f in fields] == [f[0] for f in cls._fields_], cname
        size = {"int32_t": 4, "int64_t": 8, "float": 4, "const float*": 8, "int32_t*": 8}
        for (n, typ), (_, ct) in zip(fields, cls._fields_):
            assert ctypes.sizeof(ct) == size[typ], (cname, n)


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: a CPU-resident model must refuse to run instead of computing with torch."""
    import torch
    from bind_your_avatar_implementation_amd import BindyouravatarTransformer3DModel
    with torch.device("meta"):
        m = BindyouravatarTransformer3DModel(num_layers=1, in_channels=48, use_rotary_positional_embeddings=True,
                                             use_learned_positional_embeddings=True, is_train_audio=True)
    from bind_your_avatar_implementation_amd.engine import DenoiseEngine
    with pytest.raises(RuntimeError, match="GPU"):
        DenoiseEngine(m)
    src = open(os.path.join(ROOT, "bind_your_avatar_implementation_amd", "engine.py")).read() + \
        open(os.path.join(ROOT, "bind_your_avatar_implementation_amd", "ops.py")).read() + \
        open(os.path.join(ROOT, "bind_your_avatar_implementation_amd", "transformer.py")).read()
    assert "oracle" not in src.replace("oracle's", ""), "the product must never import the oracle"
    assert "F.scaled_dot_product_attention" not in src and "torch.nn.functional" not in src


def test_generated_gemm_schedules_match_their_tables():
    """The hand-placed instruction streams of gemm_v4.hip, gemm_fp8_v4.hip and attn_w4.hip are emitted by
    tools/gen_*_schedule.py from placement tables; the committed sources must be exactly what the tables generate."""
    import subprocess
    import sys
    for gen in ("gen_gemm_v4_schedule.py", "gen_gemm_v5_schedule.py", "gen_gemm_v6_schedule.py", "gen_gemm_fp8_schedule.py", "gen_attn_w4_schedule.py"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen), "--check"], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr


def test_valu_only_kernels_hold_no_packed_fp32(lib_path):
    """packed-fp32 VALU arithmetic (v_pk_mul / fma / add / mov) in the VALU-only kernels returned
    wrong lanes whenever another process kept MFMA workgroups resident; build.py compiles those translation units
    without the SLP vectoriser.  Guard it on the built device code (and check the disassembly is really read: the
    router's tiny-attention kernels hold MFMAs, the MFMA GEMM holds both)."""
    import shutil
    import subprocess
    import tempfile
    from bind_your_avatar_implementation_amd import build
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm llvm binutils not found")
    packed = re.compile(r"\bv_pk_(mul|fma|add|mov)_(f32|b32)\b")

    def device_asm(src, tmp):
        obj = os.path.join(build.PKG_DIR, "build", src.replace(".hip", ".o"))
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([tools[0], f"--dump-section=.hip_fatbin={fat}", obj], check=True, stdin=subprocess.DEVNULL)
        subprocess.run([tools[1], "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
                        f"--output={co}", "--unbundle"], check=True, stdin=subprocess.DEVNULL)
        return subprocess.run([tools[2], "-d", co], check=True, capture_output=True, text=True,
                              stdin=subprocess.DEVNULL).stdout

    tmp = tempfile.mkdtemp()
    try:
        for src in sorted(build.NO_SLP_SOURCES):
            asm = device_asm(src, tmp)
            assert "s_endpgm" in asm, f"{src}: no device code disassembled"
            hits = packed.findall(asm)
            assert not hits, f"{src}: {len(hits)} packed-fp32 instructions in a VALU-only translation unit"
        assert "v_mfma" in device_asm("router.hip", tmp)
        # The MFMA translation units DO hold packed-fp32 instructions (epilogues, softmax): if the multi-process finding of
        # DESIGN.md section 5 is what it looks like, they are exposed in the same setting (several processes on one GPU).
        # That finding is a workaround, not a closed root cause; the exposure is recorded here so that it is a number and
        # not a guess (profiles/history/r3_packed_fp32_counts.json, rewritten with BYA_RECORD_PACKED_COUNTS=1), and must not grow
        # unnoticed.
        import json
        rec_path = os.path.join(ROOT, "profiles", "history", "r3_packed_fp32_counts.json")
        counts = {src: len(packed.findall(device_asm(src, tmp))) for src in ("gemm.hip", "gemm_v4.hip", "gemm_fp8_v4.hip", "attn.hip", "rowgemm.hip")}
        print("packed-fp32 instructions in the MFMA translation units:", counts)
        if os.environ.get("BYA_RECORD_PACKED_COUNTS") == "1":
            with open(rec_path, "w") as f:
                json.dump(counts, f, indent=1, sort_keys=True)
        assert counts["gemm.hip"] > 0                               # the pattern does match where packed ops exist
        recorded = json.load(open(rec_path))
        for src, n in counts.items():
            assert src in recorded and n <= 1.25 * recorded[src] + 16, (src, n, recorded.get(src))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_library_path_override_fails_loudly_when_the_file_is_missing(tmp_path):
    """BYA_HIP_LIB points the loader at another build of the library (A/B runs of two builds in one call); a path that does
    not exist must fail like a missing in-tree build does -- there is no fallback to look for."""
    import subprocess
    import sys
    code = ("import torch\n"
            "from bind_your_avatar_implementation_amd import _hip\n"
            "try:\n    _hip.load()\nexcept Exception as e:\n    print(type(e).__name__, str(e)[:200]); raise SystemExit(3)\n"
            "print('loaded')\n")
    env = dict(os.environ, BYA_HIP_LIB=str(tmp_path / "nope.so"), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=300)
    assert r.returncode == 3 and "nope.so" in r.stdout and "no fallback" in r.stdout, (r.stdout, r.stderr[-500:])


def test_options_are_set_through_the_abi_and_never_through_the_environment(lib_path):
    """(r6) The C entry points read no environment: `grep getenv csrc/` finds nothing.  Options go through bya_set_option /
    bya_get_option: defaults, ranges, unknown keys; the Python host maps its BYA_* variables onto them when the library is
    loaded (``_hip.apply_env_options``) and ``ops.options`` scopes a change."""
    import subprocess
    import sys
    from bind_your_avatar_implementation_amd import _hip
    csrc = os.path.join(ROOT, "bind_your_avatar_implementation_amd", "csrc")
    for f in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, f)).read(), f"{f}: the C ABI must not read the environment"
    src = open(os.path.join(ROOT, "include", "bya.h")).read()
    keys = dict((m.group(1).lower(), int(m.group(2))) for m in re.finditer(r"BYA_OPT_(\w+) = (\d+)", src))
    count = keys.pop("count")
    assert keys == _hip.OPTIONS and count == len(keys)                   # header enum = the Python table
    refs = dict((m.group(1).lower(), int(m.group(2))) for m in re.finditer(r"BYA_REF_(\w+) = (\d+)", src))
    assert refs == _hip.REFERENCE_FORMS
    lib = _hip.load()
    for name, default in _hip.OPTION_DEFAULTS.items():
        _hip.set_option(name, default)
        assert _hip.get_option(name) == default
    assert lib.bya_set_option(99, 0) == -1 and lib.bya_set_option(-1, 0) == -1            # unknown key
    assert lib.bya_set_option(_hip.OPTIONS["gemm_splitk"], 3) == -1                       # out of range: nothing changes
    assert lib.bya_set_option(_hip.OPTIONS["p2p_groups"], 8) == -1 and lib.bya_set_option(_hip.OPTIONS["p2p_groups"], 64) == 0
    assert _hip.get_option("gemm_splitk") == 0 and _hip.get_option("p2p_groups") == 64
    _hip.set_option("p2p_groups", 0)
    assert lib.bya_get_option(_hip.OPTIONS["gemm_tile"], None) == -1
    # the environment reaches the table once, at load time, in a fresh process
    code = ("from bind_your_avatar_implementation_amd import _hip; _hip.load(); "
            "print(_hip.get_option('gemm_splitk'), _hip.get_option('gemm_variant'), _hip.get_option('gemm_tile'), _hip.get_option('attn_streamk'))")
    env = dict(os.environ, BYA_GEMM_SPLITK="0", BYA_GEMM_VARIANT="w8", BYA_GEMM_TILE="4", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.split() == ["0", "1", "4", "1"], out.stdout + out.stderr[-500:]
