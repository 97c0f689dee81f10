"""The bench.py contract (one JSON line; metric / config of BASELINE.json; roofline and cpu_baseline objects) checked on the
committed output of the end-of-round run (profiles/r6_final_bench.json = stdout of ``python bench.py --steps 20 --warmup 5``
on an MI355X) and on the script's command line, without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(path):
    lines = [l for l in open(path).read().splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, "bench.py prints exactly ONE JSON line"
    return json.loads(lines[0])


def test_committed_bench_line_meets_the_contract():
    import glob
    d = _line(os.path.join(ROOT, "profiles", "r6_final_bench.json"))                             # the end-of-round line of round 6
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    # BASELINE.json: "denoise-steps/sec + latent frames/sec, 49x480x720 bf16, 1/2/4/8 MI355X"
    assert base["metric"].startswith(d["metric"]) and d["unit"] == "steps/s" and "latent_frames_per_sec" in d
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] >= 5 and d["warmup"] >= 2 and d["higher_is_better"] is True
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["layers"] == 42
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["peak"] == 2500.0
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == d["unit"]
    # the dominant kernel by time is the one reported as `roofline`
    km = d["kernel_ms_per_step"]
    assert max(km, key=km.get) == "bya_gemm_bf16" and "bya_gemm_bf16" in r["kernel"]
    # fixed names since round 5: `frac` prices EVERY Linear of the step (the definition of rounds 1-3), `token_stream_frac` leaves
    # the conditioning's weight-streaming Linears out (round 4's `frac`); executed MFMA work beside the algorithmic 443.9 TFLOP
    assert r["frac"] == r["all_linears_frac"] and r["token_stream_frac"] >= r["frac"]
    assert 350 < d["executed_tflop_per_step"] < 443.9 and d["mfma_executed_frac"] < d["mfma_roofline_frac_whole_step"]
    # the q/k-norm launches are gone (inside the QKV projection's epilogue); the variants beside the headline are there
    assert "bya_qknorm_rope" not in km
    assert d["large_qk_gain_variant"]["attention_variants"] == {"joint:d64_static_bound_w4": 168} and d["large_qk_gain_variant"]["finite"]
    assert d["fp8_weights_variant"]["value"] > d["value"]
    # the SURVEY 8(d) form of the CPU baseline is quoted from its committed run
    assert c["config0_reference"]["file"].startswith("profiles/history/r5_") and c["config0_reference"]["value"] > 0
    # (r6) the line can be read on a pool whose boxes differ by +-3 %: what THIS board sustains on a bare-MFMA loop, and the
    # two big kernels against it beside the spec-peak fractions (which stay the headline); the router's share; the hand-off mode
    cal = d["board_calibration_tflops"]
    assert 1500 < cal < 2500 and abs(r["frac_of_board"] - r["achieved"] / cal) < 1e-9 and r["frac_of_board"] > r["frac"]
    assert abs(d["attn_roofline"]["frac_of_board"] - d["attn_roofline"]["achieved"] / cal) < 1e-9
    assert abs(d["mfma_roofline_frac_of_board_whole_step"] - d["mfma_roofline_frac_whole_step"] * 2500.0 / cal) < 1e-9
    from bench import ROUTER_TIMERS
    assert abs(d["router_ms_per_step"] - sum(km.get(k, 0.0) for k in ROUTER_TIMERS)) < 0.01
    assert "stream-K joint attention" in d["handoff_mode"]


def test_committed_rehearsal_lines_of_the_n_gpu_path_carry_their_own_reference():
    """(r6) A line of `bench.py --gpus N` names the same node's UNSHARDED step and the speed-up over it (a 1-GPU line from another
    box of the pool cannot be divided into it), validates the transport before it times, replays from a hipGraph on the P2P rungs
    and, for N >= 4, measures the CFG pair as well.  Checked on the committed one-GPU rehearsals (all ranks on one GPU: the code
    path, not a measurement)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r6_final_bench_*_ranks_on_one_gpu_42_layers.json")))
    assert files, "no committed rehearsal of the N > 1 bench path"
    for f in files:
        d = _line(f)
        n = d["n_gpus"]
        assert n in (2, 4, 8) and d["config"]["layers"] == 42 and d["config"]["launch"] == "hipGraph replay"
        val = d["config"]["validated_against_unsharded_step"]
        assert val["rungs"][-1]["bit_identical_to_unsharded_step"] and val["default_mode_rel_fro_vs_reference"] <= val["default_mode_bound"]
        assert len(d["unsharded_step_ms_per_rank"]) == n and len(d["board_calibration_tflops_per_rank"]) == n
        assert abs(d["speedup_vs_unsharded_same_node"] - d["unsharded_step_ms_same_node"] / d["ms_per_step"]) < 1e-6
        if n >= 4:
            cp = d["cfg_pair_variant"]
            assert cp["batch"] == 2 and cp["rel_fro_vs_unsharded_batch2_step"] <= 3e-2
            assert abs(cp["speedup_vs_unsharded_batch2_same_node"] - cp["unsharded_batch2_step_ms_same_node"] / cp["ms_per_step"]) < 1e-6


def test_bench_command_line_parses_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's command shape for N > 1) must start
    the two ranks itself -- as a child `python -m torch.distributed.run`, never by re-executing a process that touched the
    GPU -- and relay rank 0's line and the exit code.  --launch-check makes the ranks meet over gloo and skip the GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"launch_check": 2, "rank_sum": 3.0}
    # a launcher whose world disagrees with --gpus is refused before anything else happens
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=dict(env, WORLD_SIZE="4", RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "nproc-per-node" in (bad.stdout + bad.stderr)
