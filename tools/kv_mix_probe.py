"""Times bya_attn_kv_mix at the step's two shapes (audio: 13 frames x 1350 rows x 48 heads x d64; face: 17550 rows x 16 heads
x d128): the <= 32-key persistent form (default) against the one-tile-per-workgroup kernel (BYA_KV_MIX32=0), and checks that
the two agree bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
T, pf, D, H, nid = 13, 1350, 3072, 48, 2
q, k, v = rnd(T, pf, D), rnd(nid, T, 32, D), rnd(nid, T, 32, D)
r = torch.sigmoid(torch.randn(T * pf, nid, device=dev)).to(torch.bfloat16)
af = torch.eye(nid, device=dev, dtype=torch.bfloat16)
z, ws = torch.empty(T, pf, D, dtype=torch.bfloat16, device=dev), torch.empty(T * pf, dtype=torch.float32, device=dev)
audio = lambda: ops.attn_kv_mix(q, k, v, r, af, z, ws, head_dim=64, heads=H, n_id=nid, n_grp=T, Sq=pf, Skv=32,
      q_strides=(pf * D, D), k_strides=(T * 32 * D, 32 * D, D), v_strides=(T * 32 * D, 32 * D, D), z_strides=(pf * D, D), scale=0.125)
def ab(name, fn, out):
    res = {}
    for flag in ("1", "0"):
        os.environ["BYA_KV_MIX32"] = flag
        out.fill_(7)
        fn(); torch.cuda.synchronize()
        res[flag] = (out.clone(), timed(fn))
    os.environ.pop("BYA_KV_MIX32")
    print(f"{name}: 32-key form {res['1'][1]:.1f} us, one tile per workgroup {res['0'][1]:.1f} us, bit-identical {torch.equal(res['1'][0], res['0'][0])}")
ab("audio d64", audio, z)
N = T * pf
qp, kv = rnd(N, 2048), rnd(nid, 32, 4096)
zp = torch.empty(N, 2048, dtype=torch.bfloat16, device=dev)
ab("face d128", lambda: ops.attn_kv_mix(qp, kv, kv[..., 2048:], r, None, zp, None, head_dim=128, heads=16, n_id=nid, n_grp=1, Sq=N, Skv=32,
      q_strides=(0, 2048), k_strides=(32 * 4096, 0, 4096), v_strides=(32 * 4096, 0, 4096), z_strides=(0, 2048), scale=128 ** -0.5), zp)
