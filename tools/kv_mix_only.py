"""bya_attn_kv_mix at the step's audio shape, a few launches (for rocprofv3 --pmc passes: tools/r4_new_kernels_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
T, pf, D, H, nid = 13, 1350, 3072, 48, 2
q, k, v = rnd(T, pf, D), rnd(nid, T, 32, D), rnd(nid, T, 32, D)
r = torch.sigmoid(torch.randn(T * pf, nid, device=dev)).to(torch.bfloat16)
af = torch.eye(nid, device=dev, dtype=torch.bfloat16)
z, ws = torch.empty(T, pf, D, dtype=torch.bfloat16, device=dev), torch.empty(T * pf, dtype=torch.float32, device=dev)
for _ in range(4):
    ops.attn_kv_mix(q, k, v, r, af, z, ws, head_dim=64, heads=H, n_id=nid, n_grp=T, Sq=pf, Skv=32, q_strides=(pf * D, D),
                    k_strides=(T * 32 * D, 32 * D, D), v_strides=(T * 32 * D, 32 * D, D), z_strides=(pf * D, D), scale=0.125)
torch.cuda.synchronize()
