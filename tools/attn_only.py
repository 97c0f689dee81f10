import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
S, H, D = 17776, 48, 64
bounded = len(sys.argv) > 2 and sys.argv[2] == "bounded"
nrm = lambda t: (t / t.view(1, S, H, D).float().norm(dim=-1, keepdim=True).repeat_interleave(D, -1).view(1, S, H * D) * 8).to(torch.bfloat16)
q, k, v = (torch.randn(1, S, H * D, device=dev) for _ in range(3))
q, k, v = nrm(q), (nrm(k).float() * (0.125 * 1.4426950408889634 if bounded else 1.0)).to(torch.bfloat16), v.to(torch.bfloat16)
out = torch.empty_like(q)
kw = dict(prescaled=True, score_bound=11.8) if bounded else {}
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ops.self_attention(q, k, v, out, heads=H, **kw)
torch.cuda.synchronize()
