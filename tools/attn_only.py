import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
S, H, D = 17776, 48, 64
q, k, v = ((torch.randn(1, S, H * D, device=dev)).to(torch.bfloat16) for _ in range(3))
out = torch.empty_like(q)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ops.self_attention(q, k, v, out, heads=H)
torch.cuda.synchronize()
