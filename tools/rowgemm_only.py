import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
M = 35100
N, ln, res, ns = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
x = torch.randn(M, 512, device=dev).to(torch.bfloat16)
w = (torch.randn(N, 512, device=dev) * 512 ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device=dev).to(torch.bfloat16)
gam, bet = torch.randn(512, device=dev).to(torch.bfloat16), torch.randn(512, device=dev).to(torch.bfloat16)
pack = ops.pack_rowgemm512(w, b, gam if ln else None, bet if ln else None)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(5):
    ops.rowgemm512(x, pack, out, res=out if res else None, nsplit=ns)
torch.cuda.synchronize()
