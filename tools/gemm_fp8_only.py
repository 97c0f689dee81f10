import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bind_your_avatar_implementation_amd import ops
dev = torch.device("cuda:0")
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (17776, 9216, 3072)))
a8, sa = ops.quantize_rows_fp8(torch.randn(M, K, device=dev).to(torch.bfloat16))
w8, sw = ops.quantize_rows_fp8((torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16))
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 3):
    ops.gemm_fp8(a8, sa, w8, sw, out)
torch.cuda.synchronize()
