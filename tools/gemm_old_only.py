import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from literalkg_amd import ops
ops._ENGINE = "bf16x3"
dev = torch.device("cuda:0"); n, d = 1_000_000, 256
x = torch.randn(n, d, device=dev); w = torch.randn(d, d, device=dev) * 0.06; out = torch.empty(n, d, device=dev)
for _ in range(6): ops.gemm(x, w, trans_b=True, out=out)
torch.cuda.synchronize()
