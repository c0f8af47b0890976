"""GateMul forward at the C3 shape, a few launches (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import literalkg_amd as L
dev = torch.device("cuda:0"); n, d = 1_000_000, 256
gate = L.GateMul(d, 2, 300).to(dev)
x = torch.randn(n, d, device=dev) * 0.05; num = torch.rand(n, 2, device=dev); txt = torch.randn(n, 300, device=dev)
out = torch.empty(n, d, device=dev)
with torch.no_grad():
    for _ in range(6):
        gate(x, num, txt, out)
torch.cuda.synchronize()
