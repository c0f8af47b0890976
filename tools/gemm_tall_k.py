"""Tall GEMM at several reduction lengths: steady-state k-loop rate vs per-workgroup prologue / epilogue cost."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from literalkg_amd import ops
dev = torch.device("cuda:0"); n, d = 1_000_000, 256
def tm(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
res = {}
for k in (64, 256, 512, 1024):
    x = torch.randn(n, k, device=dev); w = torch.randn(d, k, device=dev) * 0.06; out = torch.empty(n, d, device=dev)
    rm = ops.row_absmax(x)
    t = tm(lambda: ops.gemm_tall((x,), ((w,),), True, None, out=out, rowmax=rm))
    res[k] = t
    print(f"K={k:5d}: {t:.3f} ms  {2*n*d*k/t/1e9:.0f} TF f32-eq   HBM floor {(n*(k+d)*4)/6.3e12*1e3:.3f} ms")
    del x, out
print(f"per 16-k step of all workgroups: {(res[1024]-res[256])/48*1e3:.2f} us ; fixed part at K=256: {res[256]-(res[1024]-res[256])/48*16:.3f} ms")
