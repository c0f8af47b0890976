import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from literalkg_amd import ops
dev = torch.device("cuda:0"); n, d = 1_000_000, 256
x = torch.randn(n, d, device=dev); w = torch.randn(d, d, device=dev) * 0.06; out = torch.empty(n, d, device=dev)
rm = ops.row_absmax(x)
def tm(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
t = tm(lambda: ops.gemm_tall((x,), ((w,),), True, None, out=out, rowmax=rm))
want = x[::4001].double() @ w.double().t()
err = float((out[::4001].double() - want).abs().max() / want.abs().max())
print("RESIDENT", os.environ.get("LKG_TALL_RESIDENT"), f"{t:.3f} ms  {2*n*d*d/t/1e9:.0f} TF  rel err {err:.2e}")
