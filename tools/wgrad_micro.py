"""Weight-gradient product (both operands k-major, k = rows): time + accuracy against f64."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from literalkg_amd import ops
dev = torch.device("cuda:0"); n = 1_000_000
def tm(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for m, k in ((256, 256), (512, 256), (512, 300)):
    g = torch.randn(n, m, device=dev); x = torch.randn(n, k, device=dev)
    out = ops.gemm(g, x, trans_a=True)
    t = tm(lambda: ops.gemm(g, x, trans_a=True))
    want = g[:, :8].double().t() @ x.double()
    err = float((out[:8].double() - want).abs().max() / want.abs().max())
    print(f"dW[{m} x {k}] over {n} rows: bf16x3 {t:.3f} ms  {2*n*m*k/t/1e9:.0f} TF f32-eq  rel err {err:.2e}")
    ca, cb = ops.col_absmax(g), ops.col_absmax(x)
    out2 = ops.gemm_wgrad(g, x, ca, cb)
    t2 = tm(lambda: ops.gemm_wgrad(g, x, ca, cb))
    err2 = float((out2[:8].double() - want).abs().max() / want.abs().max())
    print(f"                         f16x2  {t2:.3f} ms  {2*n*m*k/t2/1e9:.0f} TF f32-eq  rel err {err2:.2e}")
