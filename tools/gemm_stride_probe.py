"""Does the row stride of A matter (HBM channel hot-spotting of column-slice reads)?  GPU box only."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from literalkg_amd import ops
dev = torch.device("cuda:0"); n, d = 1_000_000, 256
w = torch.randn(d, d, device=dev) * 0.06; out = torch.empty(n, d, device=dev)
def tm(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for ld in (256, 260, 272, 288, 320, 384):
    x = torch.randn(n, ld, device=dev)[:, :d]
    rm = ops.row_absmax(x)
    ops._ENGINE = "f16x2-all"
    t1 = tm(lambda: ops.gemm_tall((x,), ((w,),), True, None, out=out, rowmax=rm))
    ops._ENGINE = "bf16x3"
    t2 = tm(lambda: ops.gemm(x, w, trans_b=True, out=out))
    o2 = torch.empty(n, ld, device=dev)[:, :d]
    t3 = tm(lambda: ops.gemm(x, w, trans_b=True, out=o2))
    print(f"lda {ld}: tall f16x2 {t1:.3f} ms | bf16x3 {t2:.3f} ms | bf16x3 with ldc {ld} too {t3:.3f} ms")
