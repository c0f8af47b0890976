import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/literalkg_amd") else os.getcwd())
from literalkg_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for n_rel, c, dout, n_g, k, scale, common in ((1, 64, 8, 683, 3, 1.0, 0.0), (1, 64, 8, 683, 3, 0.05, 0.0), (1, 64, 8, 683, 3, 0.01, 1.0), (4, 64, 8, 683, 3, 0.01, 1.0),
                                              (1, 64, 8, 683, 1, 0.01, 1.0), (3, 300, 300, 683, 3, 0.01, 1.0), (16, 300, 300, 683, 3, 0.01, 1.0), (1, 64, 8, 64, 1, 0.05, 0.0)):
    n = 9000
    emb = (torch.randn(n, c) * scale + common * torch.randn(1, c)).to(dev).requires_grad_(True)       # (common: rows that share a large part)
    rel = (torch.randn(n_rel, dout) * 0.1).to(dev).requires_grad_(True)
    M = (torch.randn(n_rel, c, dout) * 0.1).to(dev).requires_grad_(True)
    hg = torch.randint(0, n, (n_g,)); rg = torch.randint(0, n_rel, (n_g,)); pg = torch.randint(0, n, (n_g,))
    h, r, pt = hg.repeat_interleave(k).to(dev), rg.repeat_interleave(k).to(dev), pg.repeat_interleave(k).to(dev)
    nt = torch.randint(0, n, (n_g * k,)).to(dev)
    loss = ops.transr_loss(emb, rel, M, h, r, pt, nt, 1e-5, None, k, False)
    loss.backward()
    # float64 reference (model.py:364-428)
    e64, r64, M64 = (t.detach().double().cpu().requires_grad_(True) for t in (emb, rel, M))
    hh, rr, pp, nn_ = h.cpu(), r.cpu(), pt.cpu(), nt.cpu()
    W = M64[rr]
    rh = torch.bmm(e64[hh].unsqueeze(1), W).squeeze(1); rp = torch.bmm(e64[pp].unsqueeze(1), W).squeeze(1); rn = torch.bmm(e64[nn_].unsqueeze(1), W).squeeze(1)
    re = r64[rr]
    pos = ((rh + re - rp) ** 2).sum(1); neg = ((rh + re - rn) ** 2).sum(1)
    l2 = lambda x: (x ** 2).sum(1).mean() / 2
    want = (-torch.nn.functional.logsigmoid(neg - pos)).mean() + 1e-5 * (l2(rh) + l2(re) + l2(rp) + l2(rn))
    want.backward()
    def err(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
    # the same in fp32 on the CPU (the reference's order: per-sample terms added before the sum over the batch)
    e32, r32, M32 = (t.detach().float().cpu().requires_grad_(True) for t in (emb, rel, M))
    W3 = M32[rr]
    a_, b_, c_ = (torch.bmm(e32[i].unsqueeze(1), W3).squeeze(1) for i in (hh, pp, nn_))
    d_ = r32[rr]
    ((-torch.nn.functional.logsigmoid(((a_ + d_ - c_) ** 2).sum(1) - ((a_ + d_ - b_) ** 2).sum(1))).mean() + 1e-5 * (l2(a_) + l2(d_) + l2(b_) + l2(c_))).backward()
    print(f"   fp32 CPU autograd g_M error {err(M32.grad, M64.grad):.2e}")
    print(f"n_rel {n_rel} c {c} dout {dout} groups {n_g} x {k} scale {scale} common {common}: loss {float(loss):.6f} vs {float(want):.6f} | g_M {err(M.grad, M64.grad):.2e} (largest {float(M64.grad.abs().max()):.2e})  g_emb {err(emb.grad, e64.grad):.2e}  g_rel {err(rel.grad, r64.grad):.2e}")
