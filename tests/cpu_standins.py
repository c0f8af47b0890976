"""TEST INFRASTRUCTURE ONLY: torch-CPU stand-ins for the HIP ops, used by the gloo rehearsal of the multi-GPU step
(tests/test_distributed_gloo.py).  There is no GPU in the CPU test tier, and the product has no CPU path; these let
the DISTRIBUTED logic of literalkg_amd/distributed.py (partitioning, exchanges, gradient routing, sharded optimizer
state) run end to end on plain autograd.  They restate the ops through the oracle's formulas (oracle/ is the checker
and may be used from tests/); nothing under literalkg_amd/ imports this file."""
import torch
import torch.nn.functional as F

from oracle import literalkg_oracle as O


def spmm(rowptr, col, val, x, n_rows, out=None, x_row_offset=0, long_rows=None, add_self=None):
    rp = rowptr.long()
    lo, hi = int(rp[0]), int(rp[-1])
    rows = torch.repeat_interleave(torch.arange(n_rows), rp[1:] - rp[:-1])
    res = torch.zeros((n_rows, x.shape[1]), dtype=x.dtype)
    res.index_add_(0, rows, val[lo:hi, None] * x[col[lo:hi].long() - x_row_offset])
    if add_self is not None:
        res = res + add_self
    if out is not None:
        out.copy_(res)
        return out
    return res


class CpuKernels:
    """The interface of literalkg_amd.distributed.HipKernels on torch CPU ops."""
    spmm = staticmethod(spmm)

    @staticmethod
    def permute(val, perm):
        return val[perm.long()]

    @staticmethod
    def add(a, b):
        return a + b

    @staticmethod
    def edge_softmax(g, ent, rel, row_lo=0, row_hi=None):
        """attention values in g's entry order, from the oracle's refresh on the raw triples of g"""
        heads = g.entry_rows()
        eptr = g.eptr.long() if g.eptr is not None else torch.arange(g.nnz + 1)
        per_entry = eptr[1:] - eptr[:-1]
        h = torch.repeat_interleave(heads, per_entry)
        t = torch.repeat_interleave(g.col.long(), per_entry)
        a = O.attention_refresh(g.n, ent, rel, h, t, g.rel.long()).coalesce()
        assert torch.equal(a.indices(), g.coo_indices())
        return a.values().contiguous()

    @staticmethod
    def gather_rows_range(block, ids, lo, hi):
        mine = (ids >= lo) & (ids < hi)
        out = torch.zeros((ids.numel(), block.shape[1]), dtype=block.dtype)
        out[mine] = block.detach()[ids[mine] - lo]
        return out

    @staticmethod
    def scatter_add_rows_range(rows, ids, lo, hi, like):
        mine = (ids >= lo) & (ids < hi)
        out = torch.zeros(like.shape, dtype=like.dtype)
        out.index_add_(0, ids[mine] - lo, rows[mine])
        return out


def patch_ops():
    """Replace the HIP-backed functions of literalkg_amd.ops by plain autograd equivalents (same signatures)."""
    from literalkg_amd import ops

    def act_layernorm(z, gamma, beta, want_norm=True, slope=ops.LEAKY_SLOPE, eps=ops.LN_EPS, norm_eps=ops.NORMALIZE_EPS,
                      drop_p=0.0, seed=None, yn_out=None, want_y=True):
        assert drop_p == 0.0, "the rehearsal runs without dropout"
        a = z if slope == 1.0 else F.leaky_relu(z, slope)
        y = F.layer_norm(a, (z.shape[1],), gamma, beta, eps)
        return y, (F.normalize(y, p=2.0, dim=1, eps=norm_eps) if want_norm else None)

    def assemble_cat(holder, parts):
        return torch.cat(list(parts), dim=1)

    def transr_loss(emb, relemb, trans_m, h, r, pos_t, neg_t, lam, keep=None, group=1):
        cfg = O.default_cfg(kg_l2loss_lambda=lam)
        return O.triple_loss_transr({"relation_embed.weight": relemb, "gat_trans_M": trans_m}, cfg, emb, h, r, pos_t, neg_t)

    def transe_loss(emb, relemb, h, r, pos_t, neg_t, lam, keep=None):
        cfg = O.default_cfg(kg_l2loss_lambda=lam)
        return O.triple_loss_transe({"relation_embed.weight": relemb}, cfg, emb, h, r, pos_t, neg_t)

    def is_grouped_batch(h, r, pos_t, k):
        b = h.numel()
        if k <= 1 or b == 0 or b % k:
            return False
        return all(bool((x.view(-1, k) == x.view(-1, k)[:, :1]).all()) for x in (h, r, pos_t))

    def gate_blend(x, gpre, zpre, out=None):
        z = torch.sigmoid(zpre)
        return (1 - z) * x + z * torch.tanh(gpre)

    repl = dict(
        linear=lambda x, w, b=None: F.linear(x, w, b),
        multi_linear=lambda xs, ws, b=None: sum(F.linear(x, w) for x, w in zip(xs, ws)) + (b if b is not None else 0),
        matmul=lambda a, b: a @ b,
        axpby=lambda a, b, alpha=1.0, beta=1.0: alpha * a + (beta * b if b is not None else beta),
        mul=lambda a, b: a * b,
        leaky_relu=lambda a, slope=ops.LEAKY_SLOPE: F.leaky_relu(a, slope),
        leaky_relu_sum=lambda a, b, slope=ops.LEAKY_SLOPE: F.leaky_relu(a, slope) + F.leaky_relu(b, slope),
        act_layernorm=act_layernorm, assemble_cat=assemble_cat, transr_loss=transr_loss, transe_loss=transe_loss,
        is_grouped_batch=is_grouped_batch, gate_blend=gate_blend,
    )
    for k, v in repl.items():
        setattr(ops, k, v)
