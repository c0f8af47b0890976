"""CPU, world_size 2, gloo: the row-range sharding + gradient all-reduce choreography of
literalkg_amd/sharding.py.  The product default backend is the HIP SpMM (GPU only); here the test
injects a torch-CPU SpMM with the same call signature so the N>1 logic is rehearsed without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def cpu_spmm(rowptr, col, val, x, n_rows, out=None, x_row_offset=0, long_rows=None, add_self=None, add2=None):
    rp = rowptr.long()
    lo, hi = int(rp[0]), int(rp[-1])
    rows = torch.repeat_interleave(torch.arange(n_rows), rp[1:] - rp[:-1])
    res = torch.zeros((n_rows, x.shape[1]), dtype=x.dtype)
    res.index_add_(0, rows, val[lo:hi, None] * x[col[lo:hi].long() - x_row_offset])
    if add_self is not None:
        res = res + add_self
    if add2 is not None:
        res = res + add2
    if out is not None:
        out.copy_(res)
        return out
    return res


def cpu_permute(val, perm):
    return val[perm.long()]


def _worker(rank, world, port, n, d, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from literalkg_amd import KGStructure
        from literalkg_amd.sharding import ShardedAggregation
        rng = np.random.default_rng(11)                       # same graph on every rank
        e = 6000
        h = (n * rng.random(e) ** 2.0).astype(np.int64)
        t = rng.integers(0, n, e)
        r = rng.integers(0, 3, e)
        full = KGStructure.from_triples(n, h, t, r)
        val_full = torch.from_numpy(rng.random(full.nnz).astype(np.float32))
        cuts = full.row_cuts(world)
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        keep = (h >= lo) & (h < hi)
        mine = KGStructure.from_triples(n, h[keep], t[keep], r[keep])
        rp = full.host("rowptr")
        val = val_full[rp[lo]:rp[hi]]                         # the shard's entries are a contiguous slice
        assert mine.nnz == val.numel()
        shard = ShardedAggregation(mine, val, lo, hi, spmm=cpu_spmm, permute=cpu_permute, n_chunks=3)
        x = torch.from_numpy(np.random.default_rng(5).standard_normal((n, d)).astype(np.float32))
        gfull = torch.from_numpy(np.random.default_rng(6).standard_normal((n, d)).astype(np.float32))
        side = shard.forward(x)
        grad = shard.backward(gfull[lo:hi].contiguous())
        a = torch.sparse_coo_tensor(full.coo_indices(), val_full, (n, n)).coalesce()
        want_side = torch.matmul(a, x)[lo:hi]
        want_grad = torch.matmul(a.t(), gfull)
        ok = torch.allclose(side, want_side, rtol=1e-5, atol=1e-5) and \
            torch.allclose(grad, want_grad, rtol=1e-5, atol=1e-4)
        q.put((rank, bool(ok), lo, hi, float((grad - want_grad).abs().max())))
    finally:
        dist.destroy_process_group()


def _feature_worker(rank, world, port, n, d, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from literalkg_amd import KGStructure
        from literalkg_amd.sharding import FeatureShardedAggregation, shard_bounds
        rng = np.random.default_rng(21)
        e = 5000
        h = (n * rng.random(e) ** 2.0).astype(np.int64)
        t = rng.integers(0, n, e)
        g = KGStructure.from_triples(n, h, t, rng.integers(0, 3, e))
        val = torch.from_numpy(rng.random(g.nnz).astype(np.float32))
        cuts = shard_bounds(g, world)
        fs = FeatureShardedAggregation(g, val, rank, world, d, cuts, spmm=cpu_spmm, permute=cpu_permute)
        x = torch.from_numpy(np.random.default_rng(5).standard_normal((n, d)).astype(np.float32))
        gside = torch.from_numpy(np.random.default_rng(6).standard_normal((n, d)).astype(np.float32))
        a = torch.sparse_coo_tensor(g.coo_indices(), val, (n, n)).coalesce()
        lo, hi = cuts[rank], cuts[rank + 1]
        cols = slice(rank * fs.dg, (rank + 1) * fs.dg)
        side_slab = fs.forward(fs.column_slab(x))                      # all rows, my columns
        ok = torch.allclose(side_slab, torch.matmul(a, x)[:, cols], rtol=1e-5, atol=1e-5)
        block = fs.to_row_block(side_slab)                             # my rows, all columns, as G panels
        want_rows = torch.matmul(a, x)[lo:hi]
        got_rows = torch.cat([block[i] for i in range(world)], dim=1)
        ok &= torch.allclose(got_rows, want_rows, rtol=1e-5, atol=1e-5)
        side2, block2 = fs.forward_to_row_block(fs.column_slab(x))      # SpMM pipelined with point-to-point sends
        ok &= torch.equal(side2, side_slab) and torch.equal(block2, block)
        # backward: my rows of grad_side arrive as panels, go back to a column slab, then A^T
        gblock = torch.stack([gside[lo:hi, i * fs.dg:(i + 1) * fs.dg] for i in range(world)]).contiguous()
        gslab = fs.to_column_slab(gblock)
        ok &= torch.equal(gslab, gside[:, cols])
        gego = fs.backward(gslab)
        ok &= torch.allclose(gego, torch.matmul(a.t(), gside)[:, cols], rtol=1e-5, atol=1e-4)
        for pieces in (None, 2, 3):                                    # exchange in column pieces, SpMM per piece
            gego2 = fs.backward_from_row_block(gblock, pieces=pieces)
            ok &= torch.allclose(gego2, gego, rtol=1e-6, atol=1e-6)
        # exchange in row sub-ranges of every block (every batch uses every link), the transpose SpMM part by part behind
        # them (the N = 8 form: D/G too narrow to cut by columns) -- the parts partition the CSC's entries: part 0 gathers from
        # this rank's own rows, part q + 1 from sub-range q of the others'
        for nb in (None, 1, 2, 3, 5):
            before = fs.bytes_sent
            gego3 = fs.backward_in_head_parts(gblock, n_batches=nb)
            ok &= torch.allclose(gego3, gego, rtol=1e-5, atol=1e-5)
            ok &= fs.bytes_sent - before == 4 * fs.dg * (hi - lo) * (world - 1)       # my rows, once to every peer
            n_chunks, parts, vals = fs.head_parts(nb)
            ok &= n_chunks == (3 if nb is None else nb) and len(parts) == n_chunks + 1
            ok &= sum(p.nnz for p in parts) == g.nnz
            heads_of_part0 = parts[0].col.long()
            ok &= bool(((heads_of_part0 >= lo) & (heads_of_part0 < hi)).all())          # part 0: my own block only
        # both exchanges folded into the pass (the integrated step's aggregation): my rows in as panels, my rows out
        for transposed, src in ((False, x), (True, gside)):
            blk_in = torch.stack([src[lo:hi, i * fs.dg:(i + 1) * fs.dg] for i in range(world)]).contiguous()
            for plus_self in (False, True):
                for nb in (None, 1, 2):
                    _, blk_out = fs.exchange_aggregate(transposed, block_in=blk_in, plus_self=plus_self, n_batches=nb, pieces=3)
                    want_full = torch.matmul(a.t() if transposed else a, src) + (src if plus_self else 0)
                    got = torch.cat([blk_out[i] for i in range(world)], dim=1)
                    ok &= torch.allclose(got, want_full[lo:hi], rtol=1e-5, atol=1e-4)
        val2 = val * 0.5                                               # an attention refresh reaches the parts' values
        fs.set_values(val2)
        ok &= torch.allclose(fs.backward_in_head_parts(gblock), 0.5 * gego, rtol=1e-5, atol=1e-5)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_feature_sharded_forward_exchange_backward(world):
    import __graft_entry__ as ge
    ge.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_feature_worker, args=(r, world, port, 403, 24, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)], res


@pytest.mark.timeout(120)
def test_two_rank_sharded_forward_backward():
    import __graft_entry__ as ge
    ge.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, 500, 12, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] for r in res), res
    assert res[0][2] == 0 and res[0][3] == res[1][2] and res[1][3] == 500     # contiguous row ranges
