"""CPU, gloo, world_size 2 and 4: the multi-GPU training step of literalkg_amd/distributed.py (row-sharded module,
both aggregation exchange schemes, loss-row gather, weight-gradient all-reduce, sharded optimizer state, update_att
over the shards) replays the REFERENCE's short training runs (tests/golden/trajectory_*.npz: Adam steps on fresh
batches with an update_att in the middle): per-step losses within 1e-4 and the final weights.

The kernels are torch-CPU stand-ins defined in tests/cpu_standins.py (no GPU in this tier, no CPU path in the
product); the same choreography runs on the HIP kernels in tests/test_distributed_gpu.py (2 ranks on one GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_cfg, golden_names, golden_params, load_golden


def replay(g, scheme, device, kernels, make_opt, group_reuse=True):
    """One rank's replay of a trajectory fixture; returns (losses, full state dict)."""
    from literalkg_amd.distributed import ShardedLiteralKG
    cfg = golden_cfg(g)
    cfg.device = device
    n, n_rel = int(g["n"]), int(g["n_rel"])
    state = dict(golden_params(g))
    state["A_in"] = torch.sparse_coo_tensor(torch.from_numpy(g["a_indices"]), torch.from_numpy(g["a_values"]),
                                            (n, n)).coalesce()
    num = torch.from_numpy(g["num"]).to(device) if "num" in g else None
    txt = torch.from_numpy(g["txt"]).to(device) if "txt" in g else None
    m = ShardedLiteralKG.from_full(cfg, n, n_rel, state, num, txt, scoring="transr", scheme=scheme, device=device,
                                   kernels=kernels)
    m.local.group_reuse = group_reuse
    m.train()                                                  # mess_dropout = 0 in the fixtures: deterministic
    opt = make_opt(m.parameters(), float(g["lr"]))
    h, t, r = (torch.from_numpy(g[k]).to(device) for k in "htr")
    losses = []
    for step, b in enumerate(g["batches"]):
        opt.zero_grad()
        loss = m(*[torch.from_numpy(x).to(device) for x in b], device=device, mode="pre_training")
        loss.backward()
        m.sync_gradients()
        opt.step()
        losses.append(float(loss.detach()))
        if step == int(g["refresh_after"]):
            m(h, t, r, list(range(n_rel)), device=device, mode="update_att")
    return losses, m.full_state_dict(), m


def check(g, losses, final, rtol_w=2e-4):
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-4)
    a = final["A_in"].coalesce()
    assert np.array_equal(a.indices().cpu().numpy(), g["final_a_indices"])
    np.testing.assert_allclose(a.values().cpu().numpy(), g["final_a_values"], rtol=1e-4, atol=1e-7)
    for k, v in g.items():
        if k.startswith("f/"):
            # Adam divides by sqrt(v): where a gradient is ~0 its rounding noise is amplified to ~lr * O(1e-3), so the
            # entity table (most rows untouched by a 90-triple batch) gets an absolute band of 1e-4 on 0.2-sized values
            atol = 1e-4 if k == "f/entity_embed.weight" else 5e-6
            np.testing.assert_allclose(final[k[2:]].detach().cpu().numpy(), v, rtol=rtol_w, atol=atol, err_msg=k)


def _worker(rank, world, port, name, scheme, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import cpu_standins
        cpu_standins.patch_ops()
        g = load_golden(name)
        losses, final, m = replay(g, scheme, torch.device("cpu"), cpu_standins.CpuKernels,
                                  lambda ps, lr: torch.optim.Adam(ps, lr=lr))
        # the optimizer only ever saw this rank's rows of the entity table
        assert m.local.entity_embed.weight.shape[0] == m.part.rows <= m.part.block
        check(g, losses, final)
        q.put((rank, "ok"))
    except Exception as exc:   # noqa: BLE001 -- reported to the parent, which fails the test
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))[-3000:]))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,scheme", [(2, "rows"), (2, "features"), (4, "features"), (4, "rows")])
@pytest.mark.parametrize("name", golden_names("trajectory_"))
def test_sharded_training_replays_the_reference_trajectory(name, world, scheme):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, scheme, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [(r, msg) for r, msg in res if msg != "ok"]
    assert not bad, bad


def test_row_partition_covers_every_row_once():
    from literalkg_amd.distributed import RowPartition
    for n, world in ((200, 2), (201, 4), (7, 8), (1_000_003, 8)):
        parts = [RowPartition(n, r, world) for r in range(world)]
        assert parts[0].lo == 0 and parts[-1].hi == n
        assert all(a.hi == b.lo for a, b in zip(parts, parts[1:]))
        assert all(p.rows <= p.block for p in parts) and sum(p.rows for p in parts) == n


# ----------------------------------------------------------------------------------------------- every mode, every exchange
def _sharded_from_fixture(g, scheme, sparse, kernels, scoring="transr", partition=None):
    from literalkg_amd.distributed import ShardedLiteralKG
    cfg = golden_cfg(g)
    cfg.device = torch.device("cpu")
    n, n_rel = int(g["n"]), int(g["n_rel"])
    state = dict(golden_params(g))
    state["A_in"] = torch.sparse_coo_tensor(torch.from_numpy(g["a_indices"]), torch.from_numpy(g["a_values"]), (n, n)).coalesce()
    num = torch.from_numpy(g["num"]) if "num" in g else None
    txt = torch.from_numpy(g["txt"]) if "txt" in g else None
    # sparse "never-plain": the dense exchange WITHOUT the pipelined passes (pipelined=False: plain all-to-all, one SpMM,
    # plain all-to-all back -- the form a first run on real links bisects against)
    return ShardedLiteralKG.from_full(cfg, n, n_rel, state, num, txt, scoring=scoring, scheme=scheme, device="cpu",
                                      kernels=kernels, sparse_backward=sparse.split("-")[0], partition=partition,
                                      pipelined=not sparse.endswith("-plain"))


def _check_grads(m, g, prefix, world):
    """every gradient the fixture holds: the entity table by this rank's rows, the replicated weights whole (after
    sync_gradients for the partial sums; the heads' own parameters are whole on every rank as they are)"""
    n_checked = 0
    grads = dict(m.local.named_parameters())
    for k, want in g.items():
        if not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        got = grads[name].grad
        assert got is not None, name
        if name == "entity_embed.weight":
            want = want[m.part.lo:m.part.hi]
        np.testing.assert_allclose(got.numpy(), want, rtol=2e-3, atol=2e-6, err_msg=f"{name} (world {world})")
        n_checked += 1
    return n_checked


def _heads_worker(rank, world, port, scheme, sparse, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import cpu_standins
        cpu_standins.patch_ops()
        from literalkg_amd import distributed as D
        K = cpu_standins.CpuKernels
        dev = torch.device("cpu")
        for name in ("encoder_gcn_l2_scale", "encoder_sage_l2", "encoder_gcn_l1_gatemul", "encoder_bi_l1_res"):
            g = load_golden(name)
            m = _sharded_from_fixture(g, scheme, sparse, K, partition="entries" if name == "encoder_sage_l2" else None)
            m.train()
            bh, br, bp, bn = (torch.from_numpy(g[k]) for k in ("bh", "br", "bp", "bn"))
            # pre_training: loss + every gradient of the fixture
            D.TRAFFIC.clear()
            loss = m(bh, br, bp, bn, device=dev, mode="pre_training")
            np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-5, err_msg=name)
            loss.backward()
            m.sync_gradients()
            assert _check_grads(m, g, "g/", world) >= 4
            t = dict(D.TRAFFIC)
            n_layers = int(golden_cfg(g).n_conv_layers)
            if sparse.startswith("always"):  # every aggregation's backward went out as frontier rows, never as a table
                assert t.get("aggregate_backward", 0) == 0 and t.get("frontier_rows", 0) > 0, t
                widest = max(int(golden_cfg(g).embed_dim), int(golden_cfg(g).conv_dim))
                n_msgs = t["frontier_ids"] // 8              # one int64 id per message row
                assert t["frontier_rows"] <= n_msgs * widest * 4, t
            elif scheme == "rows":           # the dense exchange: a padded N x D table per aggregation, (G-1)/G of it leaves
                assert t.get("frontier_rows", 0) == 0, t
                assert t["aggregate_backward"] >= (world - 1) * m.part.block * 4 * n_layers, t
            # fine_tuning: loss + entity gradient
            m.zero_grad(set_to_none=True)
            ft = m(bh, bp, bn, device=dev, mode="fine_tuning")
            np.testing.assert_allclose(float(ft), float(g["ft_loss"]), rtol=1e-5, err_msg=name)
            ft.backward()
            assert _check_grads(m, g, "ft_g/", world) == 1
            # predict / calc_score
            m.eval()
            hid, tid = torch.from_numpy(g["score_heads"]), torch.from_numpy(g["score_tails"])
            with torch.no_grad():
                np.testing.assert_allclose(m.local.calc_score(hid, tid).numpy(), g["score"], rtol=1e-4, atol=1e-4)
                assert np.array_equal(m(hid, tid, device=dev, mode="predict").numpy(), g["predict"])
            assert m(hid, tid, device=dev, mode="no_such_mode") is None
        for name in golden_names("mlp_"):
            g = load_golden(name)
            m = _sharded_from_fixture(g, scheme, sparse, K, scoring="transr" if bool(g["init_mlp"]) else "transe")
            m.train()
            heads, tails = torch.from_numpy(g["heads"]), torch.from_numpy(g["tails"])
            out = m(heads, tails, device=dev, mode="mlp")
            np.testing.assert_allclose(out.detach().numpy().reshape(-1), g["out_train"], rtol=1e-4, atol=1e-5)
            loss = torch.nn.functional.binary_cross_entropy(out.reshape(-1), torch.from_numpy(g["labels"]))
            np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-5)
            loss.backward()
            m.sync_gradients()
            assert _check_grads(m, g, "g/", world) >= 6
            sd = m.local.state_dict()
            for k, want in g.items():
                if k.startswith("after/"):
                    np.testing.assert_allclose(sd[k[6:]].numpy(), want, rtol=1e-4, atol=1e-6, err_msg=k)
            m.eval()
            with torch.no_grad():
                np.testing.assert_allclose(m(heads, tails, device=dev, mode="mlp").numpy().reshape(-1), g["out_eval"],
                                           rtol=1e-4, atol=1e-5)
        q.put((rank, "ok"))
    except Exception as exc:   # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))[-3000:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,scheme,sparse", [(2, "rows", "always"), (2, "rows", "never"), (2, "features", "always"),
                                                  (2, "features", "never"), (4, "rows", "always"), (3, "features", "auto"),
                                                  (2, "features", "never-plain"), (8, "features", "never")])
def test_sharded_module_serves_every_mode(world, scheme, sparse):
    """pre_training / fine_tuning / predict / mlp of the row-sharded module against the reference's fixtures (losses,
    scores, every gradient), with the aggregation's backward as a frontier exchange and as the dense exchange, and the
    bytes each hands to the collective library."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_heads_worker, args=(r, world, port, scheme, sparse, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [(r, msg) for r, msg in res if msg != "ok"]
    assert not bad, bad


def test_row_partition_padded_coordinates():
    from literalkg_amd.distributed import RowPartition
    p = RowPartition(23, 1, 3, [0, 4, 15, 23])
    assert (p.block, p.n_pad, p.lo, p.hi, p.pad_lo, p.rows) == (11, 33, 4, 15, 11, 11)
    ids = torch.arange(23)
    pad = p.to_padded(ids)
    assert pad.tolist() == list(range(0, 4)) + list(range(11, 22)) + list(range(22, 30))
    assert torch.equal(p.from_padded(pad), ids)
    assert RowPartition(20, 0, 4).identity and not p.identity
    h = torch.tensor([0] * 50 + [1] * 2 + list(range(2, 40)))
    b = [RowPartition.balanced(40, r, 4, h) for r in range(4)]
    assert b[0].cuts[0] == 0 and b[0].cuts[-1] == 40 and b[0].cuts == b[3].cuts and b[0].rows < b[3].rows


def _empty_rank_worker(rank, world, port, scheme, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import cpu_standins
        cpu_standins.patch_ops()
        from literalkg_amd.distributed import ShardedLiteralKG
        from oracle import literalkg_oracle as O
        g = load_golden("encoder_gcn_l2_scale")
        cfg = golden_cfg(g)
        cfg.device = torch.device("cpu")
        n, n_rel = int(g["n"]), int(g["n_rel"])
        state = dict(golden_params(g))
        state["A_in"] = torch.sparse_coo_tensor(torch.from_numpy(g["a_indices"]), torch.from_numpy(g["a_values"]), (n, n)).coalesce()
        cuts = [0, n // 2, n // 2, n]                          # the MIDDLE rank owns no row at all
        m = ShardedLiteralKG.from_full(cfg, n, n_rel, state, scoring="transr", scheme=scheme, device="cpu",
                                       kernels=cpu_standins.CpuKernels, partition=cuts, sparse_backward="always").train()
        assert m.part.rows == (0 if rank == 1 else (n // 2 if rank == 0 else n - n // 2))
        batch = [torch.from_numpy(g[k]) for k in ("bh", "br", "bp", "bn")]
        loss = m(*batch, device=cfg.device, mode="pre_training")
        np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-5)
        loss.backward()
        m.sync_gradients()                                      # the bucket has one layout on every rank, rows or no rows
        assert _check_grads(m, g, "g/", world) >= 4
        h, t, r = (torch.from_numpy(g[k]) for k in "htr")
        m(h, t, r, list(range(n_rel)), device=cfg.device, mode="update_att")
        want = O.attention_refresh(n, torch.from_numpy(g["p/entity_embed.weight"]), torch.from_numpy(g["p/relation_embed.weight"]),
                                   h, t, r).coalesce()
        got = m.full_state_dict()["A_in"].coalesce()
        assert torch.equal(got.indices(), want.indices())
        torch.testing.assert_close(got.values(), want.values(), rtol=1e-4, atol=1e-6)
        q.put((rank, "ok"))
    except Exception as exc:   # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))[-3000:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scheme", ["rows", "features"])
def test_a_rank_without_rows_takes_part_in_every_collective(scheme):
    """More ranks than row ranges worth having (world > N in the limit): a rank that owns NO entity row still runs every
    exchange -- table gathers, frontier counts, the fixed weight-gradient bucket -- and the results are the reference's."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_rank_worker, args=(r, 3, port, scheme, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [(r, msg) for r, msg in res if msg != "ok"]
    assert not bad, bad
