"""GPU: the multi-GPU training step of literalkg_amd/distributed.py on the REAL HIP kernels -- 2 ranks sharing the one
GPU of the test box (gloo transport, host-staged exchange), both aggregation schemes, fused Adam over each rank's
shard -- replays the reference's training trajectories (tests/golden/trajectory_*.npz); a second test runs the same
over RCCL with one GPU per rank and is skipped where fewer than two GPUs are visible."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, name, scheme, backend, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    nccl = backend == "nccl"
    dev = torch.device("cuda", rank if nccl else 0)
    torch.cuda.set_device(dev)
    if nccl:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import test_distributed_gloo as T
        from literalkg_amd.optim import Adam
        g = load_golden(name)
        losses, final, m = T.replay(g, scheme, dev, None, lambda ps, lr: Adam(ps, lr=lr))
        T.check(g, losses, final)
        # same batch, general (ungrouped) TransR path: same losses
        losses2, _, _ = T.replay(g, scheme, dev, None, lambda ps, lr: Adam(ps, lr=lr), group_reuse=False)
        assert max(abs(a - b) for a, b in zip(losses, losses2)) < 1e-5
        q.put((rank, "ok"))
    except Exception as exc:   # noqa: BLE001 -- reported to the parent, which fails the test
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))[-3000:]))
    finally:
        dist.destroy_process_group()


def _run(name, scheme, backend, world=2):
    import __graft_entry__ as ge
    ge.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, scheme, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    bad = [(r, msg) for r, msg in res if msg != "ok"]
    assert not bad, bad


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scheme", ["rows", "features"])
@pytest.mark.parametrize("name", golden_names("trajectory_"))
def test_two_ranks_on_one_gpu_replay_the_reference_trajectory(gpu_device, name, scheme):
    _run(name, scheme, "gloo")


@pytest.mark.timeout(300)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (the test boxes have one)")
@pytest.mark.parametrize("scheme", ["rows", "features"])
def test_two_ranks_over_rccl_replay_the_reference_trajectory(gpu_device, scheme):
    _run("trajectory_gcn_l2_gatemul_scale", scheme, "nccl")


def _frontier_worker(rank, world, port, scheme, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from types import SimpleNamespace
        import literalkg_amd as L
        from literalkg_amd import distributed as D, io
        from literalkg_amd.synth import make_batch, make_kg
        n, dim = 100_000, 64
        h, t, r = make_kg(n, 400_000, seed=3)
        cfg = SimpleNamespace(use_pretrain=0, device=dev, embed_dim=dim, relation_dim=dim, scale_gat_dim=None,
                              use_residual=False, alpha=0.1, lamda=0.5, aggregation_type="gcn", n_conv_layers=2,
                              conv_dim=dim, mess_dropout=0.0, kg_l2loss_lambda=1e-5, fine_tuning_l2loss_lambda=1e-5,
                              pre_training_neg_rate=3, fine_tuning_neg_rate=3, num_lit_dim=2, txt_lit_dim=300,
                              use_num_lit=False, use_txt_lit=False, milestone_score=0.5, n_mlp_layers=2, mlp_hidden_dim=64)
        torch.manual_seed(5)
        full = L.LiteralKG(cfg, n, 16, io.initial_a_in(n, h, t, r))
        state = dict(full.state_dict())
        batch = [torch.from_numpy(x).to(dev) for x in make_batch(n, 30, 3, seed=1)]
        res = {}
        for mode in ("auto", "never"):
            m = D.ShardedLiteralKG.from_full(cfg, n, 16, state, scheme=scheme, device=dev, sparse_backward=mode).train()
            D.TRAFFIC.clear()
            loss = m(*batch, device=dev, mode="pre_training")
            loss.backward()
            m.sync_gradients()
            torch.cuda.synchronize()
            res[mode] = (float(loss.detach()), {k: p.grad.clone() for k, p in m.local.named_parameters() if p.grad is not None},
                         dict(D.TRAFFIC))
        (la, ga, ta), (ln, gn, tn) = res["auto"], res["never"]
        assert abs(la - ln) <= 1e-6 * max(1.0, abs(ln))
        assert ga.keys() == gn.keys() and "entity_embed.weight" in ga
        for k in ga:
            torch.testing.assert_close(ga[k], gn[k], rtol=2e-4, atol=1e-7, msg=k)
        # the frontier exchange moved rows, not tables: both aggregations' backward in a small fraction of the dense bytes
        assert ta.get("aggregate_backward", 0) == 0 and ta["frontier_rows"] > 0, ta
        assert tn.get("frontier_rows", 0) == 0 and tn["aggregate_backward"] > 0, tn
        assert ta["frontier_rows"] + ta["frontier_ids"] < 0.25 * tn["aggregate_backward"], (ta, tn)
        q.put((rank, "ok"))
    except Exception as exc:   # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))[-3000:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scheme", ["rows", "features"])
def test_frontier_exchange_equals_the_dense_exchange_on_the_kernels(gpu_device, scheme):
    """100 k entities, 2 layers, 2 ranks on the one GPU: the row sets the loss leaves on its gradient reach both
    aggregations' backward (tags through act_ln / Linear on the real kernels), which then exchange frontier rows; loss and
    every gradient equal the dense exchange's."""
    import __graft_entry__ as ge
    ge.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_frontier_worker, args=(r, 2, port, scheme, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    bad = [(r, msg) for r, msg in res if msg != "ok"]
    assert not bad, bad
