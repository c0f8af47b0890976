"""Synthetic KGs of the BASELINE.json shapes.  Host-side numpy, seeded.

heads : 'zipf'    h = perm[floor(N * u^1.75)]  -- a power-law out-degree (density ~ x^-0.43 over the
                  entity rank, the real files' skew: median ~0.6x mean, heavy head), out-degree
                  clipped at 4096 by re-drawing the excess edges' heads uniformly;
        'uniform' h ~ U[0, N)
tails : t ~ U[0, N);  relations r ~ U[0, R), R = 16
(h, r, t) duplicates are removed like the loader does (dataloader.py:189) and re-drawn until exactly E
distinct triples exist; ~0.02 % of them are forced duplicate (h, t) pairs under a second relation so
that the merge path of update_att (coalesce, model.py:468-470) is always exercised.
"""
from __future__ import annotations

import numpy as np

N_REL = 16
DEGREE_CAP = 4096


def _draw(rng, n, e, skew, perm):
    if skew == "zipf":
        h = perm[np.minimum((n * rng.random(e) ** 1.75).astype(np.int64), n - 1)]
    elif skew == "uniform":
        h = rng.integers(0, n, e, dtype=np.int64)
    else:
        raise ValueError(skew)
    return h, rng.integers(0, n, e, dtype=np.int64), rng.integers(0, N_REL, e, dtype=np.int64)


def _clip_degree(rng, n, h):
    deg = np.bincount(h, minlength=n)
    if deg.max() <= DEGREE_CAP:
        return h
    order = np.argsort(h, kind="stable")
    hs = h[order]
    first = np.r_[0, np.flatnonzero(np.diff(hs)) + 1]
    rank = np.arange(len(hs)) - np.repeat(first, np.diff(np.r_[first, len(hs)]))
    over = order[rank >= DEGREE_CAP]
    h = h.copy()
    h[over] = rng.integers(0, n, len(over), dtype=np.int64)
    return h


def make_kg(n_entities: int, n_edges: int, skew: str = "zipf", seed: int = 2022, dup_frac: float = 2e-4):
    """Returns (h, t, r) int64 arrays of exactly n_edges distinct (h, r, t) triples."""
    rng = np.random.default_rng(seed)
    n, e = int(n_entities), int(n_edges)
    perm = rng.permutation(n) if skew == "zipf" else None
    n_dup = int(round(e * dup_frac))
    keys = np.empty(0, np.int64)
    want = e - n_dup
    while len(keys) < want:
        h, t, r = _draw(rng, n, int((want - len(keys)) * 1.02) + 16, skew, perm)
        keys = np.unique(np.concatenate([keys, (h * N_REL + r) * n + t]))
    if len(keys) > want:
        keys = keys[rng.permutation(len(keys))[:want]]
    if n_dup:
        src = keys[rng.choice(len(keys), n_dup, replace=False)]
        hh, tt = src // (n * N_REL), src % n
        r2 = ((src // n) % N_REL + 1 + rng.integers(0, N_REL - 1, n_dup)) % N_REL
        keys = np.unique(np.concatenate([keys, (hh * N_REL + r2) * n + tt]))
    t = keys % n
    hr = keys // n
    h, r = hr // N_REL, hr % N_REL
    if skew == "zipf":
        h2 = _clip_degree(rng, n, h)
        if h2 is not h:   # clipping can (rarely) recreate a duplicate triple: drop those
            k2 = np.unique((h2 * N_REL + r) * n + t)
            t, hr = k2 % n, k2 // n
            h, r = hr // N_REL, hr % N_REL
    return h.astype(np.int64), t.astype(np.int64), r.astype(np.int64)


def make_kg_device(n_entities: int, n_edges: int, skew: str, seed: int, device):
    """(h, t, r) int64 DEVICE tensors drawn like make_kg's heads / tails / relations, in milliseconds instead of the minute
    numpy takes for 100 M triples: for timing shapes far beyond the parity-tested ones.  Out-degrees clipped at DEGREE_CAP like
    make_kg's; no (h, r, t) de-duplication (the structure build merges equal (h, t) pairs)."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    n, e = int(n_entities), int(n_edges)
    if skew == "zipf":
        perm = torch.randperm(n, generator=gen, device=device)
        u = torch.rand(e, generator=gen, device=device, dtype=torch.float64)
        h = perm[(n * u ** 1.75).long().clamp_(max=n - 1)]
    elif skew == "uniform":
        h = torch.randint(0, n, (e,), generator=gen, device=device)
    else:
        raise ValueError(skew)
    if skew == "zipf":             # make_kg's degree clip: the edges beyond DEGREE_CAP of a head get a uniformly drawn head
        order = torch.argsort(h, stable=True)
        hs = h[order]
        first = torch.ones(e, dtype=torch.bool, device=device)
        first[1:] = hs[1:] != hs[:-1]
        start = torch.cummax(torch.where(first, torch.arange(e, device=device), torch.zeros((), dtype=torch.int64, device=device)), 0).values
        over = order[(torch.arange(e, device=device) - start) >= DEGREE_CAP]
        h[over] = torch.randint(0, n, (over.numel(),), generator=gen, device=device)
        del order, hs, first, start, over
    t = torch.randint(0, n, (e,), generator=gen, device=device)
    r = torch.randint(0, N_REL, (e,), generator=gen, device=device)
    return h, t, r


def make_batch(n_entities: int, groups: int, neg_rate: int, seed: int = 2022):
    """G groups x K negatives in the layout of generate_kg_batch (dataloader.py:283-330): each head
    contributes K consecutive entries with identical (h, r, t+) and K distinct t-."""
    rng = np.random.default_rng(seed + 1)
    h = np.repeat(rng.integers(0, n_entities, groups), neg_rate)
    r = np.repeat(rng.integers(0, N_REL, groups), neg_rate)
    p = np.repeat(rng.integers(0, n_entities, groups), neg_rate)
    n = rng.integers(0, n_entities, groups * neg_rate)
    return h, r, p, n


def xavier_table(n: int, d: int, device, seed: int = 2022):
    """xavier_uniform_ like model.py:233-237, generated on the target device."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    bound = (6.0 / (n + d)) ** 0.5
    return (torch.rand((n, d), generator=gen, device=device, dtype=torch.float32) * 2 - 1) * bound
