"""CPU restatement of the reference's KG batch sampler -- TEST INFRASTRUCTURE ONLY (checker for the device-side
sampler ``lkg_sample_kg_batch``; never imported by literalkg_amd/).

Restates /root/reference/dataloader.py:249-330 with the same data structures and the same accept / reject rules:
  sample_pos_triples_for_head (249-266)  one index drawn uniformly over the head's triple list kg_dict[h] = [(t, r) ..]
  sample_neg_triples_for_head (268-281)  candidates drawn uniformly from ``training_tails`` (the tail COLUMN of the
                                         de-duplicated triples, dataloader.py:358: a tail is drawn in proportion to its
                                         multiplicity), rejected when (candidate, relation) is a positive of the head or
                                         when the candidate already is one of this head's negatives
  generate_kg_batch (283-316)            heads: random.sample without replacement when enough heads exist, else
                                         random.choice per slot (that branch raises TypeError in the reference under
                                         Python 3 -- random.choice on a dict_keys view, dataloader.py:290-291 -- it is
                                         restated here as written, on the materialised keys); per head 1 positive and
                                         neg_rate negatives
  generate_batch_by_neg_rate (318-330)   h / r / t+ repeated neg_rate times, element by element
Pinned: the functions draw from the same generators in the same order as the reference (the global ``random`` module
and ``numpy.random``), so under equal seeds they reproduce the reference's batch bit for bit --
tests/golden/sampler_ref_batch.npz was written by oracle/gen_golden.py from the reference's own methods and
tests/test_oracle_golden.py replays it.  Those streams cannot be reproduced on the device, so the HIP sampler's parity
with this oracle is distributional (tests/test_gpu_parity.py::test_kg_batch_sampler_distribution_matches_the_oracle).
"""
import collections
import random

import numpy as np


def build_kg_dict(h, t, r):
    """train_kg_dict of DataLoader.construct_data (dataloader.py:392-402): head -> [(tail, relation), ...]"""
    kg = collections.defaultdict(list)
    for hh, tt, rr in zip(h.tolist(), t.tolist(), r.tolist()):
        kg[hh].append((tt, rr))
    return kg


def sample_pos_triples_for_head(kg_dict, head, n_sample, nprand=np.random):
    pos = kg_dict[head]
    rels, tails = [], []
    while len(rels) < n_sample:
        i = nprand.randint(low=0, high=len(pos), size=1)[0]
        tail, rel = pos[i]
        if rel not in rels and tail not in tails:
            rels.append(rel)
            tails.append(tail)
    return rels, tails


def sample_neg_triples_for_head(kg_dict, head, relation, n_sample, training_tails, rnd=random):
    pos = kg_dict[head]
    out = []
    while len(out) < n_sample:
        tail = rnd.choice(training_tails)
        if (tail, relation) not in pos and tail not in out:
            out.append(tail)
    return out


def generate_kg_batch(kg_dict, batch_size, neg_rate, training_tails, rnd=random, nprand=np.random):
    """Returns (h, r, pos_t, neg_t) int64 arrays of (batch_size // neg_rate) * neg_rate entries.  ``kg_dict`` = the
    dict the batch's heads are drawn from (the reference passes an epoch's sampled dict, main_pretraining.py:93-101)."""
    pool = tuple(kg_dict.keys())          # random.sample(dict_keys) materialises the keys the same way (Python 3.10)
    groups = int(batch_size / neg_rate)
    if groups <= len(pool):
        batch_head = rnd.sample(pool, groups)
    else:
        batch_head = [rnd.choice(pool) for _ in range(groups)]
    rel, pos, neg = [], [], []
    for hh in batch_head:
        rr, pp = sample_pos_triples_for_head(kg_dict, hh, 1, nprand)
        rel += rr
        pos += pp
        neg += sample_neg_triples_for_head(kg_dict, hh, rr[0], neg_rate, training_tails, rnd)
    rep = lambda x: np.repeat(np.asarray(x, np.int64), neg_rate)
    return rep(batch_head), rep(rel), rep(pos), np.asarray(neg, np.int64)
