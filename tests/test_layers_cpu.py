"""Host-side behaviour of scda_amd.layers that needs no GPU: the conv -> pool fusion hand-over under copies of a model."""
import copy
import pickle

import torch.nn as nn


def _planned():
    from scda_amd import layers as L, autograd_ops as A
    seq = nn.Sequential(L.Conv2d(8, 8, 3, padding=1, fused_act=A.ACT_RELU), L.MaxPool2x2(), L.Conv2d(8, 8, 3, padding=1, fused_act=A.ACT_RELU))
    assert L.plan_act_fusion(seq) == 1
    return seq


def test_fusion_pairing_is_checked_from_both_sides():
    seq = _planned()
    conv, pool = seq[0], seq[1]
    assert conv.pool_next and conv._my_pool() is pool


def test_deepcopy_and_pickle_drop_the_pairing_and_can_be_replanned():
    from scda_amd import layers as L
    seq = _planned()
    dup = copy.deepcopy(seq)                       # weak references do not travel: the copy runs un-fused ...
    assert not dup[0].pool_next and dup[0]._my_pool() is None and dup[1]._producer is None
    assert seq[0]._my_pool() is seq[1]             # ... and the original is untouched
    assert L.plan_act_fusion(dup) == 1 and dup[0]._my_pool() is dup[1]      # ... until it is planned again
    again = pickle.loads(pickle.dumps(seq))        # torch.save(model) pickles the module objects
    assert not again[0].pool_next and again[0]._my_pool() is None


def test_a_shallow_replica_does_not_signal_the_original_pool():
    """nn.DataParallel replicas copy __dict__: the replica's conv still holds the original's weak reference, but the original's pool
    names the original conv as its producer -- the replica must not announce a pooled tensor to a pool it does not feed"""
    seq = _planned()
    replica = copy.copy(seq[0])
    replica.__dict__ = dict(seq[0].__dict__)
    assert replica._pool_ref is not None and replica._my_pool() is None
    assert seq[1]._pooled_shape is None


def test_a_stale_announcement_is_cleared_by_the_next_unfused_call():
    import torch
    seq = _planned()
    seq[1].expect_pooled((1, 8, 2, 2))             # left behind by a call that raised between the conv and its pool
    try:
        seq[0](torch.zeros(1, 8, 4, 4))            # CPU tensor: the product path refuses it (no CPU fallback) ...
    except Exception:
        pass
    assert seq[1]._pooled_shape is None            # ... but the announcement is gone before anything else happens
