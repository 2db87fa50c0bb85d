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


def _decoder_branch():
    from scda_amd.dropin.models.faster_rcnn import common_net as cn
    seq = nn.Sequential(cn.LinUnsRes_cluster(16, 64, 64, 4), cn.INSResBlock(16, 16, dropout=0.5), cn.INSResBlock(16, 16, dropout=0.5),
                        cn.LeakyReLUConvTranspose2d_2(16, 8, kernel_size=3, stride=1, padding=1, output_padding=0),
                        cn.LeakyReLUConvTranspose2d_2(8, 4, kernel_size=3, stride=1, padding=1, output_padding=0))
    return cn, seq


def test_decoder_pairs_each_interpolate_with_the_norm_in_front():
    """common_net.pair_decoder_upsamples: the first up-sampling block's Interpolate is fed by the LAST residual block (whose fused tail
    holds the norm), the second one's by the first block's instance norm; residual blocks that feed another residual block stay unpaired"""
    from scda_amd import layers as L
    cn, seq = _decoder_branch()
    assert cn.pair_decoder_upsamples(seq) == 2
    up1, up2 = seq[3].model[0].up, seq[4].model[0].up
    assert L.my_upsample(seq[2]) is up1 and L.my_upsample(seq[3].model[2]) is up2
    assert L.my_upsample(seq[1]) is None and L.my_upsample(seq[4].model[2]) is None
    plain = cn.INSResBlock(16, 16, dropout=0.0)            # no dropout: no fused tail, nothing to pair
    seq2 = nn.Sequential(plain, cn.LeakyReLUConvTranspose2d_2(16, 8, kernel_size=3, stride=1, padding=1, output_padding=0))
    assert cn.pair_decoder_upsamples(seq2) == 0 and L.my_upsample(plain) is None


def test_norm_upsample_pairing_under_copies_and_the_off_switch(monkeypatch):
    from scda_amd import layers as L
    cn, seq = _decoder_branch()
    cn.pair_decoder_upsamples(seq)
    up1 = seq[3].model[0].up
    up1.expect_upsampled((1, 2, 3, 4))
    assert L.my_upsample(seq[2]) is up1 and up1._upsampled_shape is None      # asking for the pair clears an announcement left behind
    dup = copy.deepcopy(seq)
    assert L.my_upsample(dup[2]) is None and dup[3].model[0].up._producer is None and L.my_upsample(seq[2]) is up1
    assert cn.pair_decoder_upsamples(dup) == 2 and L.my_upsample(dup[2]) is dup[3].model[0].up
    again = pickle.loads(pickle.dumps(seq))
    assert L.my_upsample(again[2]) is None and L.my_upsample(again[3].model[2]) is None
    replica = copy.copy(seq[2]); replica.__dict__ = dict(seq[2].__dict__)     # a shallow replica must not announce to the original's module
    assert replica._up_ref is not None and L.my_upsample(replica) is None
    monkeypatch.setenv("SCDA_NO_NORM_UP_FUSION", "1")
    assert L.my_upsample(seq[2]) is None
