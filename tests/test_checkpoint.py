"""checkpoint / pre-training I/O (scda_amd/checkpoint.py): reference key layout in, flat-bucket views intact, full round trip"""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from test_host_functions import CFG  # noqa: E402


def _detector():
    from scda_amd.dropin.models.faster_rcnn.vgg_adver_expansion_cluster import vgg16
    torch.manual_seed(3)
    return vgg16(cfg=dict(CFG['shared'], gan_model_flag=2))


def test_load_pretrain_torchvision_layout_into_flat_views():
    from scda_amd import checkpoint as C
    from scda_amd.flat import FlatParams
    det = _detector()
    flat = FlatParams(det)
    g = torch.Generator().manual_seed(9)
    # torchvision vgg16: features.{0..28}.*, classifier.{0,3,6}.* ; classifier.6 (1000-way) has no counterpart, as in the reference run
    tv = {k: torch.randn(v.shape, generator=g) for k, v in det.state_dict().items() if k.startswith(('features.', 'classifier.'))}
    tv['classifier.6.weight'] = torch.randn(1000, 4096, generator=g)
    tv['classifier.6.bias'] = torch.randn(1000, generator=g)
    before = det.state_dict()['rpn_head.conv3x3.weight'].clone()
    C.load_pretrain(det, {'module.' + k: v for k, v in tv.items()})
    sd = det.state_dict()
    for k, v in tv.items():
        if k in sd:
            assert torch.equal(sd[k], v), k
    assert torch.equal(sd['rpn_head.conv3x3.weight'], before)            # untouched
    lo, hi = flat.data.data_ptr(), flat.data.data_ptr() + flat.numel * 4
    assert all(lo <= p.data_ptr() < hi for p in det.parameters())        # still views of the bucket
    w = det.features[0].weight
    off = (w.data_ptr() - lo) // 4
    assert torch.equal(flat.data[off:off + w.numel()].view_as(w), tv['features.0.weight'])
    with pytest.raises(AssertionError):
        C.load_pretrain(det, {'nothing.matches': torch.zeros(1)})
    C.load_pretrain(det, {'state_dict': {'features.0.bias': torch.ones(64)}})   # *.tar layout
    assert float(det.features[0].bias.sum()) == 64.0


def test_save_restore_round_trip(tmp_path):
    from scda_amd import checkpoint as C
    from scda_amd.flat import FlatAdam, FlatParams
    from scda_amd.train_step import builder_gan

    def make(seed):
        torch.manual_seed(seed)
        det = _detector()
        dis, dec, dis_patch = builder_gan()
        tr = types.SimpleNamespace(model=det, dec=dec, dis=dis, dis_patch=dis_patch)
        tr.flat = {k: FlatParams(m) for k, m in (('det', det), ('dec', dec), ('dis', dis), ('dis_patch', dis_patch))}
        tr.opt = {k: FlatAdam(f, 1e-3) for k, f in tr.flat.items()}
        return tr

    a = make(1)
    for k, o in a.opt.items():
        o.step_count = 7
        o.exp_avg.normal_(); o.exp_avg_sq.uniform_()
    for m in (a.dec, a.dis, a.dis_patch):
        for p in m.parameters():
            p.data.normal_()
    path = str(tmp_path / 'checkpoint_e3.pth')
    C.save_checkpoint(a, path, epoch=3, best_recall=0.5)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert {'epoch', 'arch', 'state_dict', 'best_recall', 'optimizer'} <= set(ck)           # the reference's fields
    assert list(ck['state_dict'].keys()) == list(a.model.state_dict().keys())
    b = make(2)
    epoch, best, arch = C.restore(b, path)
    assert (epoch, best, arch) == (3, 0.5, 'vgg16_FasterRCNN')
    for ma, mb in ((a.model, b.model), (a.dec, b.dec), (a.dis, b.dis), (a.dis_patch, b.dis_patch)):
        for (k, va), (_, vb) in zip(ma.state_dict().items(), mb.state_dict().items()):
            assert torch.equal(va, vb), k
    for k in a.opt:
        assert b.opt[k].step_count == 7 and torch.equal(a.opt[k].exp_avg, b.opt[k].exp_avg) and torch.equal(a.opt[k].exp_avg_sq, b.opt[k].exp_avg_sq)
    # a reference-made file (detector only, torch Adam state dict) restores the detector and leaves the rest alone
    c = make(4)
    dec_before = {k: v.clone() for k, v in c.dec.state_dict().items()}
    C.restore(c, {'epoch': 1, 'arch': 'vgg16_FasterRCNN', 'best_recall': 0.1, 'optimizer': {'state': {}, 'param_groups': []},
                  'state_dict': {'module.' + k: v for k, v in a.model.state_dict().items()}})
    assert torch.equal(c.model.state_dict()['fc_rcnn_cls.weight'], a.model.state_dict()['fc_rcnn_cls.weight'])
    assert all(torch.equal(v, c.dec.state_dict()[k]) for k, v in dec_before.items())
    assert c.opt['det'].step_count == 0
    # a head with another class count must not load silently (the reference's load_state_dict raises on size mismatches)
    bad = {k: v for k, v in a.model.state_dict().items()}
    bad['fc_rcnn_cls.weight'] = torch.zeros(21, 4096)
    with pytest.raises(ValueError, match='fc_rcnn_cls.weight'):
        C.load_pretrain(c.model, bad)
    C.load_pretrain(c.model, bad, allow_mismatch=True)
    # ... nor optimiser moments recorded under other hyper-parameters
    ck['optimizer']['weight_decay'] = 0.5
    with pytest.raises(ValueError, match='weight_decay'):
        C.restore(make(5), ck)


@pytest.mark.gpu
def test_resumed_trainer_continues_bit_identically(cuda, tmp_path):
    """step, save, restore into a fresh trainer: the next step (same inputs, same RNG) gives the same losses to the last bit
    (all kernels are deterministic; Adam moments and BN statistics travel with the checkpoint)"""
    import copy
    from scda_amd import checkpoint as C
    from scda_amd.train_step import ScdaTrainer
    cfg = copy.deepcopy(CFG)
    cfg['shared']['gan_model_flag'] = 2
    H, W = 256, 512
    g = torch.Generator().manual_seed(5)
    src = torch.randn(1, 3, H, W, generator=g).clamp_(-1, 1).to(cuda)
    tgt = torch.randn(1, 3, H, W, generator=g).clamp_(-1, 1).to(cuda)
    gts = torch.tensor([[[30., 40., 200., 180., 3.], [250., 60., 400., 200., 5.]]])
    info = torch.tensor([[H, W, 1.0]])

    def step(tr, seed):
        np.random.seed(seed); torch.manual_seed(seed)
        out = tr.step(src, gts, info, tgt)
        torch.cuda.synchronize()
        return {k: float(v) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1}

    torch.manual_seed(0)
    a = ScdaTrainer(cfg, cuda, lr=1e-3, new_w=W, new_h=H)
    step(a, 1)
    C.save_checkpoint(a, str(tmp_path / 'ck.pth'), epoch=1)
    want = step(a, 2)
    torch.manual_seed(123)                      # different initial weights: everything must come from the file
    b = ScdaTrainer(cfg, cuda, lr=1e-3, new_w=W, new_h=H)
    C.restore(b, str(tmp_path / 'ck.pth'))
    got = step(b, 2)
    assert got == want, {k: (got[k], want[k]) for k in want if got[k] != want[k]}
