"""The SCDA nets on the RECTANGULAR maps of BASELINE.json configs[3] / [4] (2048-d RoI features unfolded to 32 x 64, 128 x 256
reconstructions and crops -- scda_amd/resnet_config.py) against the torch-CPU oracle's nets built with the same geometry: decoder,
image discriminators and patch discriminator, forward and every parameter / input gradient.  Dropout masks and activation sign
selections are the oracle's (replayed), so what is compared is the kernels' arithmetic on non-square maps: 1e-4 relative L2.
(The square 64 x 64 / 256 x 256 geometry of the VGG configuration is covered by the whole-iteration tests.)"""
import numpy as np
import pytest
import torch

import model_common as mc

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_scda_nets_on_rectangular_maps_match_oracle(cuda):
    from oracle import torch_ref as R
    from scda_amd import autograd_ops as A, layers as L, resnet_config as RC
    from scda_amd.train_step import builder_gan
    import seeded_init as si
    C, T, (fw, fh), (rh, rw) = 4, 128, RC.FEAT_HW, RC.RECON_HW
    g = torch.Generator().manual_seed(5)
    src_patch = torch.randn(C, T, fw * fh, generator=g)
    tgt_patch = torch.randn(C, T, fw * fh, generator=g)
    x_small = torch.randn(C, 3, rh, rw, generator=g).clamp_(-1, 1)
    up_dec = [torch.randn(C, 3, rh, rw, generator=g) for _ in range(2)]
    n_dis = (rh // 8) * (rw // 8)
    up_dis = [torch.randn(C, n_dis, generator=g) for _ in range(4)]
    up_patch = torch.randn(C, 2 * 2 * T, generator=g)

    def run(dec, dis, dis_patch, dev):
        a, b = src_patch.detach().clone().to(dev).requires_grad_(), tgt_patch.detach().clone().to(dev).requires_grad_()
        ra, rb = dec(a, b)
        assert tuple(ra.shape) == (C, 3, rh, rw)
        da, db = dis(ra, rb)                                  # through the decoders' outputs
        xa, xb = dis(x_small.to(dev), x_small.flip(0).to(dev))
        pp = dis_patch(a)
        assert tuple(da.shape) == (C, n_dis) and tuple(pp.shape) == (C, 2 * 2 * T)
        loss = ((ra * up_dec[0].to(dev)).sum() + (rb * up_dec[1].to(dev)).sum() + (da * up_dis[0].to(dev)).sum()
                + (db * up_dis[1].to(dev)).sum() + (xa * up_dis[2].to(dev)).sum() + (xb * up_dis[3].to(dev)).sum()
                + (pp * up_patch.to(dev)).sum())
        loss.backward()
        outs = {'ra': ra, 'rb': rb, 'da': da, 'db': db, 'xa': xa, 'xb': xb, 'pp': pp, 'd_src_patch': a.grad, 'd_tgt_patch': b.grad}
        return {k: v.detach().cpu() for k, v in outs.items()}

    # ---- oracle, CPU
    torch.manual_seed(1)
    ref = (R.RefDecoder(T, 3, 3, 0.5, C, w=fw, h=fh), R.RefDis(32, 3), R.RefDisPatch(T, 2 * T, C, w=fw, h=fh))
    for m, seed in zip(ref, (12, 13, 14)):
        si.seeded_reinit(m, seed, 'gan')
        m.train()
    R.RecordingDropout.tape = []
    rec = R.SelectionRecorder()
    handles = rec.attach(*ref)
    try:
        torch.manual_seed(31)
        torch.set_num_threads(8)
        want = run(*ref, torch.device('cpu'))
    finally:
        rec.detach(handles)
        R.SelectionRecorder.active = None
        torch.set_num_threads(1)
    tape = list(R.RecordingDropout.tape)
    R.RecordingDropout.tape = None
    assert len(tape) == 6                                       # three residual blocks per decoder

    # ---- product, device: same weights by key
    torch.manual_seed(1)
    dis, dec, dis_patch = builder_gan(C, T, 256, neww=fw, newh=fh)
    for m, seed, r in zip((dec, dis, dis_patch), (12, 13, 14), ref):
        si.seeded_reinit(m, seed, 'gan')
        assert sorted(m.state_dict()) == sorted(r.state_dict())
        m.to(cuda).train()
    rp = mc.ReplaySource(rec, cuda)
    with mc.probed(dropout_masks=lambda shape, p, device: tape.pop(0).to(device), replay=rp):
        got = run(dec, dis, dis_patch, cuda)
        torch.cuda.synchronize()
        used = rp.used
    assert not tape and used >= 20, (len(tape), used)
    for k in want:
        assert rel_l2(got[k], want[k]) <= 1e-4, (k, rel_l2(got[k], want[k]))
    worst = ('', 0.0)
    n = 0
    for m, r in zip((dec, dis, dis_patch), ref):
        rp = dict(r.named_parameters())
        scale = max(float(p.grad.abs().max()) for p in rp.values() if p.grad is not None)
        for k, p in m.named_parameters():
            if rp[k].grad is None or float(rp[k].grad.abs().max()) < 1e-6 * scale:   # conv biases in front of a norm: mathematically zero
                continue
            e = rel_l2(p.grad, rp[k].grad)
            n += 1
            if e > worst[1]:
                worst = (k, e)
    print("rectangular SCDA nets: %d gradient tensors, worst %s %.2e" % (n, worst[0], worst[1]))
    assert n >= 40 and worst[1] <= 1e-4, worst
    bn = [k for k in dis_patch.state_dict() if k.endswith('running_mean')]
    assert bn and all(rel_l2(dis_patch.state_dict()[k], ref[2].state_dict()[k]) <= 1e-5 for k in bn)
