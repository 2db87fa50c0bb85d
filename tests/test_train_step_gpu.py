"""The four-phase SCDA iteration on the MI355X (scda_amd.train_step.ScdaTrainer, all compute through the C ABI)
against the CPU oracle (oracle/torch_ref.py, itself pinned to the reference's train()).

Same seeded weights / inputs / numpy RNG stream; the oracle's dropout keep-masks are replayed on the device.
Tolerances: losses 1e-4 relative (fp32, different summation order across ~1e9 MACs per output); gradients
compared per tensor as max|diff| / max|ref| < 2e-3."""
import numpy as np
import pytest
import torch

import model_common as mc

pytestmark = pytest.mark.gpu


def build_product():
    from scda_amd.dropin.models.faster_rcnn.vgg_adver_expansion_cluster import vgg16
    from scda_amd.train_step import builder_gan
    det = vgg16(cfg=dict(mc.CFG['shared'], gan_model_flag=2))
    dis, dec, dis_patch = builder_gan()
    return det, dec, dis, dis_patch


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_iteration_matches_oracle(cuda):
    from scda_amd import layers as L
    from scda_amd.train_step import ScdaTrainer
    H, W, lr = 256, 512, 1e-3
    ref, ref_models, masks = mc.oracle_iteration(H, W, lr=lr, record_masks=True, capture=True)

    torch.manual_seed(1)
    models = mc.seeded_models(build_product)
    tr = ScdaTrainer(mc.CFG, cuda, lr=lr, new_w=W, new_h=H, models=models)
    tr.capture = True
    src, tgt, gts, info = mc.seeded_inputs(H, W)
    tape = list(masks)

    def replay(shape, p, device):
        m = tape.pop(0)
        assert tuple(m.shape) == tuple(shape), (m.shape, shape)
        return m.to(device)

    L.Dropout.mask_source = replay
    try:
        np.random.seed(mc.SEEDS['numpy'])
        out = tr.step(src.to(cuda), gts, info, tgt.to(cuda))
        torch.cuda.synchronize()
    finally:
        L.Dropout.mask_source = None
    assert not tape, "the device consumed fewer dropout masks than the oracle drew"

    for k in ('rpn_cls', 'rpn_loc', 'rcnn_cls', 'rcnn_loc', 'adloss', 'dis_patch_loss', 'recon_loss', 'fake_loss1_source',
              'fake_loss_target', 'fake_loss_source', 'loss'):
        a, b = float(out[k]), float(ref[k])
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (k, a, b)
    assert abs(float(out['rpn_acc'][0]) - float(ref['rpn_acc'][0])) < 0.5
    assert abs(float(out['rcnn_acc'][0]) - float(ref['rcnn_acc'][0])) < 0.5

    # per-phase gradients (the oracle's are captured right after the phase's backward)
    worst = {}
    for name in ('dis', 'dis_patch', 'dec', 'det'):
        rg, pg = ref['_trace'][name], tr.trace[name]
        assert set(rg) == set(pg)
        worst[name] = max(rel(pg[k], rg[k]) for k in rg)
    assert all(v < 2e-3 for v in worst.values()), worst

    # parameters after the step: Adam's first step moves every weight by ~lr*sign(g); allow sign noise on |g|~0
    for pm, rm in zip((tr.model, tr.dec, tr.dis, tr.dis_patch), ref_models):
        rsd = rm.state_dict()
        for k, v in pm.state_dict().items():
            if v.dtype.is_floating_point:
                d = (v.detach().cpu().double() - rsd[k].double()).abs()
                assert float(d.mean()) < 0.02 * lr and float(d.max()) <= 2.001 * lr, (k, float(d.mean()), float(d.max()))
    dp = tr.dis_patch.state_dict()
    assert int(dp['model_A_patch.0.model.1.num_batches_tracked']) == 3
