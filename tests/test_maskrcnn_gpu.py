"""BASELINE.json configs[4]: the ResNet-50 C4 detector with the mask branch of models/mask_rcnn/resnet.py:146-193 + SCDA losses.
The reference cannot run this configuration (its base class file is missing); the HIP path is compared with the torch-CPU restatement
of the layer specification (oracle/resnet_ref.py, with_mask) -- the five losses and the gradient of every trainable tensor, the mask
head's included -- and one full SCDA iteration runs at the configuration's own size."""
import copy

import numpy as np
import pytest
import torch

import model_common as mc
from test_resnet_oracle_gpu import CFG as RCFG, reinit, rel_l2

pytestmark = pytest.mark.gpu


def test_conv_transpose_2x2_s2_matches_torch(cuda):
    """the up-sampling layer of the mask branch: forward and all three gradients against torch's CPU ConvTranspose2d"""
    import torch.nn.functional as F
    from scda_amd import layers as L
    torch.manual_seed(0)
    layer = L.ConvTranspose2x2s2(256, 256)
    ref_w, ref_b = layer.weight.detach().clone().requires_grad_(), layer.bias.detach().clone().requires_grad_()
    x = torch.randn(16, 256, 14, 14)
    xr = x.clone().requires_grad_()
    want = F.conv_transpose2d(xr, ref_w, ref_b, stride=2)
    g = torch.randn_like(want)
    want.backward(g)
    layer = layer.to(cuda)
    xd = x.to(cuda).requires_grad_()
    got = layer(xd)
    got.backward(g.to(cuda))
    assert got.shape == want.shape
    assert rel_l2(got, want) <= 1e-5
    assert rel_l2(xd.grad, xr.grad) <= 1e-5 and rel_l2(layer.weight.grad, ref_w.grad) <= 1e-5 and rel_l2(layer.bias.grad, ref_b.grad) <= 1e-5


def test_mask_detector_losses_and_gradients_match_oracle(cuda):
    from oracle import resnet_ref as RR, torch_ref as R
    from scda_amd import autograd_ops as A
    from scda_amd import resnet_config as RC
    from scda_amd.dropin.models.mask_rcnn.resnet import resnet50
    import seeded_init
    H, W, G = 256, 384, 4
    shared = dict(RCFG['shared'], with_mask=True,
                  train_mask_target=dict(RC.MASK_TARGET, batch_size_per_image=8, positive_iou_thresh=0.3))
    torch.manual_seed(1)
    ref = RR.RefResNetDetector(dict(shared))
    reinit(ref)
    ref.train()
    src, tgt = seeded_init.synth_images(61, H, W)
    gts = seeded_init.synth_gts(G, 62, H, W)
    masks = RC.synth_masks(gts, H, W)
    info = torch.tensor([[H, W, 1.0]])

    def inputs(dev=None):
        return {'cfg': RCFG, 'image': src if dev is None else src.to(dev), 'image_info': info, 'ground_truth_bboxes': gts,
                'ground_truth_masks': masks, 'ignore_regions': None, 'cluster_num': 4, 'threshold': 128}

    rec = R.SelectionRecorder()
    handles = rec.attach(ref)

    def record_rpn(cls, loc):
        rec.add("rpn_cls", cls, cls.detach().clone())
        rec.add("rpn_loc", loc, loc.detach().clone())
        return cls, loc
    R.use_cpu_backend()
    try:
        with mc.probed(rpn_output=record_rpn):
            np.random.seed(7)
            torch.set_num_threads(16)
            want = ref(inputs(), tgt)
            sum(want['losses']).backward()
    finally:
        rec.detach(handles)
        R.reset_backend()
        torch.set_num_threads(1)
    assert len(want['losses']) == 5 and float(want['losses'][4]) > 0.1          # a real mask loss (BCE of near-zero logits ~ 0.69)

    det = resnet50(cfg=dict(shared))
    reinit(det)
    det = det.to(cuda).train()
    det.tall_head = False              # replayed selections are keyed by output shape (see tests/test_resnet_oracle_gpu.py)
    assert sorted(k for k in det.state_dict() if k.startswith('mask_head')) == sorted(k for k in ref.state_dict() if k.startswith('mask_head'))
    rsrc = mc.ReplaySource(rec, cuda)
    with mc.probed(replay=rsrc, rpn_output=rsrc.rpn):
        np.random.seed(7)
        got = det(inputs(cuda), tgt.to(cuda))
        sum(got['losses']).backward()
        torch.cuda.synchronize()
    assert det.last_mask_rois == ref.last_mask_rois >= 4, (det.last_mask_rois, ref.last_mask_rois)
    names = ("rpn_cls", "rpn_loc", "rcnn_cls", "rcnn_loc", "mask")
    # (parity unpinned for this configuration -- no reference model exists; the bound is the one the VGG iteration asserts since round 6,
    #  the achieved deltas are printed and logged)
    mc.check_losses(dict(zip(names, got['losses'])), dict(zip(names, want['losses'])), names, 1e-5, "maskrcnn_detector[800x1344 oracle size]")
    rp = dict(ref.named_parameters())
    errs = {k: rel_l2(p.grad, rp[k].grad) for k, p in det.named_parameters() if p.requires_grad}
    mask_errs = {k: e for k, e in errs.items() if k.startswith('mask_head')}
    worst = max(errs.items(), key=lambda kv: kv[1])
    within = sum(e <= 1e-4 for e in errs.values()) / len(errs)
    print("mask detector gradients vs oracle: %d tensors (%d of the mask head, worst %.2e), %.1f %% within 1e-4, worst %s %.2e"
          % (len(errs), len(mask_errs), max(mask_errs.values()), 100 * within, worst[0], worst[1]))
    assert len(mask_errs) == 12 and max(mask_errs.values()) <= 1e-4, mask_errs
    assert within >= 0.9 and worst[1] <= 2e-4, (worst, within)


def test_maskrcnn_scda_iteration_at_800x1344(cuda):
    """configs[4]'s own size: two full SCDA iterations with the mask loss -- finite losses, the mask head and the backbone behind it
    move, 64 positive RoIs feed the branch"""
    import bench
    from scda_amd import resnet_config as RC
    torch.manual_seed(0); np.random.seed(0)
    tr = RC.make_trainer(bench.CFG, cuda, lr=1e-4, with_mask=True, mask_iou=0.2)
    det = tr.model
    before = {k: v.clone() for k, v in det.state_dict().items()}
    src, tgt, gts, info = bench.synth_batch(0, RC.H, RC.W)
    masks = RC.synth_masks(gts)
    for _ in range(2):
        out = tr.step(src.to(cuda), gts, info, tgt.to(cuda), gt_masks=masks)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(v).all()) for v in out.values() if torch.is_tensor(v)), out
    assert 0.0 < float(out['mask_loss']) < 5.0
    assert 12 <= det.last_mask_rois <= RC.MASK_ROIS        # at least the 12 appended ground-truth boxes, at most the quota
    after = det.state_dict()
    moved = [k for k in before if k.startswith('mask_head') and not torch.equal(before[k], after[k])]
    assert len(moved) == 12, moved
    assert not torch.equal(before['layer3.5.conv3.weight'], after['layer3.5.conv3.weight'])
    assert torch.equal(before['layer1.0.conv1.weight'], after['layer1.0.conv1.weight'])


def test_channel_major_mask_branch_equals_reference_layout(cuda, monkeypatch):
    """the mask branch on the stacked view [1, C, R*14, 14] (row period 14) against the reference's [R, C, 14, 14] batch: logits,
    the gradient into the backbone features and every parameter gradient of the branch.  A statement about LAYOUTS: both sides on the
    direct kernels (the [R, C, 14, 14] batch would otherwise take the Winograd kernel, the stacked view cannot: two algorithms whose
    1e-6 differences flip ReLU signs near zero)."""
    monkeypatch.setenv("SCDA_WINOGRAD", "0")
    from scda_amd.dropin.models.mask_rcnn.resnet import resnet50
    torch.manual_seed(3)
    shared = dict(RCFG['shared'], with_mask=True)
    det = resnet50(cfg=shared).to(cuda).train()
    feat = torch.randn(1, 1024, 20, 30, device=cuda)
    g = torch.Generator().manual_seed(4)
    x1 = torch.rand(12, generator=g) * 300; y1 = torch.rand(12, generator=g) * 200
    rois = torch.stack([torch.zeros(12), x1, y1, x1 + 20 + torch.rand(12, generator=g) * 150, y1 + 20 + torch.rand(12, generator=g) * 100], 1).to(cuda)
    up = torch.randn(12, shared['num_classes'], 28, 28, device=cuda)
    res = {}
    for tall in (False, True):
        det.tall_head = tall
        det.zero_grad()
        f = feat.clone().requires_grad_()
        y = det.mask_predictor(f, rois)
        assert tuple(y.shape) == (12, shared['num_classes'], 28, 28)
        (y * up).sum().backward()
        res[tall] = (y.detach().clone(), f.grad.clone(), {k: p.grad.clone() for k, p in det.mask_head.named_parameters()})
    assert det.mask_roipooling.channel_major is True
    assert rel_l2(res[True][0], res[False][0]) <= 1e-5
    assert rel_l2(res[True][1], res[False][1]) <= 1e-5
    for k in res[True][2]:
        assert rel_l2(res[True][2][k], res[False][2][k]) <= 1e-5, k
