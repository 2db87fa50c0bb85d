"""ResNet-50 C4 detector (BASELINE.json configs[3]) against its CPU oracle (oracle/resnet_ref.py: plain torch-CPU convolutions /
batch norms / autograd + the C RoIAlign), whole detector: the four losses and the gradient of every trainable tensor.

The oracle records which element it selected at every ReLU (tests/model_common.ReplaySource, as the VGG iteration test does) and
its RPN outputs are handed to the proposal ranking after checking that the device agrees to 1e-5 -- what is compared is then the
kernels' arithmetic on identical RoIs: losses 1e-4, gradients 1e-4 relative L2 per tensor (>= 90 % of the tensors; 2e-4 for the
cancelling batch-norm sums at the far end of the chain), batch-norm running statistics 1e-5.
Plus one full SCDA iteration at the configuration's own size, 800 x 1344."""
import copy

import numpy as np
import pytest
import torch

import model_common as mc      # puts tests/golden on the path (seeded_init)

pytestmark = pytest.mark.gpu

CFG = copy.deepcopy(mc.CFG)
for _k in CFG:
    CFG[_k].update(gan_model_flag=2, roi_align=True)


def reinit(det):
    """seeded weights (same keys -> same values on both sides); the He-style draw makes the RPN head's outputs O(30) -- saturated
    objectness, exp() of huge size deltas -- so the two heads are scaled to the magnitude the model's own init gives them"""
    import seeded_init
    seeded_init.seeded_reinit(det, 51, 'det')
    with torch.no_grad():
        for k, v in det.state_dict().items():
            if k.startswith('rpn_head.') and k.endswith('weight'):
                v.mul_(0.05)


def rel_l2(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("H,W,G", [(256, 384, 4)])
def test_resnet_detector_losses_and_gradients_match_oracle(cuda, H, W, G):
    from oracle import resnet_ref as RR, torch_ref as R
    from scda_amd import autograd_ops as A
    from scda_amd.dropin.models.mask_rcnn.resnet import resnet50
    import seeded_init
    torch.manual_seed(1)
    ref = RR.RefResNetDetector(dict(CFG['shared']))
    reinit(ref)
    ref.train()
    src, tgt = seeded_init.synth_images(61, H, W)
    gts = seeded_init.synth_gts(G, 62, H, W)
    info = torch.tensor([[H, W, 1.0]])

    def inputs(dev=None):
        return {'cfg': CFG, 'image': src if dev is None else src.to(dev), 'image_info': info, 'ground_truth_bboxes': gts,
                'ignore_regions': None, 'cluster_num': 4, 'threshold': 128}

    # ---- oracle, CPU
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
    before = {k: v.clone() for k, v in ref.state_dict().items()}

    # ---- product, device: same weights (the oracle's state BEFORE its forward updated the BN statistics is gone: re-draw)
    det = resnet50(cfg=dict(CFG['shared']))
    reinit(det)
    det = det.to(cuda).train()
    # the RoI head in the reference's [R, C, 7, 7] layout: the replayed selections are keyed by output shape.  The channel-major head
    # the product runs by default is compared with THIS layout tensor by tensor in tests/test_resnet_gpu.py
    # (test_channel_major_roi_head_equals_reference_layout).
    det.tall_head = False
    rsrc = mc.ReplaySource(rec, cuda)
    with mc.probed(replay=rsrc, rpn_output=rsrc.rpn):
        np.random.seed(7)
        got = det(inputs(cuda), tgt.to(cuda))
        sum(got['losses']).backward()
        torch.cuda.synchronize()
        used = rsrc.used
    assert used >= 30, used
    names = ("rpn_cls", "rpn_loc", "rcnn_cls", "rcnn_loc")
    # (parity unpinned for this configuration -- no reference model exists; the bound is the one the VGG iteration asserts since round 6,
    #  the achieved deltas are printed and logged)
    mc.check_losses(dict(zip(names, got['losses'])), dict(zip(names, want['losses'])), names, 1e-5, "resnet50_detector")
    rp = dict(ref.named_parameters())
    errs = {}
    for k, p in det.named_parameters():
        if not p.requires_grad:
            assert rp[k].grad is None and p.grad is None, k          # stem + layer1 frozen on both sides
            continue
        errs[k] = rel_l2(p.grad, rp[k].grad)
    worst = max(errs.items(), key=lambda kv: kv[1])
    within = sum(e <= 1e-4 for e in errs.values()) / len(errs)
    print("ResNet-50 C4 detector gradients vs oracle: %d tensors, %.1f %% within 1e-4 relative L2, worst %s %.2e"
          % (len(errs), 100 * within, worst[0], worst[1]))
    # 1e-4 for the bulk; the batch-norm bias / weight gradients at the far end of the backward chain (layer2.0: ~35 train-mode batch
    # norms behind the loss) are signed sums over a whole map whose terms largely cancel -- their relative error is the summation
    # order's (measured: 93 % of the 136 tensors within 1e-4, worst 1.4e-4)
    assert len(errs) > 100 and within >= 0.9 and worst[1] <= 2e-4, (worst, within)
    sd = det.state_dict()
    for k, v in before.items():
        if k.endswith(("running_mean", "running_var")):
            d = float((sd[k].cpu() - v).abs().max() / (v.abs().max() + 1e-12))
            assert d <= 1e-5, (k, d)
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v), k
    assert int(sd['layer1.0.bn1.num_batches_tracked']) == 0 and int(sd['layer2.0.bn1.num_batches_tracked']) == 2


def test_resnet50_scda_iteration_at_800x1344(cuda):
    """BASELINE.json configs[3]'s own size: one full SCDA iteration (detector + decoders + discriminators, four optimiser steps) on
    800 x 1344 -- finite losses, the RoI quota is reached on both images (512 sampled RoIs need >= 512 proposals from the target
    image, else the reference's fallback reuses the source clusters), trainable layers move, the frozen stem / layer1 do not."""
    import bench
    from scda_amd import resnet_config as RC
    torch.manual_seed(0); np.random.seed(0)
    tr = RC.make_trainer(bench.CFG, cuda, lr=1e-4)
    det = tr.model
    before = {k: v.clone() for k, v in det.state_dict().items()}
    src, tgt, gts, info = bench.synth_batch(0, RC.H, RC.W)
    out = tr.step(src.to(cuda), gts, info, tgt.to(cuda))
    torch.cuda.synchronize()
    for k in ('loss', 'rpn_cls', 'rpn_loc', 'rcnn_cls', 'rcnn_loc', 'adloss', 'dis_patch_loss', 'recon_loss', 'fake_loss_source',
              'fake_loss_target'):
        assert np.isfinite(float(out[k])), k
    assert min(tr.last_num_proposals) >= 512, tr.last_num_proposals
    after = det.state_dict()
    moved = [k for k in before if before[k].dtype.is_floating_point and not torch.equal(before[k], after[k])]
    assert all(any(k.startswith(p) for k in moved) for p in ('layer2.', 'layer3.', 'layer4.', 'rpn_head.', 'fc_rcnn_cls.', 'fc_rcnn_loc.'))
    assert not [k for k in moved if k.startswith(('conv1.', 'bn1.', 'layer1.'))]
    assert int(after['layer3.5.bn3.num_batches_tracked']) == 2 and int(after['layer4.2.bn3.num_batches_tracked']) == 2
    assert int(after['layer1.0.bn1.num_batches_tracked']) == 0
