"""ResNet C4 detector pieces (BASELINE.json configs[3], performance configuration: scda_amd/dropin/models/mask_rcnn/resnet.py,
scda_amd/resnet_config.py).  The reference has no runnable model here (its base class file is missing), so the checks are
against plain PyTorch fp32 modules of the same structure -- the torchvision bottleneck the reference's class body spells out
(models/mask_rcnn/resnet.py:69-106) -- plus one full SCDA iteration as a smoke test."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(a, b, tol=2e-4):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    assert a.shape == b.shape
    err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
    assert err < tol, f"relative-to-max error {err:.3e}"


@pytest.mark.parametrize("shape", [(2, 5, 17, 23), (1, 64, 40, 64), (3, 4, 8, 8)])
def test_maxpool_3x3_stride2(cuda, shape):
    from scda_amd import native
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1))
    assert torch.equal(native.maxpool3x3s2_fwd(x.to(cuda)).cpu(), F.max_pool2d(x, 3, 2, 1))


@pytest.mark.parametrize("shape", [(512, 32, 7, 7), (1, 16, 50, 84), (1, 64, 100, 168), (2, 8, 200, 336),
                                   (1, 8, 3584, 7), (1, 4, 160, 256), (1, 4, 200, 300), (1, 8, 7, 7), (1, 3, 30, 34)])
def test_batch_norm_train_on_roi_shaped_input(cuda, shape):
    """the (image, pixel) index space of a channel is walked as one range: 512 RoIs x 7 x 7 (layer4) and 1 x 50 x 84; batch-1
    planes of <= 16 x 4096 floats go through the register-resident plane kernels (2 / 5 / 7 / 10 / 16 float4 per thread: 4200,
    16800, 25088 = the channel-major RoI head, 40960, 60000 elements), others through the per-channel / sliced forms"""
    from scda_amd import autograd_ops as A
    g = torch.Generator().manual_seed(2)
    x = torch.randn(*shape, generator=g) * 1.3 + 0.2
    C = shape[1]
    ga, be = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xr, gr, br = x.clone().requires_grad_(), ga.clone().requires_grad_(), be.clone().requires_grad_()
    rm, rv = torch.zeros(C), torch.ones(C)
    y = F.relu(F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5))
    dy = torch.randn(y.shape, generator=g); y.backward(dy)
    xg, gg, bg = x.to(cuda).requires_grad_(), ga.to(cuda).requires_grad_(), be.to(cuda).requires_grad_()
    rmg, rvg = torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
    yg = A.BatchNormTrainFn.apply(xg, gg, bg, rmg, rvg, 1e-5, 0.1, 1, 0.0); yg.backward(dy.to(cuda))
    close(yg, y, 1e-5); close(xg.grad, xr.grad, 1e-4); close(gg.grad, gr.grad, 1e-4); close(bg.grad, br.grad, 1e-4)
    close(rmg, rm, 1e-5); close(rvg, rv, 1e-5)


@pytest.mark.parametrize("shape", [(1, 16, 50, 84), (1, 8, 100, 168), (1, 8, 3584, 7), (1, 4, 160, 256), (1, 8, 7, 8)])
def test_batch_norm_residual_join_one_kernel(cuda, shape, monkeypatch):
    """bn3 + "out += residual" + ReLU of a bottleneck (models/mask_rcnn/resnet.py:95-104) in the batch norm's own pass, both ways:
    against plain torch, and bit-identical to the batch norm kernel followed by the join kernel (SCDA_BN_NO_JOIN=1) -- output,
    running statistics, all four gradients"""
    from scda_amd import layers as L
    from scda_amd import native
    g = torch.Generator().manual_seed(7)
    x = torch.randn(*shape, generator=g) * 1.3 + 0.2
    r = torch.randn(*shape, generator=g)
    dy = torch.randn(*shape, generator=g)
    C = shape[1]
    ga, be = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xr, rr, gr, br = x.clone().requires_grad_(), r.clone().requires_grad_(), ga.clone().requires_grad_(), be.clone().requires_grad_()
    rm, rv = torch.zeros(C), torch.ones(C)
    y = F.relu(F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5) + rr); y.backward(dy)
    assert native.batchnorm_add_relu_ok(x.to(cuda))

    def run(joined):
        if joined:
            monkeypatch.delenv("SCDA_BN_NO_JOIN", raising=False)
        else:
            monkeypatch.setenv("SCDA_BN_NO_JOIN", "1")
        bn = L.BatchNorm2d(C).to(cuda).train()
        bn.weight.data.copy_(ga); bn.bias.data.copy_(be)
        xg, rg = x.to(cuda).requires_grad_(), r.to(cuda).requires_grad_()
        yg = bn.forward_add_relu(xg, rg)
        assert (type(yg.grad_fn).__name__ == "BatchNormAddReluFnBackward") == joined
        yg.backward(dy.to(cuda))
        return yg.detach(), xg.grad, rg.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()

    one, two = run(True), run(False)
    for a, b in zip(one, two):
        assert torch.equal(a, b)
    close(one[0], y, 1e-5); close(one[1], xr.grad, 1e-4); close(one[2], rr.grad, 1e-5); close(one[3], gr.grad, 1e-4)
    close(one[4], br.grad, 1e-4); close(one[5], rm, 1e-5); close(one[6], rv, 1e-5)


def test_batch_norm_residual_join_falls_back(cuda):
    """batch > 1, eval mode and planes that are no multiple of 4 floats run the two kernels (same results as plain torch)"""
    from scda_amd import layers as L
    g = torch.Generator().manual_seed(8)
    for shape, train in (((3, 4, 6, 10), True), ((1, 4, 5, 7), True), ((1, 4, 8, 8), False)):
        x, r = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g)
        ref = nn.BatchNorm2d(shape[1]).train(train)
        bn = L.BatchNorm2d(shape[1]).to(cuda).train(train)
        yg = bn.forward_add_relu(x.to(cuda).requires_grad_(), r.to(cuda))
        assert type(yg.grad_fn).__name__ == "AddReluFnBackward"
        close(yg, F.relu(ref(x) + r), 1e-5)


class TorchBottleneck(nn.Module):
    """models/mask_rcnn/resnet.py:69-106 in plain torch"""

    def __init__(self, inplanes, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False); self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4)) if down else None

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return F.relu(out + (x if self.downsample is None else self.downsample(x)))


@pytest.mark.parametrize("inplanes,planes,stride,down,shape", [(64, 32, 2, True, (2, 64, 14, 18)), (128, 32, 1, False, (40, 128, 7, 7)),
                                                               (128, 32, 1, False, (1, 128, 28, 36)), (64, 32, 2, True, (1, 64, 28, 40))])
def test_bottleneck_forward_backward(cuda, inplanes, planes, stride, down, shape):
    from scda_amd import layers as L
    from scda_amd.dropin.models.mask_rcnn.resnet import Bottleneck
    torch.manual_seed(3)
    ref = TorchBottleneck(inplanes, planes, stride, down)
    for m in ref.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    ds = None
    if down:
        ds = nn.Sequential(L.Conv2d(inplanes, planes * 4, kernel_size=1, stride=stride, bias=False), L.BatchNorm2d(planes * 4))
    mine = Bottleneck(inplanes, planes, stride, ds)
    mine.load_state_dict(ref.state_dict())
    mine.to(cuda).train(); ref.train()
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(4))
    xr = x.clone().requires_grad_(); y = ref(xr)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)); y.backward(dy)
    xg = x.to(cuda).requires_grad_(); yg = mine(xg); yg.backward(dy.to(cuda))
    close(yg, y, 1e-4); close(xg.grad, xr.grad, 5e-4)
    rp = dict(ref.named_parameters())
    for k, p in mine.named_parameters():
        close(p.grad, rp[k].grad, 5e-4)
    for k, b in mine.named_buffers():
        if 'running' in k:
            close(b, dict(ref.named_buffers())[k], 1e-5)


def test_resnet50_scda_iteration(cuda):
    """one full SCDA iteration with the ResNet-50 C4 detector at a reduced size (the step, the GAN nets on 32 x 64 feature
    maps / 128 x 256 patches, flat buckets over the TRAINABLE parameters only): finite losses, trainable layers move, the frozen
    stem / layer1 (weights and BN statistics) do not"""
    import bench
    from scda_amd import resnet_config as RC
    from scda_amd.train_step import ScdaTrainer
    Hh, Ww = 384, 640
    torch.manual_seed(0); np.random.seed(0)
    tr = ScdaTrainer(bench.CFG, cuda, lr=1e-4, new_w=Ww, new_h=Hh, models=RC.build_models(bench.CFG), recon_hw=RC.RECON_HW)
    det = tr.model
    assert not det.layer1.training and det.layer2.training and not det.conv1.weight.requires_grad
    before = {k: v.clone() for k, v in det.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    src = torch.randn(1, 3, Hh, Ww, generator=g).clamp_(-1, 1).to(cuda); tgt = torch.randn(1, 3, Hh, Ww, generator=g).clamp_(-1, 1).to(cuda)
    gts = torch.tensor([[[40., 50., 300., 250., 3.], [350., 100., 600., 330., 5.], [100., 200., 220., 370., 1.]]])
    out = tr.step(src, gts, torch.tensor([[Hh, Ww, 1.0]]), tgt)
    torch.cuda.synchronize()
    for k in ('loss', 'rpn_cls', 'rpn_loc', 'rcnn_cls', 'rcnn_loc', 'adloss', 'dis_patch_loss', 'recon_loss'):
        assert np.isfinite(float(out[k])), k
    after = det.state_dict()
    moved = [k for k in before if before[k].dtype.is_floating_point and not torch.equal(before[k], after[k])]
    assert any(k.startswith('layer2.') for k in moved) and any(k.startswith('layer4.') for k in moved) and 'fc_rcnn_cls.weight' in moved
    assert not [k for k in moved if k.startswith(('conv1.', 'bn1.', 'layer1.'))]
    assert int(after['layer2.0.bn1.num_batches_tracked']) == 2 and int(after['layer1.0.bn1.num_batches_tracked']) == 0


@pytest.mark.parametrize("R,C,Co,k", [(32, 32, 48, 3), (48, 16, 16, 3), (16, 64, 32, 1), (5, 8, 8, 3)])
def test_conv_on_stacked_maps_equals_batched_conv(cuda, R, C, Co, k):
    """row_period (include/scda_ops.h: the row_period argument of the conv entry points): a 3x3 convolution of the channel-major view [1, C, R*7, 7]
    with row period 7 == the convolution of the batch [R, C, 7, 7], forward, data gradient and weight gradient (the direct-to-
    LDS kernels when R*49 % 16 == 0 and C % 16 == 0, the register-staged ones otherwise)"""
    from scda_amd import native
    g = torch.Generator().manual_seed(R + C)
    x = torch.randn(R, C, 7, 7, generator=g); w = torch.randn(Co, C, k, k, generator=g) * 0.1
    dy = torch.randn(R, Co, 7, 7, generator=g)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    y = F.conv2d(xr, wr, None, 1, k // 2); y.backward(dy)

    def tall(t):   # [R, C, 7, 7] -> [1, C, R*7, 7]
        return t.permute(1, 0, 2, 3).reshape(1, t.shape[1], R * 7, 7).contiguous().to(cuda)

    def back(t):   # [1, C, R*7, 7] -> [R, C, 7, 7]
        return t.view(t.shape[1], R, 7, 7).permute(1, 0, 2, 3).cpu()

    xt, dyt, wg = tall(x), tall(dy), w.to(cuda)
    close(back(native.conv2d_fwd(xt, wg, None, 1, k // 2, row_period=7)), y, 1e-5)
    close(back(native.conv2d_dgrad(dyt, wg, xt.shape, 1, k // 2, row_period=7)), xr.grad, 1e-5)
    close(native.conv2d_wgrad(dyt, xt, w.shape, 1, k // 2, row_period=7).cpu(), wr.grad, 1e-5)
    if k == 3:   # without the period the taps DO cross maps: the modifier is what makes the difference
        leak = back(native.conv2d_fwd(xt, wg, None, 1, 1))
        assert (leak - y).abs().max() > 1e-3
        with pytest.raises(native.ScdaNativeError):     # a convolution it cannot honour fails loudly
            native.conv2d_fwd(xt, wg, None, 1, 1, row_period=R * 7 - 1)


def test_roi_align_channel_major(cuda):
    from scda_amd import native
    from test_oracle_golden import rand_rois
    rs = np.random.RandomState(5)
    feat = torch.from_numpy(rs.randn(2, 24, 50, 84).astype(np.float32)).to(cuda)
    rois = torch.from_numpy(rand_rois(rs, 48, B=2, W=84 * 16, H=50 * 16)).to(cuda)
    a = native.roi_align_fwd(feat, rois, 8, 8, 1 / 16.)
    b = native.roi_align_fwd(feat, rois, 8, 8, 1 / 16., channel_major=True)
    assert b.shape == (24, 48, 8, 8) and torch.equal(b.permute(1, 0, 2, 3), a)
    top = torch.randn_like(a)
    ga = native.roi_align_bwd(top, rois, feat.shape, 8, 8, 1 / 16.)
    gb = native.roi_align_bwd(top.permute(1, 0, 2, 3).contiguous(), rois, feat.shape, 8, 8, 1 / 16., channel_major=True)
    close(gb, ga, 1e-5)


def rel_l2(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("plane_bn", [False, True])
def test_channel_major_roi_head_equals_reference_layout(cuda, monkeypatch, plane_bn):
    """ResNet.rcnn(): the MI355X layout of the RoI head (pooled maps [C, R, 7, 7], layer4 on the view [1, C, R*7, 7], 3x3
    convolutions with row period 7) against the reference's [R, C, 7, 7] layout on the same weights: features, class / box
    outputs, gradient w.r.t. the backbone feature map, every layer4 / head parameter gradient and the BN running statistics.
    plane_bn=False: both layouts normalise with the per-channel kernel family -> element-wise agreement to 2e-5 / 5e-5 of the
    maximum.  plane_bn=True (the default build): the tall layout's batch norms are the register-resident plane kernels; they
    agree with the others to 2e-7 per call (scripts/debug_bn_plane.py), which is enough to flip the fused ReLU gate of a
    pre-activation that is zero to the last bit -- one such element moves one row of a weight gradient by a few percent -- so
    that variant's gradients are compared in relative L2 (< 1e-2; a layout bug -- a tap crossing RoIs, a permuted channel -- is
    O(1))."""
    if not plane_bn:
        monkeypatch.setenv("SCDA_BN_NO_PLANE", "1")
        # ... and both layouts' 3x3 convolutions on the direct kernel (the stacked 7 x 7 maps take the Winograd kernel by default,
        # whose sums associate differently: 3e-5 per layer, tests/test_conv_wino_gpu.py::test_wino_on_stacked_7x7_maps)
        monkeypatch.setenv("SCDA_WINO_STACKED", "0")
    import bench
    from scda_amd.dropin.models.mask_rcnn.resnet import resnet50
    from test_oracle_golden import rand_rois
    torch.manual_seed(7)
    det = resnet50(cfg=dict(bench.CFG['shared'], roi_align=True, gan_model_flag=2)).to(cuda).train()
    rs = np.random.RandomState(9)
    feat = (torch.randn(1, 1024, 24, 40, generator=torch.Generator().manual_seed(8)) * 0.5).to(cuda)
    rois = torch.from_numpy(rand_rois(rs, 32, B=1, W=40 * 16, H=24 * 16)).to(cuda)
    state = {k: v.clone() for k, v in det.state_dict().items()}
    res = {}
    for name in ("tall", "nchw"):
        det.load_state_dict(state)
        det.zero_grad(set_to_none=True)
        if name == "nchw":
            monkeypatch.setenv("SCDA_RESNET_HEAD_NCHW", "1")
        f = feat.clone().requires_grad_()
        x_fea, cls, loc = det.rcnn(f, rois)
        assert det.roipooling.channel_major == (name == "tall")
        g = torch.Generator().manual_seed(10)
        loss = (x_fea * torch.randn(x_fea.shape, generator=g).to(cuda)).sum() + (cls * torch.randn(cls.shape, generator=g).to(cuda)).sum() \
            + (loc * torch.randn(loc.shape, generator=g).to(cuda)).sum()
        loss.backward()
        res[name] = dict(x_fea=x_fea.detach(), cls=cls.detach(), loc=loc.detach(), dfeat=f.grad,
                         grads={k: p.grad.clone() for k, p in det.named_parameters() if p.grad is not None},
                         stats={k: v.clone() for k, v in det.state_dict().items() if 'layer4' in k and 'running' in k})
    assert any(k.startswith("layer4.0.conv2") for k in res["tall"]["grads"])
    for k in ("x_fea", "cls", "loc"):
        close(res["tall"][k], res["nchw"][k], 2e-4 if plane_bn else 2e-5)      # (default build: Winograd against the direct kernel)
    for k, v in res["nchw"]["stats"].items():
        close(res["tall"]["stats"][k], v, 1e-5)
    if plane_bn:
        # a flipped gate changes one element of one gradient map; by the time it has travelled back through the blocks below
        # it is a dense perturbation of ~1e-3 relative L2 (measured: 2.4e-3 on layer4.0.conv1.weight)
        assert rel_l2(res["tall"]["dfeat"], res["nchw"]["dfeat"]) < 1e-2
        for k, gr in res["nchw"]["grads"].items():
            assert rel_l2(res["tall"]["grads"][k], gr) < 1e-2, k
    else:
        close(res["tall"]["dfeat"], res["nchw"]["dfeat"], 2e-5)
        for k, gr in res["nchw"]["grads"].items():
            close(res["tall"]["grads"][k], gr, 5e-5)


def test_frozen_conv_bn_folding(cuda, monkeypatch):
    """stem + layer1 (frozen, eval-mode BN): conv -> BN -> ReLU as one convolution with folded weights and a bias / ReLU epilogue
    against the separate kernels (SCDA_RESNET_NO_FOLD=1), on non-trivial BN parameters and running statistics; the cache follows
    an in-place weight update"""
    import bench
    from scda_amd.dropin.models.mask_rcnn.resnet import resnet50
    torch.manual_seed(11)
    det = resnet50(cfg=dict(bench.CFG['shared'], roi_align=True, gan_model_flag=2)).to(cuda).train()
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for m in list(det.layer1.modules()) + [det.bn1]:
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5); m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.2); m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    x = torch.randn(1, 3, 96, 160, generator=g).to(cuda)

    def stem_layer1():
        if not_folded:
            monkeypatch.setenv("SCDA_RESNET_NO_FOLD", "1")
        else:
            monkeypatch.delenv("SCDA_RESNET_NO_FOLD", raising=False)
        with torch.no_grad():
            if not_folded:
                t = det.maxpool(det.bn1(det.conv1(x)))
            else:
                from scda_amd.dropin.models.mask_rcnn.resnet import folded_conv_bn
                from scda_amd.autograd_ops import ACT_RELU
                t = det.maxpool(folded_conv_bn(x, det.conv1, det.bn1, ACT_RELU))
            return det.layer1(t)
    not_folded = True; ref = stem_layer1()
    not_folded = False; got = stem_layer1()
    assert hasattr(det.layer1[0].conv1, "_scda_folded") and not det.layer1[0].bn1.training
    close(got, ref, 2e-5)
    with torch.no_grad():
        det.layer1[1].conv2.weight.mul_(1.5)         # e.g. load_state_dict: in place, version bump -> re-folded
    not_folded = True; ref2 = stem_layer1()
    not_folded = False; got2 = stem_layer1()
    close(got2, ref2, 2e-5)
    assert (ref2 - ref).abs().max() > 1e-3
