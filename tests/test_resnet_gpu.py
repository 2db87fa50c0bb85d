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


@pytest.mark.parametrize("shape", [(512, 32, 7, 7), (1, 16, 50, 84), (1, 64, 100, 168), (2, 8, 200, 336)])
def test_batch_norm_train_on_roi_shaped_input(cuda, shape):
    """the (image, pixel) index space of a channel is walked as one range: 512 RoIs x 7 x 7 (layer4) and 1 x 50 x 84"""
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


@pytest.mark.parametrize("inplanes,planes,stride,down,shape", [(64, 32, 2, True, (2, 64, 14, 18)), (128, 32, 1, False, (40, 128, 7, 7))])
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
