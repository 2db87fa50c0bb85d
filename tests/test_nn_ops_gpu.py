"""Layer kernels (scda_amd/csrc/nn_ops.hip) vs the plain PyTorch fp32 CPU reference of the same op.
fp32 tolerance 1e-5 relative-to-max unless stated (reductions differ in summation order only)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(a, b, tol=1e-5):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
    assert err < tol, f"relative-to-max error {err:.3e}"


def gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("hw", [(8, 12), (7, 9), (6, 11), (9, 10)])      # odd extents: floor mode drops the last row / column
def test_maxpool(cuda, hw):
    from scda_amd import autograd_ops as A
    x = torch.randn(2, 5, *hw, generator=gen(1))
    x[0, 0, 0, :4] = 1.0  # ties -> first wins
    xr = x.clone().requires_grad_()
    y = F.max_pool2d(xr, 2, 2)
    dy = torch.randn(y.shape, generator=gen(2))
    y.backward(dy)
    xg = x.to(cuda).requires_grad_()
    yg = A.MaxPool2x2Fn.apply(xg)
    yg.backward(dy.to(cuda))
    assert torch.equal(yg.cpu(), y.detach())
    assert torch.equal(xg.grad.cpu(), xr.grad)


@pytest.mark.parametrize("mode,fn", [("relu", F.relu), ("leaky", lambda t: F.leaky_relu(t, 0.01)), ("tanh", torch.tanh),
                                     ("sigmoid", torch.sigmoid)])
def test_activations(cuda, mode, fn):
    from scda_amd import autograd_ops as A, native as N
    x = torch.randn(3, 7, 5, generator=gen(3)) * 2
    xr = x.clone().requires_grad_()
    y = fn(xr); dy = torch.randn(y.shape, generator=gen(4)); y.backward(dy)
    xg = x.to(cuda).requires_grad_()
    yg = A.ActFn.apply(xg, N.ACT_MODE[mode], 0.01); yg.backward(dy.to(cuda))
    close(yg, y, 1e-6); close(xg.grad, xr.grad, 1e-5)


def test_dropout_and_mask_statistics(cuda):
    from scda_amd import autograd_ops as A, native as N
    m = N.dropout_mask((1024, 1024), 0.5, 1234, cuda)
    frac = m.float().mean().item()
    assert abs(frac - 0.5) < 0.005
    m2 = N.dropout_mask((1024, 1024), 0.5, 1235, cuda)
    assert (m != m2).float().mean().item() > 0.4          # different seed -> different mask
    assert torch.equal(m, N.dropout_mask((1024, 1024), 0.5, 1234, cuda))  # same seed -> same mask
    x = torch.randn(64, 64, device=cuda, requires_grad=True)
    mk = N.dropout_mask((64, 64), 0.5, 7, cuda)
    y = A.DropoutFn.apply(x, mk, 2.0)
    y.backward(torch.ones_like(y))
    assert torch.equal(y.detach(), x.detach() * mk.float() * 2.0) and torch.equal(x.grad, mk.float() * 2.0)


@pytest.mark.parametrize("R,C,ignore", [(30720, 2, -1), (512, 9, -100), (77, 5, -1)])
def test_cross_entropy(cuda, R, C, ignore):
    from scda_amd import autograd_ops as A
    x = torch.randn(R, C, generator=gen(5)) * 3
    t = torch.randint(0, C, (R,), generator=gen(6))
    if ignore == -1:
        t[torch.rand(R, generator=gen(7)) < 0.6] = -1
    xr = x.clone().requires_grad_()
    l = F.cross_entropy(xr, t, ignore_index=ignore); (l * 0.37).backward()
    xg = x.to(cuda).requires_grad_()
    lg = A.cross_entropy(xg, t.to(cuda), ignore); (lg * 0.37).backward()
    close(lg, l, 1e-5); close(xg.grad, xr.grad, 1e-5)


def test_accuracy_and_row_softmax(cuda):
    from scda_amd import native as N
    x = torch.randn(5000, 2, generator=gen(8)); t = torch.randint(-1, 2, (5000,), generator=gen(9))
    keep = t != -1
    ref = (x[keep].argmax(1) == t[keep]).float().mean() * 100
    close(N.accuracy(x.to(cuda), t.to(cuda), -1)[0], ref, 1e-6)
    close(N.row_softmax(x.to(cuda)), F.softmax(x, 1), 1e-6)


@pytest.mark.parametrize("with_mask", [True, False])
def test_smooth_l1(cuda, with_mask):
    from scda_amd import autograd_ops as A
    p = torch.randn(1, 60, 32, 64, generator=gen(10)); t = torch.randn(1, 60, 32, 64, generator=gen(11)) * 0.5
    m = (torch.rand(1, 60, 32, 64, generator=gen(12)) < 0.1).float() if with_mask else None
    pr = p.clone().requires_grad_()
    d = (pr * m if with_mask else pr) - t
    a = d.abs(); near = (a < 1 / 9.).float()
    l = (d.pow(2) * 9 / 2. * near + (a - 0.5 / 9.) * (1 - near)).sum() / 256.
    l.backward()
    pg = p.to(cuda).requires_grad_()
    lg = A.smooth_l1_sum(pg, m.to(cuda) if with_mask else None, t.to(cuda), 3.0, 1 / 256.)
    lg.backward()
    close(lg, l, 1e-5); close(pg.grad, pr.grad, 1e-5)


# plane sizes: generic path (16x16, 24x40) and the register-cached float4 paths (64x64, 128x128, 256x256 planes)
@pytest.mark.parametrize("shape", [(4, 6, 16, 16), (2, 3, 24, 40), (2, 5, 64, 64), (2, 3, 128, 128), (1, 2, 256, 256)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_instance_norm(cuda, act, shape):
    from scda_amd import autograd_ops as A
    x = torch.randn(*shape, generator=gen(13)) * 2 + 0.5
    xr = x.clone().requires_grad_()
    y = F.instance_norm(xr, eps=1e-5)
    y = [lambda v: v, F.relu, lambda v: F.leaky_relu(v, 0.01)][act](y)
    dy = torch.randn(y.shape, generator=gen(14)); y.backward(dy)
    xg = x.to(cuda).requires_grad_()
    yg = A.InstanceNormFn.apply(xg, 1e-5, act, 0.01); yg.backward(dy.to(cuda))
    close(yg, y, 1e-5); close(xg.grad, xr.grad, 1e-4)


@pytest.mark.parametrize("act", [0, 2])
def test_batch_norm_train(cuda, act):
    from scda_amd import autograd_ops as A
    B, C, H, W = 4, 10, 8, 8
    x = torch.randn(B, C, H, W, generator=gen(15)) * 1.5 + 0.3
    ga = 1 + 0.1 * torch.randn(C, generator=gen(16)); be = 0.1 * torch.randn(C, generator=gen(17))
    rm = 0.1 * torch.randn(C, generator=gen(18)); rv = 1 + 0.1 * torch.rand(C, generator=gen(19))
    xr, gr, br = x.clone().requires_grad_(), ga.clone().requires_grad_(), be.clone().requires_grad_()
    rm_r, rv_r = rm.clone(), rv.clone()
    y = F.batch_norm(xr, rm_r, rv_r, gr, br, True, 0.1, 1e-5)
    if act == 2:
        y = F.leaky_relu(y, 0.01)
    dy = torch.randn(y.shape, generator=gen(20)); y.backward(dy)
    xg, gg, bg = x.to(cuda).requires_grad_(), ga.to(cuda).requires_grad_(), be.to(cuda).requires_grad_()
    rm_g, rv_g = rm.to(cuda), rv.to(cuda)
    yg = A.BatchNormTrainFn.apply(xg, gg, bg, rm_g, rv_g, 1e-5, 0.1, act, 0.01); yg.backward(dy.to(cuda))
    close(yg, y, 1e-5); close(xg.grad, xr.grad, 1e-4); close(gg.grad, gr.grad, 1e-4); close(bg.grad, br.grad, 1e-4)
    close(rm_g, rm_r, 1e-5); close(rv_g, rv_r, 1e-5)


@pytest.mark.parametrize("act", [0, 1, 2])
def test_batch_norm_eval_mode(cuda, act):
    """scda_amd.layers.BatchNorm2d in eval mode (vgg16_bn validation, dis_patch.eval()): running statistics, no update; the
    gradient w.r.t. the input treats statistics and affine parameters as constants"""
    from scda_amd import layers as L
    B, C, H, W = 3, 12, 9, 11
    x = torch.randn(B, C, H, W, generator=gen(41)) * 1.5 + 0.3
    ref = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        ref.weight.copy_(1 + 0.1 * torch.randn(C, generator=gen(42))); ref.bias.copy_(0.1 * torch.randn(C, generator=gen(43)))
        ref.running_mean.copy_(0.2 * torch.randn(C, generator=gen(44))); ref.running_var.copy_(0.5 + torch.rand(C, generator=gen(45)))
    mod = L.BatchNorm2d(C, fused_act=act, slope=0.01)
    mod.load_state_dict(ref.state_dict())
    ref.eval(); mod.to(cuda).eval()
    xr = x.clone().requires_grad_()
    y = [lambda t: t, F.relu, lambda t: F.leaky_relu(t, 0.01)][act](ref(xr))
    dy = torch.randn(y.shape, generator=gen(46)); y.backward(dy)
    xg = x.to(cuda).requires_grad_()
    yg = mod(xg); yg.backward(dy.to(cuda))
    close(yg, y, 1e-6); close(xg.grad, xr.grad, 1e-6)
    sd = mod.state_dict()
    assert torch.equal(sd['running_mean'].cpu(), ref.running_mean) and int(sd['num_batches_tracked']) == 0


@pytest.mark.parametrize("lr_schedule", [False, True])
def test_flat_adam_matches_torch_adam_under_schedulers(cuda, lr_schedule):
    """FlatAdam (one fused kernel on the bucket) == torch.optim.Adam on the individual parameters, step for step, also while
    torch's MultiStepLR and the warm-up scheduler move the learning rate (the lr reaches the kernel through param_groups)"""
    from torch.optim.lr_scheduler import MultiStepLR
    from scda_amd.flat import FlatAdam, FlatParams
    from scda_amd.lr_schedule import IterExponentialLR
    torch.manual_seed(3)
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.Linear(7, 3))
    mine = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.Linear(7, 3))
    mine.load_state_dict(ref.state_dict())
    mine.to(cuda)
    flat = FlatParams(mine)
    o_ref = torch.optim.Adam(ref.parameters(), 1e-2, betas=(0.9, 0.999), weight_decay=1e-4)
    o_mine = FlatAdam(flat, 1e-2, betas=(0.9, 0.999), weight_decay=1e-4)
    scheds = []
    if lr_schedule:
        scheds = [IterExponentialLR(o_ref, 2.0), IterExponentialLR(o_mine, 2.0)]
    for it in range(6):
        if it == 3 and lr_schedule:
            for o in (o_ref, o_mine):
                for g in o.param_groups:
                    g['initial_lr'] = g['lr']
            scheds = [MultiStepLR(o_ref, [1, 2], 0.1), MultiStepLR(o_mine, [1, 2], 0.1)]
        for sch in scheds:
            sch.step()
        assert o_ref.param_groups[0]['lr'] == pytest.approx(o_mine.param_groups[0]['lr'], rel=1e-12)
        g = torch.Generator().manual_seed(100 + it)
        o_mine.zero_grad()
        for pr, pm in zip(ref.parameters(), mine.parameters()):
            gr = torch.randn(pr.shape, generator=g) * 0.1
            pr.grad = gr.clone()
            pm.grad.copy_(gr)
        o_ref.step(); o_mine.step()
    if lr_schedule:
        assert o_mine.param_groups[0]['lr'] == pytest.approx(1e-2 * 4 * 0.01)
    for pr, pm in zip(ref.parameters(), mine.parameters()):
        close(pm, pr, 2e-6)
    sd = o_mine.state_dict()
    assert int(sd['state'][0]['step']) == 6 and sd['state'][0]['exp_avg'].numel() == flat.numel


@pytest.mark.parametrize("shape", [(4, 3, 64, 64), (1, 2, 5, 7), (4, 8, 128, 128)])
def test_upsample2x(cuda, shape):
    from scda_amd import autograd_ops as A
    x = torch.randn(*shape, generator=gen(21))
    xr = x.clone().requires_grad_()
    y = F.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=True)
    dy = torch.randn(y.shape, generator=gen(22)); y.backward(dy)
    xg = x.to(cuda).requires_grad_()
    yg = A.Upsample2xFn.apply(xg); yg.backward(dy.to(cuda))
    close(yg, y, 1e-5); close(xg.grad, xr.grad, 1e-5)


@pytest.mark.parametrize("shape", [(4, 128, 128, 128), (4, 64, 256, 256), (2, 3, 16, 24), (1, 2, 256, 8)])
def test_upsample2x_backward_rows_form_is_bit_identical(cuda, shape, monkeypatch):
    """the gradient of the bilinear x2 with eight input rows in flight per workgroup (upsample2_bwd_rows_kernel: the decoders'
    [4, 128, 128, 128] and [4, 64, 256, 256] output gradients) against the row-at-a-time kernel it replaces
    (SCDA_UPSAMPLE_BWD_ROWWISE=1): the same weights in the same order, bit for bit; both against torch"""
    from scda_amd import native as N
    dy = torch.randn(*shape, generator=gen(31))
    x = torch.zeros(shape[0], shape[1], shape[2] // 2, shape[3] // 2, requires_grad=True)
    F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True).backward(dy)
    d = dy.to(cuda)
    new = N.upsample2x_bwd(d)
    monkeypatch.setenv("SCDA_UPSAMPLE_BWD_ROWWISE", "1")
    old = N.upsample2x_bwd(d)
    assert torch.equal(new, old)
    close(new, x.grad, 1e-5)


def test_bce_gap_rowmean_add(cuda):
    from scda_amd import autograd_ops as A, native as N
    p = torch.rand(1, 1024, generator=gen(23)).clamp(1e-6, 1 - 1e-6); p[0, 0] = 0.0; p[0, 1] = 1.0
    t = torch.rand(1, 1024, generator=gen(24))
    pr = p.clone().requires_grad_()
    l = F.binary_cross_entropy(pr, t); (l * 1.7).backward()
    pg = p.to(cuda).requires_grad_()
    lg = A.binary_cross_entropy(pg, t.to(cuda)); (lg * 1.7).backward()
    close(lg, l, 1e-5); close(pg.grad[:, 2:], pr.grad[:, 2:], 1e-5)
    x = torch.randn(4, 512, 8, 8, generator=gen(25)); xr = x.clone().requires_grad_()
    y = F.adaptive_avg_pool2d(xr, 1).flatten(1); dy = torch.randn(y.shape, generator=gen(26)); y.backward(dy)
    xg = x.to(cuda).requires_grad_(); yg = A.GlobalAvgPoolFn.apply(xg); yg.backward(dy.to(cuda))
    close(yg, y, 1e-5); close(xg.grad, xr.grad, 1e-5)
    close(N.row_mean(y.detach().to(cuda).contiguous()), y.detach().mean(1), 1e-5)
    a = torch.randn(3, 4, 5, generator=gen(27)); b = torch.randn(3, 4, 5, generator=gen(28))
    close(A.AddFn.apply(a.to(cuda), b.to(cuda)), a + b, 1e-7)


@pytest.mark.parametrize("n", [1, 5, 1027, 4096 * 257])
def test_adam_matches_torch(cuda, n):
    from scda_amd import native as N
    p = torch.randn(n, generator=gen(29)); gs = [torch.randn(n, generator=gen(30 + i)) * 0.1 for i in range(3)]
    pr = p.clone().requires_grad_()
    opt = torch.optim.Adam([pr], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)
    pg = torch.zeros((n + 3) // 4 * 4, device=cuda)[:n]; pg.copy_(p)
    m = torch.zeros_like(pg); v = torch.zeros_like(pg)
    for i, g in enumerate(gs):
        pr.grad = g.clone(); opt.step()
        N.adam_step(pg, g.to(cuda), m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-4, i + 1)
    close(pg, pr, 1e-6)


def test_conv_transpose_1x1_and_layers(cuda):
    from scda_amd import layers as L
    ref = torch.nn.ConvTranspose2d(32, 3, 1, 1, 0)
    mine = L.ConvTranspose1x1(32, 3).to(cuda)
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(4, 32, 16, 16, generator=gen(40)); xr = x.clone().requires_grad_()
    y = ref(xr); dy = torch.randn(y.shape, generator=gen(41)); y.backward(dy)
    xg = x.to(cuda).requires_grad_(); yg = mine(xg); yg.backward(dy.to(cuda))
    close(yg, y, 1e-4); close(xg.grad, xr.grad, 1e-4); close(mine.weight.grad, ref.weight.grad, 1e-4); close(mine.bias.grad, ref.bias.grad, 1e-4)


def test_decoder_32_forward_backward(cuda):
    """GAN_decoder_AE_32 end to end on the device: [4,128,64*64] cluster features -> [4,3,128,128] images in (-1,1)"""
    from scda_amd.dropin.models.faster_rcnn.faster_rcnn_adver_expansion_reweight_cluster import GAN_decoder_AE_32
    torch.manual_seed(0)
    m = GAN_decoder_AE_32({'ch': 128, 'input_dim_b': 3, 'n_gen_res_blk': 3, 'n_gen_front_blk': 3, 'res_dropout_ratio': 0.5}).to(cuda)
    xa = torch.randn(4, 128, 4096, generator=gen(31)).to(cuda)
    xb = torch.randn(4, 128, 4096, generator=gen(32)).to(cuda)
    ya, yb = m(xa, xb)
    assert tuple(ya.shape) == (4, 3, 128, 128) and tuple(yb.shape) == (4, 3, 128, 128)
    assert float(ya.abs().max()) <= 1.0 and torch.isfinite(ya).all()
    (ya.mean() + yb.mean()).backward()
    g = m.decode_A[0].model[0].weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0


def test_adversarial_loss_matches_per_cluster_bce(cuda):
    """A.adversarial_loss == the reference's sum over clusters of F.binary_cross_entropy(torch.sigmoid(d)[c], label) terms
    (tools/faster_rcnn_train_val.py:584-600), forward and gradient, incl. row weights, full-size labels and a saturated logit"""
    from scda_amd import autograd_ops as A
    C, n = 4, 1024
    d1 = torch.randn(C, n, generator=gen(51)) * 2; d2 = torch.randn(C, n, generator=gen(52)) * 2
    d1[0, 0] = 40.0; d1[1, 1] = -40.0                    # sigmoid saturates: the -100 log clamp / 1e-12 clamp paths
    t1 = torch.rand(1, n, generator=gen(53)) * 0.2 + 0.8; t0 = torch.rand(1, n, generator=gen(54)) * 0.3
    tf = torch.rand(C, n, generator=gen(55))
    w = torch.rand(C, generator=gen(56))
    r1, r2 = d1.clone().requires_grad_(), d2.clone().requires_grad_()
    p1, p2 = torch.sigmoid(r1), torch.sigmoid(r2)
    ref = 0.0
    for c in range(C):
        ref = ref + (F.binary_cross_entropy(p1[c:c + 1], t1) + w[c] * F.binary_cross_entropy(p2[c:c + 1], t0))
    ref = (ref + F.binary_cross_entropy(p1, tf)) * 0.5
    ref.backward()
    g1, g2 = d1.to(cuda).requires_grad_(), d2.to(cuda).requires_grad_()
    out = A.adversarial_loss([(g1, t1.to(cuda), None), (g2, t0.to(cuda), w.to(cuda))], scale=0.5) + \
        A.adversarial_loss([(g1, tf.to(cuda), None)], scale=0.5 / C)
    out.backward()
    assert abs(float(out) - float(ref)) <= 2e-6 * abs(float(ref))
    close(g1.grad, r1.grad, 2e-6); close(g2.grad, r2.grad, 2e-6)


def test_seeded_dropout_equals_mask_path(cuda):
    """DropoutSeededFn (keep decisions recomputed from the seed in both passes) == dropout_mask + dropout_apply with that seed;
    with relu_input the backward additionally applies x > 0"""
    from scda_amd import autograd_ops as A, native as N
    x = torch.randn(512, 300, generator=gen(61)).to(cuda)
    dy = torch.randn(512, 300, generator=gen(62)).to(cuda)
    seed, p = 0x1234ABCD5678, 0.5
    mask = N.dropout_mask(tuple(x.shape), p, seed, cuda)
    for relu_input in (False, True):
        xa = (x.relu() if relu_input else x).clone().requires_grad_()
        ya = A.DropoutSeededFn.apply(xa, p, seed, relu_input); ya.backward(dy)
        assert torch.equal(ya, N.dropout_apply(xa.detach(), mask, 2.0))
        want = N.dropout_apply(dy, mask, 2.0)
        if relu_input:
            want = want * (xa.detach() > 0)
        assert torch.equal(xa.grad, want)
    assert 0.45 < float(mask.float().mean()) < 0.55


def test_act_fusion_plan_gives_identical_gradients(cuda):
    """layers.plan_act_fusion moves each fused ReLU / LeakyReLU gradient into its consumer (next conv's data gradient, pool
    backward, dropout backward): parameter and input gradients must be BIT-identical to the un-fused backward"""
    from scda_amd import layers as L
    from scda_amd.autograd_ops import ACT_LEAKY, ACT_RELU

    def conv_chain():
        return torch.nn.Sequential(
            L.Conv2d(3, 16, 3, padding=1, fused_act=ACT_RELU), L.FusedAct(), L.Conv2d(16, 32, 3, padding=1, fused_act=ACT_RELU), L.FusedAct(),
            L.MaxPool2x2(), L.Conv2d(32, 32, 3, stride=2, padding=1, fused_act=ACT_LEAKY), L.FusedAct("LeakyReLU"),
            L.Conv2d(32, 48, 3, padding=1, fused_act=ACT_LEAKY), L.FusedAct("LeakyReLU"), L.Conv2d(48, 1, 1))

    def fc_chain():
        return torch.nn.Sequential(L.Linear(64, 96, fused_act=ACT_RELU), L.FusedAct(), L.Dropout(),
                                   L.Linear(96, 80, fused_act=ACT_RELU), L.FusedAct(), L.Dropout(), L.Linear(80, 7))

    for make, shape in ((conv_chain, (2, 3, 24, 40)), (fc_chain, (33, 64))):
        torch.manual_seed(7)
        plain = make().to(cuda)
        fused = make().to(cuda)
        fused.load_state_dict(plain.state_dict())
        n = L.plan_act_fusion(fused)
        assert n == (4 if make is conv_chain else 2)
        x = torch.randn(*shape, generator=gen(70)).to(cuda)
        outs = []
        for net in (plain, fused):
            torch.manual_seed(11)                         # same dropout seeds
            xi = x.clone().requires_grad_()
            y = net(xi)
            y.square().sum().backward()
            outs.append((y.detach(), xi.grad, [p.grad.clone() for p in net.parameters()]))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        for a, b in zip(outs[0][2], outs[1][2]):
            assert torch.equal(a, b)


def test_lazily_zeroed_dense_gradients(cuda):
    """FlatParams does not zero-fill large 2-D weights (FC6 / FC7 class, >= 8 M elements) before a backward pass: the first weight-
    gradient GEMM of the phase overwrites the slice, later ones accumulate, and finalize_grads() zeroes what nobody wrote.
    Gradients must equal the plain zero-fill + accumulate result bit for bit, phase after phase."""
    from scda_amd import layers as L
    from scda_amd.flat import FlatParams
    torch.manual_seed(5)
    net = torch.nn.Sequential(L.Linear(4096, 2048), L.Linear(2048, 16)).to(cuda)     # 8.4 M-element weight: lazy; the rest: eager
    flat = FlatParams(net)
    assert [tuple(p.shape) for p in flat.lazy] == [(2048, 4096)]
    x1 = torch.randn(64, 4096, generator=gen(81)).to(cuda); x2 = torch.randn(64, 4096, generator=gen(82)).to(cuda)
    for phase in range(3):
        flat.zero_grad()
        assert flat.fresh
        if phase == 2:                      # a phase in which the big weight gets NO gradient: finalize must clear the stale values
            flat.finalize_grads()
            assert float(net[0].weight.grad.abs().sum()) == 0.0
            continue
        net(x1).square().sum().backward()
        net(x2).sum().backward()            # second contribution of the same phase accumulates
        assert not flat.fresh
        w = net[0].weight.detach().clone().requires_grad_(); b = net[0].bias.detach().clone().requires_grad_()
        w2 = net[1].weight.detach().clone().requires_grad_(); b2 = net[1].bias.detach().clone().requires_grad_()
        ref = lambda x: F.linear(F.linear(x, w, b), w2, b2)  # noqa: E731
        (ref(x1).square().sum() + ref(x2).sum()).backward()
        close(net[0].weight.grad, w.grad, 2e-5); close(net[1].weight.grad, w2.grad, 2e-5); close(net[0].bias.grad, b.grad, 2e-5)
    # zero_grad() on the bucket, then Module.zero_grad() (set_to_none) on top: autograd now builds the big weight's gradient OUTSIDE the
    # bucket (LinearFn has no sink, the `fresh` token is never taken); check_aliases() copies it in -- and finalize_grads() must not wipe
    # the slice it has just filled (it did: FC6 / FC7 would have trained on weight decay alone)
    flat.zero_grad()
    net.zero_grad(set_to_none=True)
    assert net[0].weight.grad is None and flat.fresh
    net(x1).square().sum().backward()
    outside = net[0].weight.grad.clone()
    flat.check_aliases(); flat.finalize_grads()
    assert flat._inside(net[0].weight.grad, flat.grad) and not flat.fresh
    assert float(outside.abs().sum()) > 0 and torch.equal(net[0].weight.grad, outside)


@pytest.mark.parametrize("shape", [(4, 128, 64, 64), (2, 3, 128, 128), (1, 2, 256, 256), (2, 5, 9, 13)])
def test_resblock_tail_fused_equals_three_launches(cuda, shape):
    """x + Dropout(InstanceNorm(h)) as ONE launch each way (autograd_ops.InstNormDropAddFn: the register-resident plane kernels for
    64x64 / 128x128 / 256x256 maps, the generic one otherwise) against InstanceNormFn -> DropoutSeededFn -> AddFn: outputs and both
    gradients bit for bit"""
    from scda_amd import autograd_ops as A
    g = gen(91)
    h = torch.randn(*shape, generator=g).to(cuda); x = torch.randn(*shape, generator=g).to(cuda); dy = torch.randn(*shape, generator=g).to(cuda)
    seed, p = 0x1234567890ABCDE, 0.5
    h1, x1 = h.clone().requires_grad_(), x.clone().requires_grad_()
    y1 = A.InstNormDropAddFn.apply(h1, x1, 1e-5, p, seed); y1.backward(dy)
    h2, x2 = h.clone().requires_grad_(), x.clone().requires_grad_()
    y2 = A.AddFn.apply(A.DropoutSeededFn.apply(A.InstanceNormFn.apply(h2, 1e-5, A.ACT_NONE, 0.0), p, seed, False), x2); y2.backward(dy)
    assert torch.equal(y1, y2) and torch.equal(h1.grad, h2.grad) and torch.equal(x1.grad, x2.grad)
    kept = float((y1 != x).float().mean())
    assert 0.4 < kept < 0.6, kept


def test_ins_res_block_uses_fused_tail(cuda, monkeypatch):
    """the decoder's residual block with and without the fused tail (SCDA_NO_RESBLOCK_TAIL_FUSION=1): same seed draw, same results"""
    from scda_amd.dropin.models.faster_rcnn.common_net import INSResBlock
    torch.manual_seed(3)
    blk = INSResBlock(16, 16, dropout=0.5).to(cuda).train()
    x = torch.randn(2, 16, 64, 64, generator=gen(92)).to(cuda)
    outs = []
    for off in ("", "1"):
        if off:
            monkeypatch.setenv("SCDA_NO_RESBLOCK_TAIL_FUSION", off)
        torch.manual_seed(7)
        xi = x.clone().requires_grad_()
        y = blk(xi); y.square().sum().backward()
        outs.append((y.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in blk.parameters()], torch.rand(1).item()))
        blk.zero_grad()
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(u, v) for u, v in zip(a[2], b[2])) and a[3] == b[3]


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("shape", [(4, 128, 64, 64), (4, 64, 128, 128), (2, 3, 32, 128), (1, 2, 128, 32)])
def test_norm_and_upsample_in_one_launch_equals_two(cuda, shape, act, monkeypatch):
    """Upsample2x(act(InstanceNorm(x))) as ONE launch each way (autograd_ops.InstanceNormUpFn: the normalised plane goes to LDS, the
    same workgroup writes the 2H x 2W map; backward: the bilinear gather hands the small plane's gradient to the norm's backward in
    registers) against InstanceNormFn -> Upsample2xFn: output and gradient bit for bit"""
    from scda_amd import autograd_ops as A, native as N
    g = gen(93)
    x = torch.randn(*shape, generator=g).to(cuda)
    dy = torch.randn(shape[0], shape[1], 2 * shape[2], 2 * shape[3], generator=g).to(cuda)
    assert N.instnorm_up2_ok(x)
    x1 = x.clone().requires_grad_()
    y1 = A.InstanceNormUpFn.apply(x1, 1e-5, act, 0.01); y1.backward(dy)
    x2 = x.clone().requires_grad_()
    y2 = A.Upsample2xFn.apply(A.InstanceNormFn.apply(x2, 1e-5, act, 0.01)); y2.backward(dy)
    assert y1.shape == y2.shape and torch.equal(y1, y2) and torch.equal(x1.grad, x2.grad)
    monkeypatch.setenv("SCDA_NO_NORM_UP_BWD_FUSION", "1")          # the forward fused, the backward as the two launches
    x3 = x.clone().requires_grad_()
    A.InstanceNormUpFn.apply(x3, 1e-5, act, 0.01).backward(dy)
    assert torch.equal(x3.grad, x2.grad)


@pytest.mark.parametrize("shape", [(4, 128, 64, 64), (2, 5, 128, 128)])
def test_resblock_tail_and_upsample_in_one_launch_equals_two(cuda, shape):
    """Upsample2x(x + Dropout(InstanceNorm(h))) as ONE launch (InstNormDropAddUpFn) against InstNormDropAddFn -> Upsample2xFn, with the
    seed as an integer and as a device slot: outputs and both gradients bit for bit"""
    from scda_amd import autograd_ops as A
    g = gen(94)
    h = torch.randn(*shape, generator=g).to(cuda); x = torch.randn(*shape, generator=g).to(cuda)
    dy = torch.randn(shape[0], shape[1], 2 * shape[2], 2 * shape[3], generator=g).to(cuda)
    p, seed = 0.5, 0x0FEDCBA987654321
    for sd in (seed, torch.tensor([seed], dtype=torch.int64, device=cuda)):
        h1, x1 = h.clone().requires_grad_(), x.clone().requires_grad_()
        y1 = A.InstNormDropAddUpFn.apply(h1, x1, 1e-5, p, sd); y1.backward(dy)
        h2, x2 = h.clone().requires_grad_(), x.clone().requires_grad_()
        y2 = A.Upsample2xFn.apply(A.InstNormDropAddFn.apply(h2, x2, 1e-5, p, sd)); y2.backward(dy)
        assert torch.equal(y1, y2) and torch.equal(h1.grad, h2.grad) and torch.equal(x1.grad, x2.grad)


def test_norm_upsample_unsupported_shapes_fail_loudly(cuda):
    from scda_amd import native as N
    x = torch.randn(1, 2, 48, 48, device=cuda)
    assert not N.instnorm_up2_ok(x)
    with pytest.raises(RuntimeError, match="not supported"):
        N.instnorm_up2_fwd(x, 1e-5, 0, 0.0)
    x = torch.randn(1, 1, 64, 64 + 1, device=cuda)[..., 1:]       # contiguous-looking shape on an offset, unaligned view
    assert not N.instnorm_up2_ok(x)


def test_decoder_fuses_norm_upsample_pairs_with_identical_results(cuda, monkeypatch):
    """a decoder branch (residual blocks -> two up-sampling blocks -> 1x1 -> tanh) with the norm + Interpolate pairs as one launch each
    (common_net.pair_decoder_upsamples: both Interpolates have a paired producer) and with SCDA_NO_NORM_UP_FUSION=1: same seed draws,
    outputs, input gradient and every parameter gradient bit for bit; a deep copy runs un-fused until paired again"""
    import copy
    from scda_amd import layers as L
    from scda_amd.dropin.models.faster_rcnn.faster_rcnn_adver_expansion_reweight_cluster import GAN_decoder_AE
    torch.manual_seed(5)
    dec = GAN_decoder_AE({'input_dim_b': 3, 'ch': 16, 'n_gen_res_blk': 2, 'n_gen_front_blk': 3, 'res_dropout_ratio': 0.5}).to(cuda).train()
    ups = [m for m in dec.decode_A.modules() if isinstance(m, L.Upsample2x)]
    assert len(ups) == 2 and all(u._producer is not None and u._producer() is not None for u in ups)
    xa = torch.randn(4, 16, 64 * 64, generator=gen(95)).to(cuda); xb = torch.randn(4, 16, 64 * 64, generator=gen(96)).to(cuda)
    calls = []
    real = L.A.Upsample2xFn.apply
    monkeypatch.setattr(L.A.Upsample2xFn, "apply", staticmethod(lambda *a: (calls.append(1), real(*a))[1]))
    outs = []
    for off in ("", "1"):
        if off:
            monkeypatch.setenv("SCDA_NO_NORM_UP_FUSION", off)
        torch.manual_seed(11)
        a, b = xa.clone().requires_grad_(), xb.clone().requires_grad_()
        n0 = len(calls)
        ya, yb = dec(a, b)
        outs.append((len(calls) - n0, ya.detach().clone(), yb.detach().clone()))
        (ya.square().sum() + yb.square().sum()).backward()
        outs[-1] += (a.grad.clone(), b.grad.clone(), [p.grad.clone() for p in dec.parameters()], torch.rand(1).item())
        dec.zero_grad()
    f, u = outs
    assert f[0] == 0 and u[0] == 4, (f[0], u[0])         # fused: no stand-alone bilinear launch; un-fused: two per branch
    assert ya.shape == (4, 3, 256, 256)
    assert torch.equal(f[1], u[1]) and torch.equal(f[2], u[2]) and torch.equal(f[3], u[3]) and torch.equal(f[4], u[4])
    assert all(torch.equal(p, q) for p, q in zip(f[5], u[5])) and f[6] == u[6]
    monkeypatch.delenv("SCDA_NO_NORM_UP_FUSION")
    clone = copy.deepcopy(dec)
    n0 = len(calls)
    clone(xa, xb)
    assert len(calls) - n0 == 4          # weak references do not travel: the copy is correct and un-fused


def test_upsample_rejects_a_tensor_that_is_not_the_announced_one(cuda):
    from scda_amd import layers as L
    up = L.Upsample2x()
    up.expect_upsampled((1, 2, 8, 8))
    with pytest.raises(RuntimeError, match="announced"):
        up(torch.zeros(1, 2, 4, 4, device=cuda))
    assert up(torch.ones(1, 2, 4, 4, device=cuda)).shape == (1, 2, 8, 8)       # the announcement is consumed either way
