"""Winograd F(2x2, 3x3) convolution (csrc/conv_wino.hip, scda_conv2d_wino_hip) vs a plain PyTorch fp32 CPU reference of the same
op -- forward (bias + activation) and data gradient (with and without the fused act' mask) -- and vs the direct implicit-GEMM kernel.
Tolerance: max|err| / max|ref| < 2e-4, the bound of tests/test_conv_gemm_gpu.py (the transforms use the exact constants 0, +-1,
+-1/2; what differs from the direct form is the association of the sums)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(a, b, tol=2e-4):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    assert a.shape == b.shape
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item() / scale
    assert err < tol, f"relative-to-max error {err:.3e}"
    return err


WINO_CASES = [
    # B, Cin, H, W, Cout
    (1, 8, 8, 32, 64),        # one workgroup, one slab
    (1, 16, 16, 64, 64),      # 2 x 2 pixel blocks: every border of a block is an image border or an inner seam
    (1, 64, 24, 96, 128),     # two m-tiles, odd number of block rows
    (2, 128, 16, 32, 128),    # batch 2 (blocks never straddle images), split-K (few tiles)
    (1, 72, 8, 64, 40),       # ragged M (40 of 64 rows), 9 slabs (odd count: the unrolled K loop's tail)
    (1, 512, 32, 64, 512),    # conv5_x / RPN: split-K 4
    (4, 128, 64, 64, 128),    # the decoders' residual convolutions
    (1, 256, 128, 256, 256),  # conv3_2 at full size
    (1, 64, 6, 20, 64),       # one PARTIAL block: 6 of 8 rows, 20 of 32 columns
    (2, 128, 50, 84, 128),    # ResNet layer3's 50 x 84 maps: partial blocks on the right and bottom edges, batch 2
    (1, 32, 100, 168, 72),    # layer2's 100 x 168
    (3, 64, 70, 200, 72),     # 378 tiles of 64 rows: the PERSISTENT form on three images, partial blocks on both edges, a partial second m-tile
]


@pytest.mark.parametrize("case", WINO_CASES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_wino_fwd(cuda, case, act):
    from scda_amd import native
    B, Cin, H, W, Cout = case
    assert native.lib().scda_conv2d_wino_supported(B, Cin, H, W, Cout)
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, stride=1, padding=1)
    ref = [lambda t: t, F.relu, lambda t: F.leaky_relu(t, 0.01)][act](ref)
    wd = w.to(cuda)
    u = native.conv2d_wino_pack(wd, False)
    y = native.conv2d_wino(x.to(cuda), u, b.to(cuda), Cout, act, 0.01)
    close(y, ref)
    if act == 0:   # and against the direct kernel (same arithmetic, other association): tighter
        close(y, native.conv2d_fwd(x.to(cuda), wd, b.to(cuda), 1, 1, 0, 0.01), 5e-5)


@pytest.mark.parametrize("case", WINO_CASES)
@pytest.mark.parametrize("masked", [False, True])
def test_wino_dgrad(cuda, case, masked):
    from scda_amd import native
    B, Cin, H, W, Cout = case
    if Cout % 8:
        pytest.skip("the data gradient reduces over Cout: needs Cout % 8 == 0")
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).requires_grad_()
    xin = F.leaky_relu(x, 0.2) if masked else x         # x is the output of a LeakyReLU whose gradient the conv's dgrad applies
    y = F.conv2d(xin, w, None, stride=1, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    u = native.conv2d_wino_pack(w.detach().to(cuda), True)
    src = xin.detach().to(cuda).contiguous() if masked else None
    dx = native.conv2d_wino(dy.to(cuda), u, None, Cin, 0, 0.0, src, 0.2, for_dgrad=True)
    close(dx, x.grad)


def test_wino_routes_through_conv2d_entry_points(cuda, monkeypatch):
    """the default (SCDA_WINOGRAD unset or 1): native.conv2d_fwd / conv2d_dgrad take the Winograd kernel for eligible layers (scda_prof sees its class)"""
    from scda_amd import native
    monkeypatch.setenv("SCDA_WINOGRAD", "1")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 128, 32, 64, generator=g); w = torch.randn(128, 128, 3, 3, generator=g) / 34.0; b = torch.randn(128, generator=g)
    names = native.prof_kernel_names()
    native.prof_enable(["conv_wino_kernel<fwd>", "conv_wino_kernel<dgrad>"])
    y = native.conv2d_fwd(x.to(cuda), w.to(cuda), b.to(cuda), 1, 1, 1)
    dx = native.conv2d_dgrad(y, w.to(cuda), x.shape, 1, 1)
    torch.cuda.synchronize()
    native.prof_enable(False)
    prof = native.prof_collect()
    assert prof["conv_wino_kernel<fwd>"][0] == 1 and prof["conv_wino_kernel<dgrad>"][0] == 1, (prof, names)
    close(y, F.relu(F.conv2d(x, w, b, padding=1)))
    # a layer below the channel threshold, a strided one and a stacked-map one stay on the direct kernel
    assert not native.wino_ok(1, 16, 32, 64, 64, 3, 3, 1, 1) and not native.wino_ok(1, 64, 32, 64, 16, 3, 3, 1, 1)
    assert not native.wino_ok(1, 128, 32, 64, 128, 3, 3, 2, 1)
    assert not native.wino_ok(1, 128, 32, 64, 128, 3, 3, 1, 1, row_period=8)
    assert native.wino_ok(1, 128, 30, 64, 128, 3, 3, 1, 1) and not native.wino_ok(1, 128, 31, 64, 128, 3, 3, 1, 1)      # any EVEN height / width


WGRAD_CASES = [
    # B, Cin, H, W, Cout
    (1, 64, 2, 16, 64),        # one slab: every halo row / column is off the image
    (1, 64, 8, 32, 64),        # 4 x 2 slabs: all four edges and the interior
    (2, 64, 6, 48, 128),       # two images (the slab cursor wraps rows and images), two output-channel tiles
    (1, 128, 16, 32, 72),      # ragged Cout (72 = 64 + 8), two input-channel tiles
    (1, 72, 4, 16, 64),        # ragged Cin
    (4, 128, 64, 64, 128),     # the decoders' residual convolutions
    (1, 256, 64, 128, 512),    # conv4_1
    (1, 64, 4, 6, 64),         # a single PARTIAL slab: 3 real tiles of 8
    (2, 128, 50, 84, 128),     # ResNet layer3's 50 x 84: 5 slabs + a partial one (4 pixels) per tile row
    (1, 64, 100, 168, 128),    # layer2's 100 x 168: partial slab of 8 pixels
    (4, 64, 64, 64, 32),       # 32 output channels (the decoders' 64 -> 32 stage): half of the 64-row tile is zero rows
    (1, 40, 32, 32, 48),       # fewer than 64 channels on both sides, ragged
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wino_wgrad(cuda, case):
    from scda_amd import native
    B, Cin, H, W, Cout = case
    assert native.lib().scda_conv2d_wino_wgrad_supported(B, Cin, H, W, Cout)
    g = torch.Generator().manual_seed(sum(case) + 2)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).requires_grad_()
    b = torch.randn(Cout, generator=g).requires_grad_()
    y = F.conv2d(x, w, b, stride=1, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dw, db = native.conv2d_wino_wgrad(dy.to(cuda), x.to(cuda), w.shape, want_bias=True)
    close(dw, w.grad)
    close(db, b.grad)
    # accumulation into existing buffers (the gradient bucket), and the bias-less form
    dw2, db2 = native.conv2d_wino_wgrad(dy.to(cuda), x.to(cuda), w.shape, out=dw.clone(), db_out=db.clone())
    close(dw2, 2 * w.grad); close(db2, 2 * b.grad)
    dw3, none = native.conv2d_wino_wgrad(dy.to(cuda), x.to(cuda), w.shape)
    assert none is None and torch.equal(dw3, dw)      # deterministic: same splits, same order


def test_wino_wgrad_routes_through_conv2d_entry_points(cuda):
    from scda_amd import native
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 128, 32, 64, generator=g); dy = torch.randn(1, 128, 32, 64, generator=g)
    native.prof_enable(["conv_wino_wgrad_kernel"])
    dw, db = native.conv2d_wgrad_bias(dy.to(cuda), x.to(cuda), (128, 128, 3, 3), 1, 1)
    dw1 = native.conv2d_wgrad(dy.to(cuda), x.to(cuda), (128, 128, 3, 3), 1, 1)
    torch.cuda.synchronize()
    native.prof_enable(False)
    assert native.prof_collect()["conv_wino_wgrad_kernel"][0] == 2
    assert torch.equal(dw, dw1)
    assert not native.wino_wgrad_ok(1, 16, 32, 64, 64, 3, 3, 1, 1) and not native.wino_wgrad_ok(1, 64, 32, 64, 64, 3, 3, 2, 1)
    assert native.wino_wgrad_ok(1, 32, 32, 64, 64, 3, 3, 1, 1) and native.wino_wgrad_ok(1, 64, 32, 64, 32, 3, 3, 1, 1)     # from 32 channels a side
    assert native.wino_wgrad_ok(1, 64, 32, 72, 64, 3, 3, 1, 1) and not native.wino_wgrad_ok(1, 64, 32, 71, 64, 3, 3, 1, 1)


def test_conv_pool_fusion_is_bit_identical_to_separate_launches(cuda, monkeypatch):
    """conv3x3 + ReLU + MaxPool2d(2, 2) in the Winograd kernel's epilogue (scda_conv2d_wino_pool_hip, layers.Conv2d.pool_next) against
    the same chain with the pool as its own launch (SCDA_CONV_POOL_FUSE=0): pooled values, winners and every gradient bit for bit --
    ties inside a window included (ReLU zeros) -- and against torch on the CPU at the kernels' usual tolerance."""
    import torch.nn as nn
    from scda_amd import layers as L
    from scda_amd import native
    from scda_amd.autograd_ops import ACT_RELU
    g = torch.Generator().manual_seed(21)
    x0 = torch.randn(1, 64, 256, 256, generator=g)
    up = torch.randn(1, 32, 64, 64, generator=g)

    def build():
        torch.manual_seed(5)
        seq = nn.Sequential(L.Conv2d(64, 64, 3, padding=1, fused_act=ACT_RELU), L.FusedAct(), L.Conv2d(64, 128, 3, padding=1, fused_act=ACT_RELU),
                            L.FusedAct(), L.MaxPool2x2(), L.Conv2d(128, 32, 3, padding=1, fused_act=ACT_RELU), L.FusedAct(), L.MaxPool2x2())
        L.plan_act_fusion(seq)
        return seq.to(cuda)

    res = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("SCDA_CONV_POOL_FUSE", fuse)
        net = build()
        assert net[2].pool_next and net[5].pool_next and not net[0].pool_next
        x = x0.to(cuda).requires_grad_()
        native.prof_enable(["conv_wino_kernel<fwd>"])
        y = net(x)
        torch.cuda.synchronize()
        native.prof_enable(False)
        launches = native.prof_collect()["conv_wino_kernel<fwd>"][0]
        (y * up.to(cuda)).sum().backward()
        res[fuse] = (y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in net.parameters()], launches)
    assert res["1"][3] == res["0"][3] == 3
    assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1])
    for a, b in zip(res["1"][2], res["0"][2]):
        assert torch.equal(a, b)
    # ... and the chain itself against torch (CPU, fp32)
    net = build().cpu()
    ref = nn.Sequential(nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 128, 3, padding=1), nn.ReLU(), nn.MaxPool2d(2, 2),
                        nn.Conv2d(128, 32, 3, padding=1), nn.ReLU(), nn.MaxPool2d(2, 2))
    ref.load_state_dict(net.state_dict(), strict=True)       # same Sequential indices: convs at 0, 2, 5
    xr = x0.clone().requires_grad_()
    yr = ref(xr)
    (yr * up).sum().backward()
    close(res["1"][0], yr)
    # the input gradient in relative L2: a pre-activation within round-off of zero may flip its ReLU / pool decision between two
    # correct implementations, which moves single elements by far more than rounding (tests/test_train_step_gpu.py discusses it)
    d = (res["1"][1].cpu().double() - xr.grad.double()).norm() / xr.grad.double().norm()
    assert float(d) < 2e-3, float(d)


@pytest.mark.parametrize("case,gm", [((1, 64, 64, 128, 512), 2), ((1, 64, 64, 128, 512), 4), ((1, 128, 64, 128, 256), 2),
                                     ((1, 128, 64, 128, 256), 4), ((2, 64, 64, 64, 128), 2), ((2, 64, 64, 64, 128), 4)])
def test_wino_xcd_split_orders_give_the_same_result(cuda, case, gm, monkeypatch):
    """the launch order (which XCD runs which (m-tile, pixel block): csrc/conv_wino.hip wino_launch) is a pure renumbering of the
    workgroups: every forced split of the XCDs over m-tile groups x pixel-block runs gives the bits of the plain order"""
    from scda_amd import native
    B, Cin, H, W, Cout = case
    g = torch.Generator().manual_seed(sum(case) + gm)
    x = torch.randn(B, Cin, H, W, generator=g).to(cuda)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(cuda)
    b = torch.randn(Cout, generator=g).to(cuda)
    u = native.conv2d_wino_pack(w, False)
    monkeypatch.setenv("SCDA_WINO_GM", "0")
    y0 = native.conv2d_wino(x, u, b, Cout, 1, 0.0)
    assert native.wino_last_order()[0][1:3] == (False, 1)
    monkeypatch.setenv("SCDA_WINO_GM", str(gm))
    y1 = native.conv2d_wino(x, u, b, Cout, 1, 0.0)
    assert native.wino_last_order()[0][1:3] == (True, gm)        # the split order WAS taken
    assert torch.equal(y0, y1)
    close(y1, F.relu(F.conv2d(x.cpu(), w.cpu(), b.cpu(), stride=1, padding=1)))


@pytest.mark.parametrize("case,splits", [((1, 64, 32, 64, 512), 2), ((1, 64, 32, 64, 512), 4), ((1, 128, 16, 64, 256), 2),
                                         ((1, 512, 32, 64, 512), 4), ((2, 64, 16, 64, 128), 4)])
def test_wino_wgrad_split_groups_give_the_same_result(cuda, case, splits, monkeypatch):
    """2 or 4 K-splits: one split and one m-tile group per XCD (conv_wino_wgrad_kernel's index decode) -- the same partial slabs,
    the same fixed-order reduce: bit-identical to the launch dealt over the XCDs as it comes"""
    from scda_amd import native
    B, Cin, H, W, Cout = case
    g = torch.Generator().manual_seed(sum(case) + splits)
    x = torch.randn(B, Cin, H, W, generator=g); dy = torch.randn(B, Cout, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).requires_grad_()
    bias = torch.zeros(Cout, requires_grad=True)
    F.conv2d(x, w, bias, stride=1, padding=1).backward(dy)
    monkeypatch.setenv("SCDA_WINO_WGRAD_SPLITS", str(splits))
    monkeypatch.setenv("SCDA_WINO_WGRAD_NO_GROUPS", "1")
    dw0, db0 = native.conv2d_wino_wgrad(dy.to(cuda), x.to(cuda), w.shape, want_bias=True)
    assert native.wino_last_order()[1] == (splits, 0)
    monkeypatch.delenv("SCDA_WINO_WGRAD_NO_GROUPS")
    dw1, db1 = native.conv2d_wino_wgrad(dy.to(cuda), x.to(cuda), w.shape, want_bias=True)
    assert native.wino_last_order()[1] == (splits, 2)            # one split + one m-tile group per XCD
    assert torch.equal(dw0, dw1) and torch.equal(db0, db1)
    close(dw1, w.grad); close(db1, bias.grad)



@pytest.mark.parametrize("case", [(1, 256, 128, 256, 256), (3, 72, 70, 200, 72), (1, 64, 256, 512, 64)])
def test_wino_persistent_form_is_bit_identical_to_one_tile_per_workgroup(cuda, case, monkeypatch):
    """launches of more 64-row tiles than CUs run as one persistent workgroup per CU that walks the tiles (next tile's patches requested
    before the epilogue, stores left to drain under the next K loop): forward (bias + activation), data gradient with the fused
    activation mask, and conv + pool -- same bits as the one-tile-per-workgroup launch, and the form is really taken"""
    from scda_amd import native
    B, Cin, H, W, Cout = case
    g = torch.Generator().manual_seed(sum(case) + 9)
    x = torch.randn(B, Cin, H, W, generator=g).to(cuda)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(cuda)
    b = torch.randn(Cout, generator=g).to(cuda)
    dy = torch.randn(B, Cout, H, W, generator=g).to(cuda)
    uf, ud = native.conv2d_wino_pack(w, False), native.conv2d_wino_pack(w, True)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SCDA_WINO_PERSIST", mode)
        y = native.conv2d_wino(x, uf, b, Cout, 1, 0.01)
        assert native.wino_last_persistent() == (mode == "1") and native.wino_last_order()[0][0] == 2
        dx = native.conv2d_wino(dy, ud, None, Cin, mask_src=x, mask_slope=0.1, for_dgrad=True)
        assert native.wino_last_persistent() == (mode == "1")
        yp = native.conv2d_wino_pool(x, uf, b, Cout, 1, 0.01) if H % 4 == 0 and W % 4 == 0 else (y, y)
        got[mode] = (y, dx) + tuple(yp)
    for a, c in zip(got["1"], got["0"]):
        assert torch.equal(a, c)


@pytest.mark.parametrize("case,splits", [((1, 256, 128, 256, 256), 2), ((1, 128, 64, 128, 128), 2), ((3, 72, 70, 200, 72), 3)])
def test_wino_forced_split_on_a_launch_of_more_tiles_than_cus(cuda, case, splits, monkeypatch):
    """SCDA_WINO_SPLITS on a layer of more 64-row tiles than CUs (the persistent form's territory; the automatic heuristic never splits
    there): the launch must take the one-tile split form + reduce and give the unsplit launch's result to rounding -- a persistent
    launch has no split-slab epilogue, and once launched nothing and summed an uninitialised workspace (advisor finding, round 5)"""
    from scda_amd import native
    B, Cin, H, W, Cout = case
    g = torch.Generator().manual_seed(sum(case) + 17)
    x = torch.randn(B, Cin, H, W, generator=g).to(cuda)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(cuda)
    b = torch.randn(Cout, generator=g).to(cuda)
    u = native.conv2d_wino_pack(w, False)
    y0 = native.conv2d_wino(x, u, b, Cout, 1, 0.0)
    was_persistent = native.wino_last_persistent()
    monkeypatch.setenv("SCDA_WINO_SPLITS", str(splits))
    native.workspace(1 << 20, x.device).view(torch.float32).fill_(float("nan"))     # a launch that skipped the kernel would sum these
    y1 = native.conv2d_wino(x, u, b, Cout, 1, 0.0)
    assert native.wino_last_order()[0][3] == splits and not native.wino_last_persistent(), (native.wino_last_order(), was_persistent)
    assert torch.isfinite(y1).all()
    close(y1, y0, 5e-5)
    close(y1, F.relu(F.conv2d(x.cpu(), w.cpu(), b.cpu(), stride=1, padding=1)))


@pytest.mark.parametrize("case", [(512, 64, 64), (37, 72, 40), (6, 512, 512), (130, 128, 256)])
@pytest.mark.parametrize("masked", [False, True])
def test_wino_on_stacked_7x7_maps(cuda, case, masked, monkeypatch):
    """x [1, C, R * 7, 7] as a stack of R independent 7 x 7 maps (row period 7: the ResNet-50 C4 detector's channel-major RoI head,
    models/mask_rcnn/resnet.py:131-148): four maps per pixel block as 8 x 8 each.  Forward (bias + ReLU) and data gradient (with /
    without the activation mask) against torch's batched convolution on [R, C, 7, 7], and the Winograd path is the one taken."""
    from scda_amd import native
    R, Cin, Cout = case
    g = torch.Generator().manual_seed(R + Cin + Cout)
    xb = torch.randn(R, Cin, 7, 7, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    dyb = torch.randn(R, Cout, 7, 7, generator=g)
    stack = lambda t: t.permute(1, 0, 2, 3).reshape(1, t.shape[1], R * 7, 7).contiguous()        # [R, C, 7, 7] -> [1, C, R * 7, 7]
    unstack = lambda t: t.reshape(t.shape[1], R, 7, 7).permute(1, 0, 2, 3)
    assert native.wino_ok(1, Cin, R * 7, 7, Cout, 3, 3, 1, 1, 7)
    y = native.conv2d_fwd(stack(xb).to(cuda), w.to(cuda), b.to(cuda), 1, 1, 1, 0.0, row_period=7)
    close(unstack(y.cpu()), F.relu(F.conv2d(xb, w, b, padding=1)))
    xg = xb.clone().requires_grad_()
    F.conv2d(xg, w, None, padding=1).backward(dyb)
    want = xg.grad * ((xb > 0).float() + 0.1 * (xb <= 0).float()) if masked else xg.grad
    dx = native.conv2d_dgrad(stack(dyb).to(cuda), w.to(cuda), (1, Cin, R * 7, 7), 1, 1, act_src=stack(xb).to(cuda) if masked else None,
                             act_slope=0.1, row_period=7)
    close(unstack(dx.cpu()), want)
    # ... and against the direct kernel on the same stack
    monkeypatch.setenv("SCDA_WINO_STACKED", "0")
    assert not native.wino_ok(1, Cin, R * 7, 7, Cout, 3, 3, 1, 1, 7)
    y0 = native.conv2d_fwd(stack(xb).to(cuda), w.to(cuda), b.to(cuda), 1, 1, 1, 0.0, row_period=7)
    close(y, y0, 5e-5)


@pytest.mark.parametrize("case", [(512, 64, 64), (37, 72, 80), (7, 128, 256)])
def test_wino_wgrad_on_stacked_7x7_maps(cuda, case, monkeypatch):
    """the weight (+ bias) gradient on stacks of 7 x 7 maps: a K-slab = one tile row of a pair of maps (odd map counts: the last
    pair's second map is masked), against torch's batched convolution on [R, C, 7, 7] and the direct kernel on the same stack"""
    from scda_amd import native
    R, Cin, Cout = case
    g = torch.Generator().manual_seed(R + Cin + Cout + 3)
    xb = torch.randn(R, Cin, 7, 7, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).requires_grad_()
    b = torch.zeros(Cout, requires_grad=True)
    dyb = torch.randn(R, Cout, 7, 7, generator=g)
    F.conv2d(xb, w, b, padding=1).backward(dyb)
    stack = lambda t: t.permute(1, 0, 2, 3).reshape(1, t.shape[1], R * 7, 7).contiguous().to(cuda)
    assert native.wino_wgrad_ok(1, Cin, R * 7, 7, Cout, 3, 3, 1, 1, 7)
    dw, db = native.conv2d_wgrad_bias(stack(dyb), stack(xb), w.shape, 1, 1, row_period=7)
    close(dw, w.grad); close(db, b.grad, 2e-5)
    dw2, db2 = native.conv2d_wgrad_bias(stack(dyb), stack(xb), w.shape, 1, 1, out=dw.clone(), db_out=db.clone(), row_period=7)
    close(dw2, 2 * w.grad); close(db2, 2 * b.grad, 2e-5)
    monkeypatch.setenv("SCDA_WINO_STACKED", "0")
    assert not native.wino_wgrad_ok(1, Cin, R * 7, 7, Cout, 3, 3, 1, 1, 7)
    close(dw, native.conv2d_wgrad_bias(stack(dyb), stack(xb), w.shape, 1, 1, row_period=7)[0], 5e-5)
