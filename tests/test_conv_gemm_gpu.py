"""fp32-MFMA conv / GEMM kernels vs a plain PyTorch fp32 CPU reference of the same op.
Tolerance: the kernels are exact-fp32 FMA chains in a different summation order than the CPU
reference; checked as max|err| / max|ref| < 2e-4 on outputs of O(1) magnitude."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _direct_kernels(monkeypatch):
    """this module tests the direct implicit-GEMM family: keep eligible 3x3 layers off the Winograd kernel (tests/test_conv_wino_gpu.py)"""
    monkeypatch.setenv("SCDA_WINOGRAD", "0")


def close(a, b, tol=2e-4):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    assert a.shape == b.shape
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item() / scale
    assert err < tol, f"relative-to-max error {err:.3e}"


CONV_CASES = [
    # B, Cin, H, W, Cout, k, s, p
    (1, 3, 32, 64, 64, 3, 1, 1),      # conv1_1-like: K=27 (ragged K), M=64
    (1, 64, 32, 48, 64, 3, 1, 1),
    # 3 input channels, stride 1 (VGG conv1_1): the direct vector-unit weight gradient (conv_wgrad_small_cin_kernel) -- ragged last
    # column strip, output channels that are no multiple of a wave's four / a workgroup's sixteen, several row blocks
    (2, 3, 37, 53, 40, 3, 1, 1),
    (1, 3, 50, 200, 7, 3, 1, 1),
    (1, 3, 128, 256, 64, 3, 1, 1),
    (2, 16, 20, 28, 40, 3, 1, 1),     # ragged everything, batch 2, non-pow2 extents
    (1, 128, 16, 32, 256, 3, 1, 1),   # M=256 (two 128 tiles)
    (1, 512, 8, 16, 512, 3, 1, 1),    # split-K path (few tiles)
    (4, 3, 64, 64, 32, 3, 2, 1),      # discriminator stride 2
    (4, 32, 32, 32, 64, 3, 2, 1),
    (2, 16, 15, 21, 32, 3, 2, 1),     # stride 2, odd extents: parity classes of the data gradient have unequal sizes
    (1, 16, 17, 18, 48, 3, 2, 0),     # stride 2 without padding
    (4, 128, 16, 16, 1, 1, 1, 0),     # 1x1 -> 1 channel
    (1, 512, 8, 16, 30, 1, 1, 0),     # RPN cls head
    (4, 32, 32, 32, 3, 1, 1, 0),      # decoder's final 1x1
    (1, 64, 20, 28, 128, 1, 2, 0),    # ResNet down-sampling shortcut: 1x1 stride 2
    (2, 256, 15, 21, 512, 1, 2, 0),   # ... odd extents, batch 2
    # batch-1 planes whose pixel count is a multiple of 4 but not of 16 (50 x 84 = 4200 pixels: ResNet layer3 at 800 x 1344): the
    # register-staged weight gradient with float4 dy loads
    (1, 32, 50, 84, 64, 3, 1, 1),
    (1, 64, 50, 84, 256, 1, 1, 0),
    (1, 256, 10, 14, 256, 3, 1, 1),
    (1, 32, 36, 26, 48, 3, 2, 1),     # output 18 x 13 = 234: not a multiple of 4 -> register-staged weight gradient
    (1, 32, 20, 28, 48, 3, 2, 1),     # output 10 x 14 = 140 = 8 slabs + 12 pixels
    (3, 32, 20, 28, 48, 3, 2, 1),     # stride-2 data gradient by parity classes: 3 images x 140 pixels per class (ragged last tile, image seams inside tiles)
    (2, 64, 12, 36, 512, 3, 2, 1),    # ... with a K long enough for split-K inside each class
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_conv2d_fwd(cuda, case, act):
    from scda_amd import native
    B, Cin, H, W, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    ref = [lambda t: t, F.relu, lambda t: F.leaky_relu(t, 0.01)][act](ref)
    y = native.conv2d_fwd(x.to(cuda), w.to(cuda), b.to(cuda), s, p, act, 0.01)
    close(y, ref)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_dgrad_wgrad(cuda, case):
    from scda_amd import native
    B, Cin, H, W, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).requires_grad_()
    y = F.conv2d(x, w, None, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dx = native.conv2d_dgrad(dy.to(cuda), w.detach().to(cuda), x.shape, s, p)
    close(dx, x.grad)
    dw = native.conv2d_wgrad(dy.to(cuda), x.detach().to(cuda), w.shape, s, p)
    close(dw, w.grad)
    dw2 = native.conv2d_wgrad(dy.to(cuda), x.detach().to(cuda), w.shape, s, p, out=dw.clone())
    close(dw2, 2 * w.grad)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_wgrad_with_fused_bias_grad(cuda, case):
    """(dw, db) in one pass where the shape allows the fusion, two kernels otherwise; fresh outputs and accumulation"""
    from scda_amd import native
    B, Cin, H, W, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case) + 2)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).requires_grad_()
    b = torch.randn(Cout, generator=g).requires_grad_()
    y = F.conv2d(x, w, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dw, db = native.conv2d_wgrad_bias(dy.to(cuda), x.to(cuda), w.shape, s, p)
    close(dw, w.grad); close(db, b.grad, 2e-5)
    dw2, db2 = native.conv2d_wgrad_bias(dy.to(cuda), x.to(cuda), w.shape, s, p, out=dw.clone(), db_out=db.clone())
    close(dw2, 2 * w.grad); close(db2, 2 * b.grad, 2e-5)


@pytest.mark.parametrize("shape", [(1, 3, 64, 96), (2, 3, 37, 53)])
def test_conv7x7_stride2_stem_forward(cuda, shape):
    """the ResNet stem (7x7, stride 2, padding 3, 3 -> 64): forward only, as the reference freezes it"""
    from scda_amd import native
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    close(native.conv2d_fwd(x.to(cuda), w.to(cuda), None, 2, 3), F.conv2d(x, w, None, stride=2, padding=3))


def test_conv_identity_asymmetric(cuda):
    """A = I with an asymmetric B catches a transposed C/D fragment mapping."""
    from scda_amd import native
    C = 64
    w = torch.zeros(C, C, 1, 1)
    w[torch.arange(C), torch.arange(C), 0, 0] = 1.0
    x = torch.arange(C * 8 * 16, dtype=torch.float32).reshape(1, C, 8, 16) * 1e-3
    y = native.conv2d_fwd(x.to(cuda), w.to(cuda), None, 1, 0)
    assert torch.equal(y.cpu(), x)


@pytest.mark.parametrize("M,N,K", [(512, 4096, 1024), (512, 9, 4096), (512, 36, 4096), (37, 50, 70), (128, 128, 16),
                                   (512, 4096, 25088 // 8),
                                   (512, 4096, 8192),      # weight gradient 4096 x 8192: the grouped tile order (tile_coords)
                                   (512, 2560, 8192)])     # ... with a ragged last group (20 M-tiles = 8 + 8 + 4)
def test_linear_fwd_bwd(cuda, M, N, K):
    from scda_amd import native
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g, requires_grad=True)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).requires_grad_()
    b = torch.randn(N, generator=g)
    y = F.linear(x, w, b)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    close(native.linear_fwd(x.detach().to(cuda), w.detach().to(cuda), b.to(cuda)), y)
    close(native.linear_fwd(x.detach().to(cuda), w.detach().to(cuda), b.to(cuda), act=1), F.relu(y))
    close(native.linear_dgrad(dy.to(cuda), w.detach().to(cuda)), x.grad)
    dw = native.linear_wgrad(dy.to(cuda), x.detach().to(cuda))
    close(dw, w.grad)
    close(native.linear_wgrad(dy.to(cuda), x.detach().to(cuda), out=dw.clone()), 2 * w.grad)


def test_conv_full_size_linearity(cuda):
    """Size-independent property at BASELINE size (conv1_2, 512x1024): conv(a*x1 + x2) == a*conv(x1) + conv(x2)."""
    from scda_amd import native
    torch.manual_seed(0)
    x1 = torch.randn(1, 64, 512, 1024, device=cuda); x2 = torch.randn(1, 64, 512, 1024, device=cuda)
    w = torch.randn(64, 64, 3, 3, device=cuda) / 24
    y1 = native.conv2d_fwd(x1, w, None, 1, 1)
    y = native.conv2d_fwd(2.5 * x1 + x2, w, None, 1, 1)
    y12 = 2.5 * y1 + native.conv2d_fwd(x2, w, None, 1, 1)
    close(y, y12, tol=1e-4)
    ref = F.conv2d(x1[:, :, :10, :66].cpu(), w.cpu(), None, padding=1)
    close(y1[:, :, :9, :65], ref[:, :, :9, :65])



def test_batched_weight_pack_equals_per_layer_pack(cuda):
    """native.conv2d_pack_all (one launch after every optimiser step: tiles through LDS, both sides coalesced) against the per-layer
    element-wise kernel, bit for bit, both directions: blocked layouts with full and partial 64-column tiles (M = 96 -> mpad 128, M =
    30 / 3 / 1 -> mpad 64), 1x1 and 3x3 taps, and the un-blocked layouts of channel counts that are no multiple of 16 (3, 30, 60)."""
    from scda_amd import layers as L
    from scda_amd import native
    from scda_amd.flat import FlatParams
    shapes = [(64, 3, 3), (64, 64, 3), (30, 512, 1), (60, 512, 1), (128, 64, 3), (256, 256, 3), (3, 32, 1), (1, 128, 1), (32, 64, 3),
              (96, 48, 3), (48, 96, 1), (16, 16, 3)]
    torch.manual_seed(9)
    net = torch.nn.Sequential(*[L.Conv2d(ci, co, kernel_size=k, padding=k // 2) for co, ci, k in shapes]).to(cuda)
    flat = FlatParams(net)
    assert len(flat.conv_weights) == len(shapes)
    native.conv2d_pack_all(flat)
    for w in flat.conv_weights:
        for d in (False, True):
            got = native._PACK_CACHE[(w.data_ptr(), d)][1]
            want = native.conv2d_pack_weight(w.detach().clone(), d, cache=False)
            assert got.shape == want.shape and torch.equal(got, want), (tuple(w.shape), d)
    with torch.no_grad():                      # a second step: new values, same plan
        flat.data.mul_(1.5).add_(0.25)
    flat.epoch += 1
    native.conv2d_pack_all(flat)
    w = flat.conv_weights[5]
    assert torch.equal(native._PACK_CACHE[(w.data_ptr(), True)][1], native.conv2d_pack_weight(w.detach().clone(), True, cache=False))
