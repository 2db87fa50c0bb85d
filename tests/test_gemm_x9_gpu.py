"""Dense GEMM on the bf16 matrix pipe with exact products (csrc/conv_gemm.hip gemm_x9_kernel: three bf16 pieces per fp32 value, nine
piece products per a * b, hi x hi and the eight small products in separate accumulators) -- against an fp64 GEMM, and against the
fp32-MFMA kernel it replaces for the FC-sized products (models/faster_rcnn/vgg_adver_expansion_cluster.py:73-80: FC6 / FC7).
The gate: its error against fp64 is NO WORSE than the fp32 kernel's, max and rms, in every operand layout."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (M, N, K, trans_a, trans_b): the three products of an nn.Linear (forward [M][K] x [N][K], data gradient [M][K] x [K][N], weight
# gradient [K][M] x [K][N]) + the fourth layout; ragged M / N (partial tiles on both edges), K any multiple of 16
CASES = [
    (512, 384, 1024, False, False),
    (512, 256, 25088, False, False),     # FC6's K
    (512, 640, 4096, False, True),       # FC7's K, the data gradient's layout
    (768, 896, 512, True, True),         # the weight gradient's K and layout
    (300, 200, 528, False, False),
    (260, 132, 272, False, True),
    (100, 520, 304, True, False),
    (700, 36, 160, True, True),
]


def _operands(M, N, K, ta, tb, seed, relu=False):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g)
    if relu:
        a = a.clamp_min(0)
    b = torch.randn(N, K, generator=g) / K ** 0.5
    ref = a.double() @ b.double().t()
    scale = a.double().abs() @ b.double().abs().t()      # sum_k |a b|: what round-off is proportional to
    A = a.t().contiguous() if ta else a
    B = b.t().contiguous() if tb else b
    return A, B, ref, scale


def _run(native, A, B, M, N, K, ta, tb, **kw):
    lda = M if ta else K
    ldb = N if tb else K
    return native.gemm(A, B, M, N, K, lda, ldb, ta, tb, **kw)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("relu", [False, True])
def test_x9_error_is_no_worse_than_the_fp32_mfma(cuda, case, relu, monkeypatch):
    from scda_amd import native
    M, N, K, ta, tb = case
    A, B, ref, scale = _operands(M, N, K, ta, tb, sum(case[:3]) + relu, relu)
    A, B = A.to(cuda), B.to(cuda)
    monkeypatch.setenv("SCDA_GEMM_X9", "0")
    c32 = _run(native, A, B, M, N, K, ta, tb)
    assert native.last_plan()[3] != 2
    monkeypatch.setenv("SCDA_GEMM_X9", "2")
    c9 = _run(native, A, B, M, N, K, ta, tb)
    assert native.last_plan()[3] == 2, native.last_plan()           # the bf16 x 9 kernel ran
    e32 = ((c32.cpu().double() - ref).abs() / scale)
    e9 = ((c9.cpu().double() - ref).abs() / scale)
    print("M %d N %d K %d ta %d tb %d relu %d: fp32 max %.2e rms %.2e | x9 max %.2e rms %.2e (of sum|ab|)"
          % (M, N, K, ta, tb, relu, e32.max(), e32.pow(2).mean().sqrt(), e9.max(), e9.pow(2).mean().sqrt()))
    # The gate: where the fp32 chain's own error has grown past its final rounding (K >= 2048 per split: the FC shapes) the x9 form is no
    # worse than it, max and rms.  On short K both are a fraction of one rounding of the result: rms no worse, max within 2e-7 of
    # sum|ab| (the bf16 MFMA truncates its 17-term sums where the fp32 MFMA rounds: 1.3e-7 vs 0.9e-7 at K = 304).
    assert e9.max() <= 2e-7
    assert e9.pow(2).mean().sqrt() <= e32.pow(2).mean().sqrt() * 1.05
    if K >= 2048:
        assert e9.max() <= e32.max() * 1.05


@pytest.mark.parametrize("case,splits", [((512, 384, 2048, False, False), 4), ((260, 132, 1024, False, True), 3),
                                         ((300, 256, 1040, True, True), 2)])
def test_x9_split_k_bias_activation(cuda, case, splits, monkeypatch):
    """split-K slabs + the fixed-order reduce with bias and activation; the unsplit launch with the fused epilogue; accumulate"""
    from scda_amd import native
    M, N, K, ta, tb = case
    A, B, ref, scale = _operands(M, N, K, ta, tb, 7 + splits)
    A, B = A.to(cuda), B.to(cuda)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    want = torch.relu(ref + bias.double())
    monkeypatch.setenv("SCDA_GEMM_X9", "2")
    monkeypatch.setenv("SCDA_GEMM_X9_SPLITS", str(splits))
    c = _run(native, A, B, M, N, K, ta, tb, bias=bias.to(cuda), bias_on_n=True, act=native.ACT_RELU)
    assert native.last_plan()[2] == splits and native.last_plan()[3] == 2, native.last_plan()
    assert ((c.cpu().double() - want).abs() / scale).max() <= 2e-7
    monkeypatch.setenv("SCDA_GEMM_X9_SPLITS", "1")
    c1 = _run(native, A, B, M, N, K, ta, tb, bias=bias.to(cuda), bias_on_n=True, act=native.ACT_RELU)
    assert native.last_plan()[2] == 1
    assert ((c1.cpu().double() - want).abs() / scale).max() <= 2e-7
    out = torch.ones(M, N, device=cuda)
    _run(native, A, B, M, N, K, ta, tb, out=out, accumulate=True)
    assert ((out.cpu().double() - 1.0 - ref).abs() / (scale + 1.0)).max() <= 2e-7


@pytest.mark.parametrize("case", CASES + [(1024, 1152, 512, True, True), (512, 2944, 1024, False, True)])
def test_x9_stream_k_form(cuda, case, monkeypatch):
    """the persistent stream-K launch (one workgroup per CU walking a contiguous range of (tile, slab) units; cut tiles summed by the
    fix-up kernel in a fixed order) forced on every shape -- ranges of a single slab (tiles cut into dozens of segments), ranges that
    cut tiles unevenly, ranges of whole tiles -- against fp64 and the one-workgroup-per-tile launch, with bias + activation, and twice
    over (deterministic)"""
    from scda_amd import native
    M, N, K, ta, tb = case
    A, B, ref, scale = _operands(M, N, K, ta, tb, 11 + sum(case[:3]))
    A, B = A.to(cuda), B.to(cuda)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    want = torch.nn.functional.leaky_relu(ref + bias.double(), 0.1)
    kw = dict(bias=bias.to(cuda), bias_on_n=True, act=native.ACT_LEAKY, slope=0.1)
    monkeypatch.setenv("SCDA_GEMM_X9", "2")
    monkeypatch.setenv("SCDA_GEMM_X9_SK", "0")
    c0 = _run(native, A, B, M, N, K, ta, tb, **kw)
    assert native.last_plan()[2] >= 1 and native.last_plan()[3] == 2
    monkeypatch.setenv("SCDA_GEMM_X9_SK", "2")
    c1 = _run(native, A, B, M, N, K, ta, tb, **kw)
    assert native.last_plan()[2] == -1 and native.last_plan()[3] == 2, native.last_plan()      # the stream-K launch ran
    c2 = _run(native, A, B, M, N, K, ta, tb, **kw)
    assert torch.equal(c1, c2)
    assert ((c1.cpu().double() - want).abs() / (scale + 1e-30)).max() <= 2e-7
    assert ((c1 - c0).abs().cpu().double() / (scale + 1e-30)).max() <= 2e-7
    out = torch.full((M, N), 2.0, device=cuda)
    _run(native, A, B, M, N, K, ta, tb, out=out, accumulate=True)
    assert ((out.cpu().double() - 2.0 - ref).abs() / (scale + 1.0)).max() <= 2e-7


def test_fc_sized_products_take_the_x9_kernel_by_default(cuda, monkeypatch):
    """FC7's three products (4096 x 4096 weights, 512 RoIs) route to the bf16 x 9 kernel without any switch, the heads' do not"""
    from scda_amd import native
    monkeypatch.delenv("SCDA_GEMM_X9", raising=False)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(512, 4096, generator=g).to(cuda)
    w = (torch.randn(4096, 4096, generator=g) / 64).to(cuda)
    dy = torch.randn(512, 4096, generator=g).to(cuda)
    y = native.linear_fwd(x, w, None)
    assert native.last_plan()[3] == 2
    dx = native.linear_dgrad(dy, w)
    assert native.last_plan()[3] == 2
    dw = native.linear_wgrad(dy, x)
    assert native.last_plan()[3] == 2
    xs, ws, dys = x.cpu().double(), w.cpu().double(), dy.cpu().double()
    for got, want in ((y, xs @ ws.t()), (dx, dys @ ws), (dw, dys.t() @ xs)):
        err = (got.cpu().double() - want).abs().max() / want.abs().max()
        assert err < 2e-6, float(err)
    wh = (torch.randn(36, 4096, generator=g) / 64).to(cuda)
    native.linear_fwd(x, wh, None)
    assert native.last_plan()[3] != 2


@pytest.mark.parametrize("case", [(512, 1024, 448, 7), (1024, 512, 448, 7), (256, 1024, 50, 84), (2048, 512, 112, 7)])
def test_batch1_1x1_convolutions_route_to_the_x9_gemm(cuda, case, monkeypatch):
    """a batch-1 1x1 convolution is a dense GEMM on the NCHW tensors as they lie (csrc/conv_gemm.hip conv1x1_as_x9: the ResNet-50 C4
    detector's layer3 / RoI-head bottlenecks, models/mask_rcnn/resnet.py:111-148): forward (+ bias + ReLU), data gradient and weight
    gradient (overwrite and accumulate) through the convolution entry points against torch's conv2d on the CPU, and the routing is
    really taken (forced here on shapes below the size threshold; 50 x 84 planes are not a multiple of 16: their weight gradient stays
    on the convolution kernel)"""
    import torch.nn.functional as F
    from scda_amd import native
    Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(1, Cin, H, W, generator=g).clamp_min(0)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).requires_grad_()
    b = torch.randn(Cout, generator=g)
    dy = torch.randn(1, Cout, H, W, generator=g)
    xr = x.clone().requires_grad_()
    yr = F.conv2d(xr, w, b)
    yr.backward(dy)
    monkeypatch.setenv("SCDA_GEMM_X9", "2")
    xd, wd, dyd = x.to(cuda), w.detach().to(cuda), dy.to(cuda)

    def close(a, ref, tol=2e-5):
        err = (a.cpu().double() - ref.double()).abs().max() / ref.double().abs().max()
        assert err < tol, float(err)
    y = native.conv2d_fwd(xd, wd, b.to(cuda), 1, 0, native.ACT_RELU)
    assert native.last_plan()[3] == 2, native.last_plan()
    close(y, F.relu(yr.detach()))
    dx = native.conv2d_dgrad(dyd, wd, tuple(x.shape), 1, 0)
    assert native.last_plan()[3] == 2, native.last_plan()
    close(dx, xr.grad)
    dw = native.conv2d_wgrad(dyd, xd, tuple(w.shape), 1, 0)
    assert (native.last_plan()[3] == 2) == ((H * W) % 16 == 0), native.last_plan()
    close(dw, w.grad)
    acc = torch.ones(Cout, Cin, 1, 1, device=cuda)
    native.conv2d_wgrad(dyd, xd, tuple(w.shape), 1, 0, out=acc)
    close(acc - 1.0, w.grad)
    monkeypatch.setenv("SCDA_GEMM_X9", "0")
    y0 = native.conv2d_fwd(xd, wd, b.to(cuda), 1, 0, native.ACT_RELU)
    assert native.last_plan()[3] != 2
    close(y, y0.cpu())
