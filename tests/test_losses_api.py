"""models.losses API mirror (reference models/losses.py) -- closed-form checks on CPU; these are not hot-path ops."""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scda_amd.dropin as dropin  # noqa: E402

dropin.install()
from models import losses as L  # noqa: E402


def test_kl_family():
    torch.manual_seed(0)
    a, b, r = torch.randn(5, 7), torch.randn(5, 7), torch.randn(5, 7)
    want = F.kl_div(F.log_softmax(a, 1), F.softmax(b, 1), reduction="batchmean")
    assert torch.allclose(L.Losses3()(a, b), want, atol=1e-6)
    assert torch.isnan(L.Losses()(a, b))          # the reference's log(log_softmax) quirk, models/losses.py:31-32
    t = L.Losses_triplet()(r, a, b)
    pos = F.kl_div(F.log_softmax(b, 1), F.softmax(r, 1), reduction="mean") * 1000
    neg = 1 - F.kl_div(F.log_softmax(a, 1), F.softmax(r, 1), reduction="mean") * 1000
    assert torch.allclose(t, pos + (neg if neg >= 0 else neg * 0), atol=1e-5)
    n = L.Losses_triplet_nll()(r, a, b)
    dp, dn = F.mse_loss(b, r), F.mse_loss(a, r)
    assert torch.allclose(n, -torch.log(torch.exp(dn) / (torch.exp(dn) + torch.exp(dp))))


def test_grad_reverse_and_bilinear():
    x = torch.randn(3, 4, requires_grad=True)
    y = L.grad_reverse(x, 0.25)
    assert torch.equal(y, x)
    y.sum().backward()
    assert torch.allclose(x.grad, torch.full_like(x, -0.25))
    m = L.Losses2(4, 6, 2)
    assert list(m.state_dict().keys()) == ["loss.weight"]
    assert m(torch.randn(3, 4), torch.randn(3, 6)).shape == (3, 2)


def test_ssim():
    g = L.gaussian(11, 1.5)
    assert abs(float(g.sum()) - 1) < 1e-6 and g.argmax() == 5
    assert abs(float(g[4] / g[5]) - math.exp(-1 / 4.5)) < 1e-6
    torch.manual_seed(1)
    a = torch.rand(2, 3, 24, 24)
    assert abs(float(L.ssim(a, a, window_size=7)) - 1) < 1e-5
    b = torch.rand(2, 3, 24, 24)
    s = L.SSIM(window_size=7)
    v = s(a, b)
    assert float(v) < 0.5 and s.channel == 3 and torch.allclose(v, L.ssim(a, b, window_size=7))
    assert L.SSIM(window_size=7, size_average=False)(a, b).shape == (2,)
