"""models.losses -- API mirror of the reference's auxiliary loss zoo (models/losses.py).

None of these is on the training hot path ("API-only"): the SCDA iteration uses only cross-entropy,
smooth-L1, BCE and L1 (train_step.py).  They are kept so that `from models.losses import ...` in user scripts keeps
working; they are plain tensor expressions on whatever device the inputs live on and follow the reference formula for
formula, including its quirks (noted per class).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class Losses(nn.Module):
    """models/losses.py:14-33.  `mean(softmax(b) * (log softmax(b) - log(log_softmax(a)))) * N`.
    The reference takes the log of an already-log quantity (a negative number), so the value is NaN for any input; that
    is what a caller of the reference gets and what this returns."""

    def forward(self, input1, input2):
        la = F.log_softmax(input1, dim=1)
        pb = F.softmax(input2, dim=1)
        return (pb * (pb.log() - la.log())).mean() * la.size(0)


class Losses3(nn.Module):
    """models/losses.py:107-124: KL(softmax(b) || softmax(a)) summed, divided by the batch size"""

    def forward(self, input1, input2):
        la = F.log_softmax(input1, dim=1)
        pb = F.softmax(input2, dim=1)
        return (pb * (pb.log() - la)).sum() / la.size(0)


class Losses_triplet(nn.Module):
    """models/losses.py:35-64: 1000*KL(real||fake_target) + max-like hinge on 1 - 1000*KL(real||fake_source)
    (element-mean KL, the legacy `size_average=True`); the hinge clamps to exactly 0 when negative."""

    def forward(self, real_img, input1, input2):
        l1 = F.log_softmax(input1, dim=1)
        l2 = F.log_softmax(input2, dim=1)
        real = F.softmax(real_img, dim=1)
        positive = F.kl_div(l2, real, reduction="mean") * 1000.0
        negative = 1.0 - F.kl_div(l1, real, reduction="mean") * 1000.0
        if bool((negative.detach() < 0.0).all()):
            negative = negative * 0.0
        return positive + negative


class Losses_triplet_nll(nn.Module):
    """models/losses.py:66-89: -log( e^{d-} / (e^{d-} + e^{d+}) ) with d = MSE distances to the real image"""

    def forward(self, real_img, input1, input2):
        d_pos = F.mse_loss(input2, real_img)
        d_neg = F.mse_loss(input1, real_img)
        pt = torch.exp(d_neg) / (torch.exp(d_neg) + torch.exp(d_pos))
        return -1.0 * torch.log(pt)


class _GradReverseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lambd):
        ctx.lambd = lambd
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * -ctx.lambd, None


class GradReverse(object):
    """models/losses.py:92-101 (a legacy stateful Function there): identity forward, gradient scaled by -lambd"""

    def __init__(self, lambd):
        self.lambd = lambd

    def __call__(self, x):
        return _GradReverseFn.apply(x, self.lambd)


def grad_reverse(x, lambd):
    """models/losses.py:104-105"""
    return GradReverse(lambd)(x)


class Losses2(nn.Module):
    """models/losses.py:126-139: a bias-free nn.Bilinear used as a learned similarity"""

    def __init__(self, in1_size, in2_size, out_size):
        super(Losses2, self).__init__()
        self.loss = nn.Bilinear(in1_size, in2_size, out_size, False)

    def forward(self, input1, input2):
        return self.loss(input1, input2)


# ---- SSIM (models/losses.py:144-215) ---------------------------------------------------------------------------------
def gaussian(window_size, sigma):
    c = window_size // 2
    g = torch.tensor([math.exp(-(i - c) ** 2 / float(2 * sigma ** 2)) for i in range(window_size)], dtype=torch.float32)
    return g / g.sum()


def create_window(window_size, channel):
    g = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = (g @ g.t()).float()[None, None]
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def _ssim(img1, img2, window, window_size, channel, size_average=True):
    pad = window_size // 2

    def blur(t):
        return F.conv2d(t, window, padding=pad, groups=channel)

    mu1, mu2 = blur(img1), blur(img2)
    mu11, mu22, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s11 = blur(img1 * img1) - mu11
    s22 = blur(img2 * img2) - mu22
    s12 = blur(img1 * img2) - mu12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu11 + mu22 + c1) * (s11 + s22 + c2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def _window_for(img, window_size, channel):
    return create_window(window_size, channel).to(device=img.device, dtype=img.dtype)


class SSIM(nn.Module):
    """window cached per (channel count, dtype/device) as the reference does; default window 110"""

    def __init__(self, window_size=110, size_average=True):
        super(SSIM, self).__init__()
        self.window_size = window_size
        self.size_average = size_average
        self.channel = 1
        self.window = create_window(window_size, self.channel)

    def forward(self, img1, img2):
        channel = img1.size(1)
        if not (channel == self.channel and self.window.dtype == img1.dtype and self.window.device == img1.device):
            self.window = _window_for(img1, self.window_size, channel)
            self.channel = channel
        return _ssim(img1, img2, self.window, self.window_size, channel, self.size_average)


def ssim(img1, img2, window_size=110, size_average=True):
    channel = img1.size(1)
    return _ssim(img1, img2, _window_for(img1, window_size, channel), window_size, channel, size_average)
