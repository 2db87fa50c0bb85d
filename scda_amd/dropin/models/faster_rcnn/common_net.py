"""GAN building blocks with the class names / constructor arguments / parameter names of
models/faster_rcnn/common_net.py (only the blocks the SCDA path instantiates: :59-80, :107-130, :160-170,
:205-245, :251-293).  Every conv / norm / activation runs on the HIP kernels; norm+activation pairs are one
kernel, conv+LeakyReLU pairs are the conv's epilogue.  nn.Sequential indices (hence state_dict keys such as
`model.0.weight`, `model.3.weight`) are identical to the reference's."""
import os

import torch
import torch.nn as nn

from scda_amd import layers as L
from scda_amd import native as N
from scda_amd import probe as P
from scda_amd import seeds
from scda_amd.autograd_ops import ACT_LEAKY, ACT_NONE, ACT_RELU, AddFn, InstNormDropAddFn, InstNormDropAddUpFn
from scda_amd.dropin.models.faster_rcnn.init import gaussian_weights_init, xavier_weights_init  # noqa: F401


class INSResBlock(nn.Module):
    """x + [conv3x3 - IN - ReLU - conv3x3 - IN - (Dropout)](x)"""

    def conv3x3(self, inplanes, out_planes, stride=1):
        return L.Conv2d(inplanes, out_planes, kernel_size=3, stride=stride, padding=1)

    def __init__(self, inplanes, planes, stride=1, dropout=0.0):
        super().__init__()
        seq = [self.conv3x3(inplanes, planes, stride), L.InstanceNorm2d(planes, fused_act=ACT_RELU), L.FusedAct("ReLU"),
               self.conv3x3(planes, planes), L.InstanceNorm2d(planes)]
        if dropout > 0:
            seq.append(L.Dropout(p=dropout))
        self.model = nn.Sequential(*seq)
        self.model.apply(gaussian_weights_init)
        self._up_ref = None      # layers.pair_norm_upsample: the Upsample2x behind this block, if this block's output feeds nothing else

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_up_ref"] = None
        return state

    def tail_fusable(self):
        """IN -> Dropout -> (+ x) of this block can run as one launch each way (training mode, a real dropout rate)"""
        tail = self.model[-2:]
        return (len(self.model) == 6 and isinstance(tail[0], L.InstanceNorm2d) and tail[0].fused_act == ACT_NONE
                and isinstance(tail[1], L.Dropout) and 0.0 < tail[1].p < 1.0 and not os.environ.get("SCDA_NO_RESBLOCK_TAIL_FUSION"))

    def forward(self, x):
        tail = self.model[-2:]
        if self.tail_fusable() and self.training and P.dropout_masks() is None and x.is_cuda:
            # IN -> Dropout -> (+ x) in one launch each way (autograd_ops.InstNormDropAddFn); the seed is drawn where the un-fused
            # Dropout module draws it, so the torch generator is consumed identically
            h = self.model[:-2](x)
            seed = seeds.draw()      # an int, or a device slot while the trainer records a hipGraph (scda_amd/seeds.py)
            up = L.my_upsample(self)
            if up is not None and N.instnorm_up2_ok(h) and x.is_contiguous() and N.aligned16(x):
                # ... and the Interpolate of the up-sampling block behind this (the decoder's last) residual block in the same launch
                y2 = InstNormDropAddUpFn.apply(h, x, tail[0].eps, tail[1].p, seed)
                up.expect_upsampled(tuple(y2.shape))
                return y2
            return InstNormDropAddFn.apply(h, x, tail[0].eps, tail[1].p, seed)
        return AddFn.apply(self.model(x), x)


def pair_decoder_upsamples(seq):
    """fusion plan of a decoder branch (an nn.Sequential of the blocks above): every LeakyReLUConvTranspose2d_2's Interpolate reads
    the output of the block in front and nothing else does, so when that output is produced by an instance norm -- the fused tail of
    the last INSResBlock, or the IN + LeakyReLU of the previous up-sampling block -- the norm's launch may write the up-sampled map
    (layers.pair_norm_upsample).  Returns the number of pairs."""
    n = 0
    blocks = list(seq.children())
    for a, b in zip(blocks, blocks[1:]):
        if not isinstance(b, LeakyReLUConvTranspose2d_2) or not isinstance(b.model[0], Interpolate):
            continue
        if isinstance(a, INSResBlock) and a.tail_fusable():
            L.pair_norm_upsample(a, b.model[0].up)
        elif isinstance(a, LeakyReLUConvTranspose2d_2) and isinstance(a.model[2], L.InstanceNorm2d):
            L.pair_norm_upsample(a.model[2], b.model[0].up)
        else:
            continue
        n += 1
    return n


class LinUnsRes_cluster(nn.Module):
    """pure reshape [cluster_num, channel, w*h] -> [cluster_num, channel, w, h]"""

    def __init__(self, channel=128, w=64, h=64, cluster_num=4):
        super().__init__()
        self.channel, self.w, self.h, self.cluster_num = channel, w, h, cluster_num

    def forward(self, x):
        return x.view(self.cluster_num, self.channel, self.w, self.h)


class LinUnsRes_cluster2(nn.Module):
    """reshape as LinUnsRes_cluster, then a bias-free 3x3 stride-2 conv: [cluster_num, channel, w/2, h/2]
    (common_net.py:132-157; first stage of GAN_decoder_AE_32)"""

    def __init__(self, channel=128, w=64, h=64, cluster_num=4):
        super().__init__()
        self.channel, self.w, self.h, self.cluster_num = channel, w, h, cluster_num
        self.model = nn.Sequential(L.Conv2d(channel, channel, kernel_size=3, stride=2, padding=1, bias=False))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        return self.model(x.view(self.cluster_num, self.channel, self.w, self.h))


class Interpolate(nn.Module):
    def __init__(self, scale_factor, mode):
        super().__init__()
        if scale_factor != 2 or mode != 'bilinear':
            raise NotImplementedError("only the x2 bilinear (align_corners=True) form is on the SCDA path")
        self.scale_factor, self.mode = scale_factor, mode
        self.up = L.Upsample2x()

    def forward(self, x):
        return self.up(x)


class ResDis_cluster(nn.Module):
    """patch discriminator trunk: [conv s2 - BN - LReLU] x2 - conv s2 - global average pool -> [cluster_num, n_out*2]"""

    def __init__(self, n_in=128, n_out=256, kernel_size=3, stride=2, padding=1, w=64, h=64, cluster_num=4):
        super().__init__()
        self.w, self.h, self.cluster_num, self.channel = w, h, cluster_num, n_in
        k = dict(kernel_size=kernel_size, stride=stride, padding=padding, bias=False)
        self.model = nn.Sequential(
            L.Conv2d(n_in, n_out, **k), L.BatchNorm2d(n_out, fused_act=ACT_LEAKY), L.FusedAct("LeakyReLU"),
            L.Conv2d(n_in * 2, n_out * 2, **k), L.BatchNorm2d(n_out * 2, fused_act=ACT_LEAKY), L.FusedAct("LeakyReLU"),
            L.Conv2d(n_out * 2, n_out * 2, **k))
        self.model.apply(gaussian_weights_init)
        self.pool = L.GlobalAvgPool()

    def forward(self, x1):
        t = self.model(x1.view(self.cluster_num, self.channel, self.w, self.h))
        return self.pool(t)  # [cluster_num, C]  (reference: AvgPool2d(full) then squeeze)


class LeakyReLUConv2d(nn.Module):
    def __init__(self, n_in, n_out, kernel_size, stride, padding=0):
        super().__init__()
        self.model = nn.Sequential(
            L.Conv2d(n_in, n_out, kernel_size=kernel_size, stride=stride, padding=padding, bias=True, fused_act=ACT_LEAKY),
            L.FusedAct("LeakyReLU"))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        return self.model(x)


class LeakyReLUConvTranspose2d_2(nn.Module):
    """bilinear x2 - conv3x3 - IN - LeakyReLU (despite the name there is no transposed conv: common_net.py:279-293)"""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0, output_padding=0):
        super().__init__()
        self.model = nn.Sequential(
            Interpolate(scale_factor=2, mode='bilinear'),
            L.Conv2d(n_in, n_out, kernel_size=kernel_size, padding=padding, stride=1, bias=True),
            L.InstanceNorm2d(n_out, fused_act=ACT_LEAKY), L.FusedAct("LeakyReLU"))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        return self.model(x)
