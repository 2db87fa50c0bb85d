"""Region-proposal head with the constructor / attribute / state_dict names of the reference's models/head.py:3-32
(`NaiveRpnHead(inplanes, num_classes, num_anchors)`, parameters `conv3x3`, `conv_cls`, `conv_loc`).

One 3x3 conv whose ReLU is applied in the MFMA kernel's epilogue, followed by two sibling 1x1 convs: per-anchor class
scores [B, A*num_classes, h, w] and per-anchor box deltas [B, A*4, h, w].  `relu3x3` is kept as a parameter-free
placeholder so that module listings read like the reference's."""
import torch.nn as nn

from scda_amd import layers as L
from scda_amd.autograd_ops import ACT_RELU


class NaiveRpnHead(nn.Module):
    def __init__(self, inplanes, num_classes, num_anchors):
        super().__init__()
        self.num_classes = num_classes
        self.num_anchors = num_anchors
        self.conv3x3 = L.Conv2d(inplanes, 512, kernel_size=3, stride=1, padding=1, fused_act=ACT_RELU)
        self.relu3x3 = L.FusedAct("ReLU")
        self.conv_cls = L.Conv2d(512, num_anchors * num_classes, kernel_size=1, stride=1)
        self.conv_loc = L.Conv2d(512, num_anchors * 4, kernel_size=1, stride=1)

    def forward(self, x):
        hidden = self.relu3x3(self.conv3x3(x))
        return self.conv_cls(hidden), self.conv_loc(hidden)
