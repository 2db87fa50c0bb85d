"""NaiveRpnHead(inplanes, num_classes, num_anchors) -- the module API of models/head.py:3-32:
3x3 conv -> ReLU -> two sibling 1x1 convs (objectness A*num_classes, box deltas A*4).
The ReLU is fused into the 3x3 conv's MFMA epilogue; `relu3x3` stays as a named placeholder."""
import torch.nn as nn

from scda_amd import layers as L
from scda_amd.autograd_ops import ACT_RELU


class NaiveRpnHead(nn.Module):
    def __init__(self, inplanes, num_classes, num_anchors):
        super().__init__()
        self.num_anchors, self.num_classes = num_anchors, num_classes
        self.conv3x3 = L.Conv2d(inplanes, 512, kernel_size=3, stride=1, padding=1, fused_act=ACT_RELU)
        self.relu3x3 = L.FusedAct("ReLU")
        self.conv_cls = L.Conv2d(512, num_anchors * num_classes, kernel_size=1, stride=1)
        self.conv_loc = L.Conv2d(512, num_anchors * 4, kernel_size=1, stride=1)

    def forward(self, x):
        """x [B, inplanes, h, w] -> (pred_cls [B, A*num_classes, h, w], pred_loc [B, A*4, h, w])"""
        t = self.relu3x3(self.conv3x3(x))
        return self.conv_cls(t), self.conv_loc(t)
