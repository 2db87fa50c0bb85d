"""VGG-16 Faster R-CNN detector -- module API of models/faster_rcnn/vgg_adver_expansion_cluster.py
(VGG :30-98, make_layers :101-114, cfg :120-125, vgg16 :173-183, vgg16_bn :185-195).

Same attribute names and state_dict keys (features.{0..28}, rpn_head.*, classifier.{0,3}, fc_rcnn_{cls,loc}), so
torchvision's vgg16-397923af.pth and reference checkpoints load unchanged.  Every conv+ReLU pair is one fp32-MFMA
kernel (the ReLU slots of the Sequential are placeholders), FC6/FC7 fuse bias+ReLU in the GEMM epilogue."""
import math

import torch.nn as nn

from scda_amd import layers as L
from scda_amd.autograd_ops import ACT_NONE, ACT_RELU
from scda_amd.dropin.extensions import RoIPool
from scda_amd.dropin.models.faster_rcnn.faster_rcnn_adver_expansion_reweight_cluster import FasterRCNN_AdEx
from scda_amd.dropin.models.head import NaiveRpnHead

__all__ = ['VGG', 'vgg16', 'vgg16_bn']

cfg = {
    'A': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
    'B': [64, 64, 'M', 128, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
    'D': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
    'E': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M'],
}


def make_layers(plan, batch_norm=False):
    """conv3x3(+BN)+ReLU / 'M' = 2x2 max-pool, in torchvision's index layout"""
    seq, cin = [], 3
    for v in plan:
        if v == 'M':
            seq.append(L.MaxPool2x2())
        elif batch_norm:
            seq += [L.Conv2d(cin, v, kernel_size=3, padding=1), L.BatchNorm2d(v, fused_act=ACT_RELU), L.FusedAct("ReLU")]
            cin = v
        else:
            seq += [L.Conv2d(cin, v, kernel_size=3, padding=1, fused_act=ACT_RELU), L.FusedAct("ReLU")]
            cin = v
    return nn.Sequential(*seq)


class VGG(FasterRCNN_AdEx):
    # parameters whose gradients are final once the RoI-pooled features' gradient exists (everything behind the pooling): the
    # data-parallel step starts their share of the gradient all-reduce from a hook there (SegmentedReduce)
    EARLY_REDUCE_PREFIXES = ('classifier.', 'fc_rcnn_cls.', 'fc_rcnn_loc.')

    def __init__(self, features, cfg):
        super().__init__(cfg['gan_model_flag'])
        # the last pooling layer is dropped so that the feature stride is 16 (reference :38)
        self.features = nn.Sequential(*list(features.children())[:-1])
        num_anchors = len(cfg['anchor_scales']) * len(cfg['anchor_ratios'])
        self.rpn_head = NaiveRpnHead(512, num_classes=2, num_anchors=num_anchors)
        self.roipooling = RoIPool(7, 7, 1.0 / cfg['anchor_stride'])
        self.classifier = nn.Sequential(
            L.Linear(512 * 7 * 7, 4096, fused_act=ACT_RELU), L.FusedAct("ReLU"), L.Dropout(),
            L.Linear(4096, 4096, fused_act=ACT_RELU), L.FusedAct("ReLU"), L.Dropout())
        self.fc_rcnn_cls = L.Linear(4096, cfg['num_classes'])
        self.fc_rcnn_loc = L.Linear(4096, cfg['num_classes'] * 4)
        # pure chains: conv -> conv / pool and FC -> dropout apply the producer's ReLU gradient (conv5_3 and the RPN's 3x3 conv
        # feed two consumers each and keep their own elementwise pass)
        L.plan_act_fusion(self.features, self.classifier)
        self._initialize_weights()

    def feature_extractor(self, x):
        return self.features(x)

    def rpn(self, x):
        return self.rpn_head(x)

    def rcnn(self, x, rois):
        assert rois.shape[1] == 5
        pooled = self.roipooling(x, rois)          # [R, 512, 7, 7]
        hook = getattr(self, '_head_grad_hook', None)      # (set by FasterRCNN_AdEx.forward for the source pass of a data-parallel step)
        if hook is not None and pooled.requires_grad:
            pooled.register_hook(lambda g: hook())       # fires behind FC6's backward: every gradient kernel of the head is enqueued
        x_fea = self.classifier(pooled.view(pooled.size(0), -1))  # [R, 4096]
        return x_fea, self.fc_rcnn_cls(x_fea), self.fc_rcnn_loc(x_fea)

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / fan))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()


def _build(plan, batch_norm, pretrained, **kwargs):
    if pretrained:
        raise RuntimeError("no network access: load weights with utils.load_helper.load_pretrain(model, path)")
    return VGG(make_layers(cfg[plan], batch_norm=batch_norm), **kwargs)


def vgg16(pretrained=False, **kwargs):
    """VGG-D detector; kwargs: cfg=<the 'shared' section of the experiment json>"""
    return _build('D', False, pretrained, **kwargs)


def vgg16_bn(pretrained=False, **kwargs):
    return _build('D', True, pretrained, **kwargs)
