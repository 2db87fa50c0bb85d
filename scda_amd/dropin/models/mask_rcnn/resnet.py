"""ResNet C4 Faster R-CNN body -- module API of models/mask_rcnn/resnet.py:69-259 (Bottleneck :69-106, ResNet :111-259,
resnet50 :300-310), as a detector of the SCDA family: BASELINE.json configs[3] ("resnet50_FasterRCNN + SCDA, 800x1333").

The reference has NO runnable model for this configuration: `models/mask_rcnn/mask_rcnn.py` (its base class) is missing from the
repository and `tools/faster_rcnn_train_val.py:83` admits VGG only.  What exists is the body specification
followed here layer for layer: 7x7/2 stem + BN + ReLU + 3x3/2 max-pool, layer1..layer3 of bottlenecks (stride 16, 1024 channels;
torchvision's `resnet50-19c8e357.pth` key layout), `NaiveRpnHead(1024)`, `RoIAlignAvg(7, 7, 1/16)`, layer4 at stride 1 on the
RoI features, 7x7 average pool, `fc_rcnn_cls / fc_rcnn_loc` on 2048 features; stem and layer1 frozen and in eval mode
(:213-238).  To serve the SCDA step it derives from FasterRCNN_AdEx and `rcnn()` also returns the pooled 2048-d RoI feature
(the cluster-region generator's input, like VGG's FC7 output).  PERFORMANCE configuration: parity is unpinned by construction."""
import math
import os

import torch
import torch.nn as nn

from scda_amd import autograd_ops as A
from scda_amd import layers as L
from scda_amd import native as N
from scda_amd.autograd_ops import ACT_NONE, ACT_RELU
from scda_amd.dropin.extensions._roi_align.modules.roi_align import RoIAlignAvg
from scda_amd.dropin.extensions import RoIPool
from scda_amd.dropin.models.faster_rcnn.faster_rcnn_adver_expansion_reweight_cluster import FasterRCNN_AdEx
from scda_amd.dropin.models.head import NaiveRpnHead

__all__ = ['ResNet', 'Bottleneck', 'resnet50', 'resnet101']


def _frozen(conv, bn):
    return (not bn.training and not conv.weight.requires_grad and not bn.weight.requires_grad and bn.running_mean is not None
            and conv.bias is None and not os.environ.get("SCDA_RESNET_NO_FOLD"))


def folded_conv_bn(x, conv, bn, act):
    """conv -> eval-mode BN (-> ReLU) of a FROZEN pair (stem and layer1: models/mask_rcnn/resnet.py:213-238 keep them in eval mode
    with requires_grad = False) as ONE convolution: w' = w * gamma / sqrt(var + eps) per output channel, b' = beta - mean * gamma /
    sqrt(var + eps), bias and ReLU in the MFMA kernel's epilogue -- the affine map of a frozen batch norm is a constant, so the
    separate pass over the layer's output (69 MB per layer1 map at 800 x 1344) disappears.  The folded tensors are cached and
    rebuilt when any of the five source tensors changes (load_state_dict copies in place and bumps their versions)."""
    key = tuple(t._version for t in (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)) + (conv.weight.data_ptr(),)
    cache = getattr(conv, "_scda_folded", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            k = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            cache = (key, (conv.weight * k.view(-1, 1, 1, 1)).contiguous(), (bn.bias - bn.running_mean * k).contiguous())
        conv._scda_folded = cache
    return A.conv2d(x, cache[1], cache[2], conv.stride[0], conv.padding[0], act, 0.0, (None, False), conv.row_period)


class Bottleneck(nn.Module):
    """1x1 - 3x3(stride) - 1x1(x4), each followed by BN (ReLU fused into the first two BN kernels), residual join = one kernel"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = L.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = L.BatchNorm2d(planes, fused_act=ACT_RELU)
        self.conv2 = L.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = L.BatchNorm2d(planes, fused_act=ACT_RELU)
        self.conv3 = L.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = L.BatchNorm2d(planes * 4)
        self.relu = L.FusedAct("ReLU")
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if _frozen(self.conv1, self.bn1) and _frozen(self.conv2, self.bn2) and _frozen(self.conv3, self.bn3):
            out = folded_conv_bn(x, self.conv1, self.bn1, ACT_RELU)
            out = folded_conv_bn(out, self.conv2, self.bn2, ACT_RELU)
            out = folded_conv_bn(out, self.conv3, self.bn3, ACT_NONE)
            residual = x if self.downsample is None else folded_conv_bn(x, self.downsample[0], self.downsample[1], ACT_NONE)
            return A.AddReluFn.apply(out, residual)
        out = self.bn1(self.conv1(x))
        out = self.bn2(self.conv2(out))
        residual = x if self.downsample is None else self.downsample(x)
        return self.bn3.forward_add_relu(self.conv3(out), residual)      # bn3 + "out += residual" + ReLU (:95-104): one kernel


class ResNet(FasterRCNN_AdEx):
    def __init__(self, block, layers, cfg):
        super().__init__(cfg.get('gan_model_flag', 2))
        self.inplanes = 64
        self.conv1 = L.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = L.BatchNorm2d(64, fused_act=ACT_RELU)
        self.relu = L.FusedAct("ReLU")
        self.maxpool = L.MaxPool3x3s2()
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        num_anchors = len(cfg['anchor_scales']) * len(cfg['anchor_ratios'])
        self.rpn_head = NaiveRpnHead(1024, num_classes=2, num_anchors=num_anchors)
        if cfg.get('roi_align', True):
            self.roipooling = RoIAlignAvg(7, 7, 1.0 / cfg['anchor_stride'])
        else:
            self.roipooling = RoIPool(7, 7, 1.0 / cfg['anchor_stride'])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=1)     # "change stride to 1" (:140)
        self.avgpool = L.GlobalAvgPool()                                     # AvgPool2d(7) on 7x7 maps
        self.fc_rcnn_cls = L.Linear(512 * block.expansion, cfg['num_classes'])
        self.fc_rcnn_loc = L.Linear(512 * block.expansion, cfg['num_classes'] * 4)
        if cfg.get('with_keypoint'):
            raise NotImplementedError("the keypoint branch (:141-144) is not part of any BASELINE configuration")
        self.with_mask = bool(cfg.get('with_mask'))
        if self.with_mask:                                                   # mask branch (:146-149), BASELINE configs[4]
            self.mask_roipooling = RoIAlignAvg(14, 14, 1.0 / cfg['anchor_stride'])
            self.mask_head = self._make_branch(1024, 256, cfg['num_classes'], 4, upscaling=False)
            self.mask_target_cfg = dict(cfg.get('train_mask_target') or DEFAULT_MASK_TARGET, num_classes=cfg['num_classes'])
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        for head, std in ((self.rpn_head, 0.01), (self.fc_rcnn_cls, 0.001), (self.fc_rcnn_loc, 0.001)):
            for m in head.modules():
                if isinstance(m, (nn.Conv2d, nn.Linear)):
                    m.weight.data.normal_(0, std)
        self.fix_layer_num = 1
        self._fix_layer(self.fix_layer_num)
        self.tall_head = True
        self._head_convs3x3 = [m for m in self.layer4.modules() if isinstance(m, L.Conv2d) and m.kernel_size == (3, 3)]

    def _make_branch(self, inplanes, midplanes, outplanes, depth, upscaling=False):
        """FCN branch (:168-193): depth x (3x3 conv + ReLU), 2x2/2 transposed conv + ReLU, 1x1 conv to `outplanes` maps"""
        if upscaling:
            raise NotImplementedError("bilinear x2 behind the deconvolution: keypoint branch only")
        seq, cin = [], inplanes
        for _ in range(depth):
            seq.append(nn.Sequential(L.Conv2d(cin, midplanes, kernel_size=3, stride=1, padding=1, fused_act=ACT_RELU),
                                     L.FusedAct("ReLU")))
            cin = midplanes
        seq.append(nn.Sequential(L.ConvTranspose2x2s2(midplanes, midplanes), L.Activation("relu")))
        seq.append(L.Conv2d(midplanes, outplanes, kernel_size=1))
        return nn.Sequential(*seq)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(L.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                       L.BatchNorm2d(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def train(self, mode=True):
        """the stem and the fixed layers stay in eval mode (their BN uses running statistics), :213-228"""
        self.training = mode
        for module in self.children():
            module.train(mode)
        self.conv1.eval()
        self.bn1.eval()
        for layer in [self.layer1, self.layer2, self.layer3, self.layer4][:self.fix_layer_num]:
            layer.eval()
        return self

    def _fix_layer(self, layer_num):
        for mod in [self.conv1, self.bn1] + [self.layer1, self.layer2, self.layer3, self.layer4][:layer_num]:
            for p in mod.parameters():
                p.requires_grad = False

    def feature_extractor(self, x):
        if _frozen(self.conv1, self.bn1):
            x = self.maxpool(folded_conv_bn(x, self.conv1, self.bn1, ACT_RELU))
        else:
            x = self.maxpool(self.bn1(self.conv1(x)))
        return self.layer3(self.layer2(self.layer1(x)))

    def rpn(self, x):
        return self.rpn_head(x)

    def rcnn(self, x, rois):
        """RoI head.  MI355X layout: the pooled maps are produced CHANNEL-MAJOR, [C, R, 7, 7], and layer4 runs on the view
        [1, C, R*7, 7] -- per channel one contiguous row of R*49 pixels.  In the reference's [R, C, 7, 7] every channel is R
        pieces of 49 floats (196 bytes, never 16-byte aligned): the 1x1 convolutions and batch-norms gather those pieces, and the
        weight gradient cannot use the direct-to-LDS kernel at all (K-slabs of 16 pixels straddle maps).  On the tall view the
        1x1 convolutions / batch-norms / residual adds are layout-blind, the three 3x3 convolutions are told that the image is a
        stack of 7-row maps (`row_period`, include/scda_ops.h) so that no tap crosses from one RoI into the next, and the 7x7
        average pool sees R*C planes.  Same arithmetic, same results (tests/test_resnet_gpu.py compares the two paths);
        SCDA_RESNET_HEAD_NCHW=1 keeps the reference layout."""
        assert rois.shape[1] == 5
        R = rois.shape[0]
        tall = (self.tall_head and isinstance(self.roipooling, RoIAlignAvg) and R > 0 and (R * 49) % 16 == 0
                and not os.environ.get("SCDA_RESNET_HEAD_NCHW"))
        self.roipooling.channel_major = tall
        for m in self._head_convs3x3:
            m.row_period = 7 if tall else 0
        x = self.roipooling(x, rois)
        if not tall:
            x = self.layer4(x)
            x_fea = self.avgpool(x).view(x.size(0), -1)       # [R, 2048]
        else:
            C = x.shape[0]
            x = self.layer4(x.view(1, C, R * 7, 7))           # [1, 2048, R*7, 7]
            x_fea = self.avgpool(x.view(x.shape[1], R, 7, 7)).t().contiguous()   # [2048, R] -> [R, 2048]
        return x_fea, self.fc_rcnn_cls(x_fea), self.fc_rcnn_loc(x_fea)


    def mask_predictor(self, x, rois):
        """:266-270: [R, 5] RoIs -> per-class mask logits [R, num_classes, 28, 28].  Like `rcnn()` the branch runs CHANNEL-MAJOR when
        it can (R * 196 pixels a multiple of 16, i.e. R % 4 == 0): pooled maps [C, R, 14, 14] seen as one [1, C, R*14, 14] image whose
        3x3 convolutions know the 14-row period; the 2x2/2 transposed convolution's pixel shuffle maps row r*14 + h to r*28 + 2h + a,
        so the stack stays a stack.  On [R, C, 14, 14] maps of 196 pixels none of the direct-to-LDS weight-gradient kernels applies
        (measured: the branch on 64 RoIs cost 10 ms of a 55 ms iteration that way).  The returned tensor is a view in the reference's
        index order either way."""
        assert rois.shape[1] == 5
        R = rois.shape[0]
        tall = self.tall_head and R > 0 and R % 4 == 0 and not os.environ.get("SCDA_RESNET_HEAD_NCHW")
        self.mask_roipooling.channel_major = tall
        for m in self.mask_head.modules():
            if isinstance(m, L.Conv2d) and m.kernel_size == (3, 3):
                m.row_period = 14 if tall else 0
        x = self.mask_roipooling(x, rois)
        if not tall:
            return self.mask_head(x)
        y = self.mask_head(x.view(1, x.shape[0], R * 14, 14))               # [1, classes, R*28, 28]
        return y.view(y.shape[1], R, 28, 28).transpose(0, 1)

    def _extra_source_losses(self, input, feat, proposals):
        """The mask loss of the source image.  The reference's loss code for this branch is in the missing
        models/mask_rcnn/mask_rcnn.py; what exists is the target contract (functions/mask.py:73-179: labels are -1 = ignore on
        every class plane but the RoI's own).  Used here: the Mask R-CNN definition -- the mean binary cross-entropy of
        sigmoid(logits) on each positive RoI's own class plane, averaged over the RoIs."""
        if not (self.with_mask and self.training) or input.get('ground_truth_masks') is None:
            return []
        from scda_amd.dropin.functions.mask import compute_mask_targets
        masks = input['ground_truth_masks']
        if torch.is_tensor(masks) and masks.is_cuda:
            # the target generation (functions/mask.py:51-179) is host code, as in the reference: masks on the device would come back
            # with a synchronous copy behind the whole compute-stream backlog, every iteration, in front of the early detector backward
            raise ValueError("ground_truth_masks must stay on the host (pass the uint8 [1, G, H, W] tensor as the loader yields it)")
        rois, labels = compute_mask_targets(proposals, self.mask_target_cfg, input['ground_truth_bboxes'],
                                            masks, input['image_info'], input.get('ignore_regions'))
        dev = feat.device
        if rois.shape[1] < 6:                        # the all-ignore placeholder row: no positive RoI in this image
            return [feat.new_zeros(())]
        cls = rois[:, 5].long()
        r = torch.arange(rois.shape[0])
        # host -> device through pinned staging, non-blocking (a pageable .to(dev) is a synchronous copy on the compute stream)
        own = N.upload(labels[r, cls].reshape(rois.shape[0], -1).contiguous(), dev)      # [R, 28*28] in {0, 1}
        rois_dev = N.upload(rois[:, :5].contiguous(), dev)
        cls_dev = N.upload(cls, dev)
        logits = self.mask_predictor(feat, rois_dev)
        sel = logits[torch.arange(rois.shape[0], device=dev), cls_dev].reshape(rois.shape[0], -1)
        self.last_mask_rois = int(rois.shape[0])
        return [A.adversarial_loss([(sel, own, None)], scale=1.0 / rois.shape[0])]


DEFAULT_MASK_TARGET = {'positive_iou_thresh': 0.5, 'batch_size_per_image': 64, 'label_h': 28, 'label_w': 28, 'append_gts': True}


def resnet50(pretrained=False, **kwargs):
    """kwargs: cfg=<the 'shared' section of the experiment json>"""
    if pretrained:
        raise RuntimeError("no network access: load weights with scda_amd.checkpoint.load_pretrain(model, path)")
    return ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)


def resnet101(pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("no network access: load weights with scda_amd.checkpoint.load_pretrain(model, path)")
    return ResNet(Bottleneck, [3, 4, 23, 3], **kwargs)
