"""CPU oracle of the ResNet-50 C4 detector (BASELINE.json configs[3]) -- TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32 restatement of the body the reference's models/mask_rcnn/resnet.py spells out:
  Bottleneck :69-106 (1x1 - 3x3(stride) - 1x1 x4, BN after each, ReLU, residual join), ResNet :111-259 (7x7/2 stem + BN + ReLU +
  3x3/2 max-pool, layer1..layer3 -> stride 16 / 1024 channels, RoIAlignAvg(7, 7, 1/16) :131, layer4 at stride 1 on the RoI maps
  :140, AvgPool2d(7), fc_rcnn_cls / fc_rcnn_loc; conv1 / bn1 / layer1 frozen and kept in eval mode :213-238; init :150-160),
  RoIAlignAvg = RoIAlign to (7+1) x (7+1) then avg_pool2d(kernel 2, stride 1)  (extensions/_roi_align/modules/roi_align.py:18-30),
with the SCDA source / target forward, losses and cluster features of oracle.torch_ref.RefDetector (the reference's
FasterRCNN_AdEx.forward, ...reweight_cluster.py:106-235) on top.  RoIAlign itself is the C restatement of
roi_align_kernel.cu in oracle/liboracle.so.

PARITY UNPINNED against the reference: `models/mask_rcnn/mask_rcnn.py`, the base class resnet.py derives from, is missing from
the reference repository, so no reference model of this configuration can be constructed or run.  What this
file gives the tests is an implementation-independent second statement (torch-CPU convolutions / batch norms / autograd) of the
same layer specification, to compare the HIP detector's losses and gradients with."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native_ops as orc
from .torch_ref import RefDetector, RefRpnHead


class _RoIAlignFn(torch.autograd.Function):
    """extensions/_roi_align/functions/roi_align.py:7-51 on the C oracle"""

    @staticmethod
    def forward(ctx, feat, rois, ah, aw, scale):
        ctx.save_for_backward(rois)
        ctx.cfg = (tuple(feat.shape), ah, aw, scale)
        return torch.from_numpy(orc.roi_align_fwd(feat.detach().numpy(), rois.detach().numpy(), ah, aw, scale))

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        shape, ah, aw, scale = ctx.cfg
        return torch.from_numpy(orc.roi_align_bwd(g.contiguous().numpy(), rois.numpy(), shape, ah, aw, scale)), None, None, None, None


class RefRoIAlignAvg(nn.Module):
    def __init__(self, ah, aw, scale):
        super().__init__()
        self.ah, self.aw, self.scale = int(ah), int(aw), float(scale)

    def forward(self, feat, rois):
        assert rois.shape[1] == 5
        x = _RoIAlignFn.apply(feat.contiguous(), rois.contiguous(), self.ah + 1, self.aw + 1, self.scale)
        return F.avg_pool2d(x, kernel_size=2, stride=1)


class RefBottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu1, self.relu2, self.relu = nn.ReLU(), nn.ReLU(), nn.ReLU()    # three modules: one recorded selection per site
        self.downsample = downsample

    def forward(self, x):
        out = self.relu1(self.bn1(self.conv1(x)))
        out = self.relu2(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + (x if self.downsample is None else self.downsample(x)))


class RefResNetDetector(RefDetector):
    """state_dict keys = torchvision's resnet layout + rpn_head.* + fc_rcnn_{cls,loc}.*, as the product's
    dropin/models/mask_rcnn/resnet.py"""

    MASK_TARGET = {'positive_iou_thresh': 0.5, 'batch_size_per_image': 64, 'label_h': 28, 'label_w': 28, 'append_gts': True}

    def __init__(self, cfg, layers=(3, 4, 6, 3)):
        nn.Module.__init__(self)
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU()
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        A = len(cfg['anchor_scales']) * len(cfg['anchor_ratios'])
        self.rpn_head = RefRpnHead(1024, 2, A)
        self.roipooling = RefRoIAlignAvg(7, 7, 1.0 / cfg['anchor_stride'])
        self.layer4 = self._make_layer(512, layers[3], stride=1)
        self.fc_rcnn_cls = nn.Linear(2048, cfg['num_classes'])
        self.fc_rcnn_loc = nn.Linear(2048, cfg['num_classes'] * 4)
        self.with_mask = bool(cfg.get('with_mask'))
        if self.with_mask:                                              # :146-149, _make_branch(1024, 256, classes, 4) :168-193
            self.mask_roipooling = RefRoIAlignAvg(14, 14, 1.0 / cfg['anchor_stride'])
            seq, cin = [], 1024
            for _ in range(4):
                seq.append(nn.Sequential(nn.Conv2d(cin, 256, kernel_size=3, stride=1, padding=1), nn.ReLU()))
                cin = 256
            seq.append(nn.Sequential(nn.ConvTranspose2d(256, 256, kernel_size=2, stride=2, padding=0), nn.ReLU()))
            seq.append(nn.Conv2d(256, cfg['num_classes'], kernel_size=1))
            self.mask_head = nn.Sequential(*seq)
            self.mask_target_cfg = dict(cfg.get('train_mask_target') or self.MASK_TARGET, num_classes=cfg['num_classes'])
        for m in self.modules():                                        # :150-160
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2. / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        self.fix_layer_num = 1
        for mod in [self.conv1, self.bn1, self.layer1]:                 # :230-238
            for p in mod.parameters():
                p.requires_grad = False

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        seq = [RefBottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        seq += [RefBottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def train(self, mode=True):                                         # :213-228
        self.training = mode
        for module in self.children():
            module.train(mode)
        self.conv1.eval()
        self.bn1.eval()
        self.layer1.eval()
        return self

    def features(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer3(self.layer2(self.layer1(x)))

    def extra_source_losses(self, inp, feat, proposals):
        """mask loss: mean binary cross-entropy of sigmoid(logits) on each positive RoI's own class plane (Mask R-CNN's definition;
        the reference's own loss code is in the missing mask_rcnn.py), targets by functions/mask.py:73-179"""
        if not (self.with_mask and self.training) or inp.get('ground_truth_masks') is None:
            return []
        from scda_amd.dropin.functions.mask import compute_mask_targets
        rois, labels = compute_mask_targets(proposals, self.mask_target_cfg, inp['ground_truth_bboxes'], inp['ground_truth_masks'],
                                            inp['image_info'], None)
        if rois.shape[1] < 6:
            return [feat.new_zeros(())]
        r, cls = torch.arange(rois.shape[0]), rois[:, 5].long()
        self.last_mask_rois = int(rois.shape[0])
        logits = self.mask_head(self.mask_roipooling(feat, rois[:, :5].contiguous()))
        return [F.binary_cross_entropy(torch.sigmoid(logits[r, cls]), labels[r, cls])]

    def rcnn(self, x, rois):
        x = self.layer4(self.roipooling(x, rois))
        fea = F.avg_pool2d(x, 7).view(x.size(0), -1)
        return fea, self.fc_rcnn_cls(fea), self.fc_rcnn_loc(fea)
