"""BASELINE.json configs[3]: resnet50_FasterRCNN + SCDA at 800 x 1333 (padded to 800 x 1344, a multiple of the stride) --
performance configuration (the reference has no runnable model for it, see dropin/models/mask_rcnn/resnet.py).

The SCDA step is the VGG one (scda_amd.train_step.ScdaTrainer) with the detector swapped:
  * RoI feature = the 2048-d pooled layer4 output; 4 clusters x 128 RoIs x 2048 unfold to [4, 128, 32, 64] maps
    (`--neww/--newh` of tools/faster_rcnn_train_val.py:151-154 with neww * newh = 2048),
  * the decoders reconstruct 128 x 256 patches (two x2 up-sampling stages), the image discriminators see 128 x 256 crops
    around the cluster centres, the patch discriminator sees the 32 x 64 maps.
Work per iteration (2 FLOP / MAC, conv + FC only, necessary work as in SURVEY.md 8d), at 800 x 1344:
  backbone conv1..layer3 forward 2 images, RPN head, RoIAlign head (layer4 on 512 RoIs) forward 2 x, backward of the
  trainable part (layer2, layer3, RPN, layer4, heads) -- computed by `f_iter_tflop()` from the layer shapes."""
import torch

H, W = 800, 1344
# kernel class that holds most of the iteration's device time (profiles/r02_resnet50_kernel_stats.md): the 1x1 convolutions of
# layer2 / layer3 / the RoI head, forward, on the 64x64 tile the planner gives one-tap layers (86 launches, 9.2 ms per iteration).
# bench.py times it with event pairs in a separate pass after the timed region (events are queue markers: timing every GEMM-class
# launch inside the region cost 5 ms of the 53 this configuration was first measured at).
DOMINANT = "conv_igemm_glds_kernel<64,*,1,1,1,fwd>"
FEAT_HW = (32, 64)            # view of the 2048-d RoI feature (h, w)
RECON_HW = (128, 256)


MASK_ROIS = 64                # positive RoIs per image the mask branch trains on (train_mask_target.batch_size_per_image)
MASK_TARGET = {'positive_iou_thresh': 0.5, 'batch_size_per_image': MASK_ROIS, 'label_h': 28, 'label_w': 28, 'append_gts': True}


def build_models(cfg, cluster_num=4, threshold=128, with_mask=False, mask_iou=None):
    """with_mask: BASELINE.json configs[4] -- the same detector with the mask branch of models/mask_rcnn/resnet.py:146-149
    (RoIAlignAvg(14, 14), four 3x3 convs 1024 -> 256, 2x2/2 transposed conv, 1x1 conv to per-class 28 x 28 masks) and its loss added
    to the detector's; the step takes ground-truth masks (`ScdaTrainer.step(..., gt_masks=)`)"""
    from scda_amd.dropin.models.mask_rcnn.resnet import resnet50
    from scda_amd.train_step import builder_gan
    shared = dict(cfg['shared'], roi_align=True, gan_model_flag=2)
    if with_mask:
        shared.update(with_mask=True, train_mask_target=dict(MASK_TARGET))
        if mask_iou is not None:      # synthetic runs: an untrained RPN puts few proposals on the objects (bench.py says what it used)
            shared['train_mask_target']['positive_iou_thresh'] = mask_iou
    det = resnet50(cfg=shared)
    dis, dec, dis_patch = builder_gan(cluster_num, threshold, 256, neww=FEAT_HW[0], newh=FEAT_HW[1])
    return det, dec, dis, dis_patch


def make_trainer(cfg, device, lr=1.25e-5, world_size=1, with_mask=False, mask_iou=None):
    from scda_amd.train_step import ScdaTrainer
    return ScdaTrainer(cfg, device, lr=lr, new_w=W, new_h=H, world_size=world_size,
                       models=build_models(cfg, with_mask=with_mask, mask_iou=mask_iou), recon_hw=RECON_HW)


def synth_masks(gts, h=H, w=W):
    """[1, G, h, w] uint8: the ellipse inscribed in every ground-truth box (synthetic instance masks)"""
    import numpy as np
    g = gts[0].numpy() if torch.is_tensor(gts) else gts[0]
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.zeros((1, g.shape[0], h, w), dtype=np.uint8)
    for i, (x1, y1, x2, y2) in enumerate(g[:, :4]):
        if x2 > x1 and y2 > y1:
            out[0, i] = ((((xx - (x1 + x2) / 2.0) / ((x2 - x1) / 2.0)) ** 2 + ((yy - (y1 + y2) / 2.0) / ((y2 - y1) / 2.0)) ** 2) <= 1.0)
    return torch.from_numpy(out)


def mask_branch_tflop(rois=MASK_ROIS):
    """forward + backward (dgrad + wgrad) of the mask head on `rois` RoIs, TFLOP (2 FLOP / MAC)"""
    macs = rois * (196 * 9 * (1024 * 256 + 3 * 256 * 256) + 196 * 256 * 256 * 4 + 784 * 256 * 9)
    return 2.0 * 3 * macs / 1e12


def _bottleneck_macs(cin, planes, hw_in, stride, first):
    """(forward MACs, MACs of convs whose input needs a gradient)"""
    hw_out = hw_in // (stride * stride)
    m = cin * planes * hw_in + 9 * planes * planes * hw_out + planes * planes * 4 * hw_out
    if first:
        m += cin * planes * 4 * hw_out
    return m, hw_out


def f_iter_tflop(rois=512):
    """necessary conv / FC work of one iteration (1 source + 1 target image), TFLOP"""
    hw = (H // 4) * (W // 4)                                  # after the stem: 200 x 336
    stem = 3 * 49 * 64 * (H // 2) * (W // 2)
    macs = {}
    cin = 64
    for name, planes, blocks, stride in (("layer1", 64, 3, 1), ("layer2", 128, 4, 2), ("layer3", 256, 6, 2)):
        tot = 0
        for b in range(blocks):
            m, hw = _bottleneck_macs(cin, planes, hw, stride if b == 0 else 1, b == 0)
            tot += m
            cin = planes * 4
        macs[name] = tot
    feat_hw = hw                                              # 50 x 84
    rpn = (1024 * 9 * 512 + 512 * (30 + 60)) * feat_hw
    head, cin4, hw4 = 0, 1024, 49
    for b in range(3):
        m, hw4 = _bottleneck_macs(cin4, 512, hw4, 1, b == 0)
        head += m
        cin4 = 2048
    head = head * rois + rois * 2048 * (9 + 36)
    fwd = 2 * (stem + macs["layer1"] + macs["layer2"] + macs["layer3"] + rpn + head)
    # backward (source only): dgrad + wgrad for layer2 (no dgrad into the frozen layer1 for its first convs), layer3, RPN, layer4 + heads
    bwd = 2 * (macs["layer2"] + macs["layer3"] + rpn + head)
    scda = 222e9 * (RECON_HW[0] * RECON_HW[1]) / (256 * 256)  # decoders / discriminators scale with the patch area (222 GMAC at 256 x 256)
    return 2.0 * (fwd + bwd + scda) / 1e12
