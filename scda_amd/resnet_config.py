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


def build_models(cfg, cluster_num=4, threshold=128):
    from scda_amd.dropin.models.mask_rcnn.resnet import resnet50
    from scda_amd.train_step import builder_gan
    shared = dict(cfg['shared'], roi_align=True, gan_model_flag=2)
    det = resnet50(cfg=shared)
    dis, dec, dis_patch = builder_gan(cluster_num, threshold, 256, neww=FEAT_HW[0], newh=FEAT_HW[1])
    return det, dec, dis, dis_patch


def make_trainer(cfg, device, lr=1.25e-5, world_size=1):
    from scda_amd.train_step import ScdaTrainer
    return ScdaTrainer(cfg, device, lr=lr, new_w=W, new_h=H, world_size=world_size, models=build_models(cfg), recon_hw=RECON_HW)


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
