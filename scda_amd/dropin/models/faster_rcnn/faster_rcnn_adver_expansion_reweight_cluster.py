"""Detector base class and SCDA adversarial nets -- module API of
models/faster_rcnn/faster_rcnn_adver_expansion_reweight_cluster.py (FasterRCNN_AdEx :19-235, smooth-L1 :238-246,
accuracy :249-267, GAN_dis_AE :270-308, GAN_dis_AE_patch :312-333, GAN_decoder_AE :336-399).

forward(input, target) returns the same dict (losses / accuracy / predict / cluster_features / cluster_centers).
Differences are all in *where* work runs: losses, soft-max, RoIPool, FC and convs are HIP kernels; the target image
goes through the backbone under no_grad (the reference builds an autograd graph for it that nothing ever
back-propagates: reweight_cluster.py:171-186), so its activations are not retained."""
import functools
import logging

import os
import torch
import torch.nn as nn

from scda_amd import autograd_ops as A
from scda_amd import layers as L
from scda_amd import native as N
from scda_amd._timing import mark
from scda_amd.dropin.functions.anchor_target import compute_anchor_targets
from scda_amd.dropin.functions.mask import compute_cluster_targets
from scda_amd.dropin.functions.predict_bbox import compute_predicted_bboxes
from scda_amd.dropin.functions.proposal_target import compute_proposal_targets
from scda_amd.dropin.functions.rpn_proposal import compute_rpn_proposals
from scda_amd.dropin.models.faster_rcnn.common_net import (INSResBlock, LeakyReLUConv2d, LeakyReLUConvTranspose2d_2, LinUnsRes_cluster2,
                                                           LinUnsRes_cluster, ResDis_cluster, gaussian_weights_init, pair_decoder_upsamples)

logger = logging.getLogger('global')


def smooth_l1_loss_with_sigma(pred, targets, sigma=3.0):
    """SUM of smooth-L1(pred - targets) with transition at 1/sigma^2"""
    return A.smooth_l1_sum(pred, None, targets.contiguous(), sigma, 1.0)


def accuracy(output, target, topk=(1,), ignore_index=-1):
    """top-1 precision (percent) over rows whose target != ignore_index; returns [1-element tensor] like the reference"""
    if tuple(topk) != (1,):
        raise NotImplementedError("only top-1 is used on the SCDA path")
    return [N.accuracy(output.detach().contiguous(), target.contiguous(), ignore_index)]


class _HostCopy:
    """asynchronous device->pinned-host copy of the RPN outputs, issued right behind the kernels that produce them so
    that the host can pick them up as soon as THAT image's RPN is done (not when everything queued later is)"""

    def __init__(self, *tensors):
        self.host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in tensors]
        for h, t in zip(self.host, tensors):
            h.copy_(t, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def get(self):
        self.event.synchronize()
        return self.host


def _rpn_outputs(prob, loc):
    """(objectness, deltas) on the device for the proposal generator, with asynchronous host copies attached (`_scda_host`: the
    host ranks the scores and exponentiates the size deltas, everything else stays on the device -- scda_amd.device_boxes);
    with SCDA_DEVICE_BOXES=0: the host copies alone, for the numpy path"""
    from scda_amd import device_boxes
    copy = _HostCopy(prob, loc)
    if not device_boxes.enabled():
        return copy.get
    prob._scda_host = copy.get
    return lambda: (prob, loc)


def _objectness(rpn_pred_cls):
    """soft-max over (bg, fg) per anchor, returned in the conv layout [B, 2A, h, w]"""
    nhwc = rpn_pred_cls.detach().permute(0, 2, 3, 1).contiguous()
    prob = N.row_softmax(nhwc.view(-1, 2)).view_as(nhwc)
    return prob.permute(0, 3, 1, 2)


class FasterRCNN_AdEx(nn.Module):
    def __init__(self, gan_model_flag):
        super().__init__()

    def feature_extractor(self, x):
        raise NotImplementedError

    def rpn(self, x):
        raise NotImplementedError

    def rcnn(self, x, rois):
        raise NotImplementedError

    def _add_rpn_loss(self, compute_anchor_targets_fn, rpn_pred_cls, rpn_pred_loc):
        cls_targets, loc_targets, loc_masks, loc_normalizer = compute_anchor_targets_fn(rpn_pred_loc.size())
        logits = rpn_pred_cls.permute(0, 2, 3, 1).contiguous().view(-1, 2)
        flat_t = cls_targets.permute(0, 2, 3, 1).contiguous().view(-1)
        rpn_loss_cls = A.cross_entropy(logits, flat_t, ignore_index=-1)
        # smooth-L1 of (pred * mask - target), summed, / #sampled anchors -- one fused kernel
        rpn_loss_loc = A.smooth_l1_sum(rpn_pred_loc, loc_masks, loc_targets, 3.0, 1.0 / loc_normalizer)
        acc = accuracy(logits, flat_t)[0]
        return rpn_loss_cls, rpn_loss_loc, acc

    def _add_rcnn_loss(self, rcnn_pred_cls, rcnn_pred_loc, cls_targets, loc_targets, loc_weights):
        rcnn_loss_cls = A.cross_entropy(rcnn_pred_cls, cls_targets)
        rcnn_loss_loc = A.smooth_l1_sum(rcnn_pred_loc, loc_weights, loc_targets, 3.0, 1.0 / cls_targets.shape[0])
        acc = accuracy(rcnn_pred_cls, cls_targets)[0]
        return rcnn_loss_cls, rcnn_loss_loc, acc

    def _pin_args_to_fn(self, cfg, ground_truth_bboxes, image_info, ignore_regions):
        fn = {}
        if self.training:
            fn['anchor_target_fn'] = functools.partial(compute_anchor_targets, cfg=cfg['train_anchor_target_cfg'],
                                                       ground_truth_bboxes=ground_truth_bboxes,
                                                       ignore_regions=ignore_regions, image_info=image_info)
            fn['proposal_target_fn'] = functools.partial(compute_proposal_targets, cfg=cfg['train_proposal_target_cfg'],
                                                         ground_truth_bboxes=ground_truth_bboxes,
                                                         ignore_regions=ignore_regions, image_info=image_info)
            fn['rpn_proposal_fn'] = functools.partial(compute_rpn_proposals, cfg=cfg['train_rpn_proposal_cfg'],
                                                      image_info=image_info)
        else:
            fn['rpn_proposal_fn'] = functools.partial(compute_rpn_proposals, cfg=cfg['test_rpn_proposal_cfg'],
                                                      image_info=image_info)
            fn['predict_bbox_fn'] = functools.partial(compute_predicted_bboxes, image_info=image_info,
                                                      cfg=cfg['test_predict_bbox_cfg'])
        return fn

    def _extra_source_losses(self, input, feat, proposals):
        """further losses on the source image's features (branches beyond RPN + RCNN); none in the Faster R-CNN detectors"""
        return []

    def forward(self, input, target=None):
        """input: dict(cfg, image [b,3,h,w], ground_truth_bboxes [b,G,5]|None, image_info [b,3], ignore_regions,
        cluster_num, threshold); target: target-domain image batch (training only)."""
        cfg = input['cfg']
        image = input['image']
        dev = image.device
        gts = input['ground_truth_bboxes']
        if torch.is_tensor(gts) and gts.device != dev:
            gts_host = gts
            gts = N.upload(gts_host, dev)  # the targets are produced on the device the gts live on
            gts._scda_host = gts_host.numpy()   # ... and the host-side labelling reads the boxes without a device round trip
        fn = self._pin_args_to_fn(cfg, gts, input['image_info'], input['ignore_regions'])
        outputs = {'losses': [], 'predict': [], 'accuracy': []}

        feat = self.feature_extractor(image)
        rpn_cls, rpn_loc = self.rpn(feat)
        src_host = _rpn_outputs(_objectness(rpn_cls), rpn_loc.detach()) if self.training else None
        # With the step's side stream: the target image's backbone + RPN go onto THAT stream, behind the source's, instead of into
        # the compute stream in front of the source's RCNN work.  The device still has them to work through while the host ranks
        # the source proposals, but the source RCNN forward and the detector backward no longer queue behind them, and where the
        # two streams overlap one launch's workgroups store while the other's multiply (24.1 -> 23.8 ms, same-box A/B; starting
        # the target backbone EARLIER, beside the source's, finishes the pair sooner -- 5.65 instead of 6.0 ms -- but then nothing
        # covers the host's 1.6 ms of proposal work: 25.7 ms).  The event sits behind the source's LAST layer: every layer is used
        # first on the compute stream, so a weight re-pack a first use triggers is ordered before the side stream reads it.
        # OPT-IN (SCDA_BACKBONE_SIDE=1): the iteration gains 0.3-1.4 % with it, but the dominant convolution launches then share the
        # chip with the other stream's kernels and their per-launch time -- what bench.py's roofline object reports -- rises from
        # 0.204 to 0.235 ms (0.76 -> 0.66 of the MFMA peak) without the kernels having changed.
        side0 = input.get('_side_stream') if self.training else None
        if side0 is not None and os.environ.get("SCDA_BACKBONE_SIDE", "0") != "1":
            side0 = None
        if side0 is not None:
            ev_src = torch.cuda.Event()
            ev_src.record()
            side0.wait_event(ev_src)
            with torch.cuda.stream(side0), torch.no_grad():
                feat_t = self.feature_extractor(target)
                rpn_cls_t, rpn_loc_t = self.rpn(feat_t)
                tgt_host = _rpn_outputs(_objectness(rpn_cls_t), rpn_loc_t)

        if not self.training:
            proposals = fn['rpn_proposal_fn'](_objectness(rpn_cls), rpn_loc.detach())
            rois = proposals[:, :5].to(dev).contiguous()
            assert rois.shape[1] == 5
            _, cls, loc = self.rcnn(feat, rois)
            prob = N.row_softmax(cls.detach().contiguous())
            outputs['predict'] = [rois, fn['predict_bbox_fn'](rois, prob, loc.detach())]
            return outputs

        # The target image's backbone + RPN are enqueued NOW, before any host-side box logic: the MI355X works through
        # them while the host labels anchors / sorts proposals for the source image.  Results are unaffected (no RNG in
        # these layers); every RNG-consuming call below keeps the reference's order.
        if side0 is None:
            with torch.no_grad():
                feat_t = self.feature_extractor(target)
                rpn_cls_t, rpn_loc_t = self.rpn(feat_t)
                tgt_host = _rpn_outputs(_objectness(rpn_cls_t), rpn_loc_t)
        ev_backbones = torch.cuda.Event()
        ev_backbones.record()
        mark('backbones_enqueued')

        # ---- source image: RPN loss, proposals, sampled RoIs, RCNN, cluster regions
        rpn_loss_cls, rpn_loss_loc, rpn_acc = self._add_rpn_loss(fn['anchor_target_fn'], rpn_cls, rpn_loc)
        mark('anchor_targets+rpn_loss')
        proposals = fn['rpn_proposal_fn'](*src_host())
        mark('src_proposals')
        rois, cls_targets, loc_targets, loc_weights = fn['proposal_target_fn'](proposals)
        mark('src_proposal_targets')
        assert rois.shape[1] == 5
        self._head_grad_hook = input.get('_after_head_backward')     # '_after_head_backward': see VGG.rcnn / SegmentedReduce
        try:
            x_fea, rcnn_cls, rcnn_loc = self.rcnn(feat, rois)
        finally:
            self._head_grad_hook = None
        mark('src_rcnn_enqueued')
        rcnn_loss_cls, rcnn_loss_loc, rcnn_acc = self._add_rcnn_loss(rcnn_cls, rcnn_loc, cls_targets, loc_targets, loc_weights)
        losses = [rpn_loss_cls, rpn_loss_loc, rcnn_loss_cls, rcnn_loss_loc]
        losses += self._extra_source_losses(input, feat, proposals)     # e.g. the mask branch of models/mask_rcnn/resnet.py

        # Optional scheduling hooks of this repository's own training step (absent when the reference's driver calls us):
        #  '_after_source_losses': called as soon as the four detector losses exist -- the step uses it to enqueue the
        #      detector backward (≈40 % of the iteration's device time) so that it runs underneath the host-side work below
        #      (k-means of the source RoIs, the whole target branch)
        #  '_side_stream': HIP stream for the target branch; its small sync-bound pieces (NMS, RoI sampling, FC head,
        #      k-means gather) then never queue behind that backward.
        hook = input.get('_after_source_losses')
        if hook is not None:
            hook(losses)
            mark('det_backward_enqueued')
        clu_fea, clu_ctr = compute_cluster_targets(rois, x_fea, N_cluster=input['cluster_num'], threshold=input['threshold'])
        mark('src_cluster')
        side = input.get('_side_stream')
        main = torch.cuda.current_stream(dev) if side is not None else None

        # ---- target image: same RPN / RCNN, no labels, nothing is differentiated through it
        def target_branch():
            with torch.no_grad():
                proposals_t = fn['rpn_proposal_fn'](*tgt_host())
                outputs['num_proposals_target'] = int(proposals_t.shape[0])
                rois_t_host = proposals_t[0:512, :5].contiguous()
                rois_t = N.upload(rois_t_host, dev)
                rois_t._scda_host = rois_t_host.numpy()
                assert rois_t.shape[1] == 5
                x_fea_t, _, _ = self.rcnn(feat_t, rois_t)
                return (x_fea_t,) + compute_cluster_targets(rois_t, x_fea_t, N_cluster=input['cluster_num'],
                                                            threshold=input['threshold'])

        if side is None:
            x_fea_t, clu_fea_t, clu_ctr_t = target_branch()
        else:
            side.wait_event(ev_backbones)
            with torch.cuda.stream(side):
                x_fea_t, clu_fea_t, clu_ctr_t = target_branch()
            main.wait_stream(side)
            for t in (x_fea_t, clu_fea_t):
                t.record_stream(main)
        assert feat_t.size() == feat.size(), "gan_features does not match the backbone"
        mark('target_branch')

        outputs['losses'] = losses
        outputs['accuracy'] = [rpn_acc, rcnn_acc]
        outputs['predict'] = [proposals]
        outputs['num_proposals'] = (int(proposals.shape[0]), outputs.pop('num_proposals_target', 0))   # post-NMS, source / target
        if x_fea_t.size(0) != 512:  # target image produced too few proposals: fall back to the source clusters
            logger.info("Different channels {} at target image".format(x_fea_t.size(0)))
            outputs['cluster_features'] = [clu_fea, clu_fea]
            outputs['cluster_centers'] = [clu_ctr, clu_ctr]
        else:
            outputs['cluster_features'] = [clu_fea, clu_fea_t]
            outputs['cluster_centers'] = [clu_ctr, clu_ctr_t]
        return outputs


def _two_branches(module, fa, fb):
    """run the independent A / B halves of a GAN module: sequentially (default, as the reference does), or -- when the
    owner set `module.branch_stream` (scda_amd.train_step does) -- B on that second HIP stream.  The decoder / discriminator
    convolutions are batch-4 launches of 0.5-1 workgroup per CU; two of them side by side fill the chip.  The autograd engine
    replays each half's backward on the stream its forward ran on and orders the streams at the graph edges; the two halves
    own disjoint parameters (disjoint slices of the flat gradient bucket)."""
    side = getattr(module, 'branch_stream', None)
    if side is None:
        return fa(), fb()
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    a = fa()                       # A first: the host-side order (dropout mask draws) stays the reference's A-then-B
    with torch.cuda.stream(side):
        b = fb()
    main.wait_stream(side)
    b.record_stream(main)
    return a, b


class GAN_dis_AE(nn.Module):
    """two image discriminators (A: source crops, B: target crops): n_layer stride-2 LeakyReLU convs + 1x1 -> 1"""

    def __init__(self, params):
        super().__init__()
        ch, cin, n_layer = params['ch'], params['input_dim_a'], params['n_layer']
        self.model_A = self._make_net(ch, cin, n_layer - 1)
        self.model_A.apply(gaussian_weights_init)
        self.model_B = self._make_net(ch, cin, n_layer - 1)
        self.model_B.apply(gaussian_weights_init)
        L.plan_act_fusion(self.model_A, self.model_B)   # LeakyReLU gradients applied by the next conv's data gradient

    def _make_net(self, ch, input_dim, n_layer):
        seq = [LeakyReLUConv2d(input_dim, ch, kernel_size=3, stride=2, padding=1)]
        for _ in range(n_layer):
            seq.append(LeakyReLUConv2d(ch, ch * 2, kernel_size=3, stride=2, padding=1))
            ch *= 2
        seq.append(L.Conv2d(ch, 1, kernel_size=1, stride=1, padding=0))
        return nn.Sequential(*seq)

    def forward(self, x_aa, x_bb):
        a, b = _two_branches(self, lambda: self.model_A(x_aa), lambda: self.model_B(x_bb))
        return a.view(a.size(0), -1), b.view(b.size(0), -1)


class GAN_dis_AE_patch(nn.Module):
    """per-cluster weighting net on RoI-feature clusters: ResDis_cluster trunk -> sigmoid, [cluster_num, 2*n_out]"""

    def __init__(self, params=None):
        super().__init__()
        if params:
            self.n_in, self.n_out, clusters = params['n_in'], params['n_out'], params['cluster_num']
        else:
            self.n_in, self.n_out, clusters = 128, 256, 4
        # the reference hard-codes the 64 x 64 unfolding of VGG's 4096-d RoI feature (:326); other detectors pass 'w' / 'h'
        w, h = (params or {}).get('w', 64), (params or {}).get('h', 64)
        self.model_A_patch = nn.Sequential(ResDis_cluster(n_in=self.n_in, n_out=self.n_out, kernel_size=3, stride=2,
                                                          padding=1, w=w, h=h, cluster_num=clusters))

    def forward(self, rois_features):
        return A.sigmoid(self.model_A_patch(rois_features))


class GAN_decoder_AE(nn.Module):
    """two decoders (A: source, B: target): reshape -> n_gen_res_blk INSResBlocks -> (n_gen_front_blk-1) x2
    up-sampling blocks -> 1x1 transposed conv -> tanh; cluster features [4, ch, 64*64] -> images [4, 3, S, S]"""

    first_stage = LinUnsRes_cluster

    def __init__(self, params):
        super().__init__()
        out_dim, ch = params['input_dim_b'], params['ch']
        n_res, n_front = params['n_gen_res_blk'], params['n_gen_front_blk']
        drop = params.get('res_dropout_ratio', 0)
        neww, newh, clusters = params.get('neww', 64), params.get('newh', 64), params.get('cluster_num', 4)

        def branch():
            seq = [type(self).first_stage(ch, neww, newh, clusters)]
            seq += [INSResBlock(ch, ch, dropout=drop) for _ in range(n_res)]
            c = ch
            for _ in range(n_front - 1):
                seq.append(LeakyReLUConvTranspose2d_2(c, c // 2, kernel_size=3, stride=1, padding=1, output_padding=0))
                c //= 2
            seq += [L.ConvTranspose1x1(c, out_dim), L.Activation("tanh")]
            return nn.Sequential(*seq)

        # creation order B then A, init order B then A: the reference's RNG consumption order (:368-392)
        dec_b, dec_a = branch(), None
        dec_a = branch()
        self.decode_B = dec_b
        self.decode_B.apply(gaussian_weights_init)
        self.decode_A = dec_a
        self.decode_A.apply(gaussian_weights_init)
        for dec in (self.decode_A, self.decode_B):       # norm + Interpolate pairs as one launch (common_net.pair_decoder_upsamples)
            pair_decoder_upsamples(dec)

    def forward(self, x_aa, x_bb):
        return _two_branches(self, lambda: self.decode_A(x_aa), lambda: self.decode_B(x_bb))


class GAN_decoder_AE_32(GAN_decoder_AE):
    """the same decoder behind a stride-2 entry conv (LinUnsRes_cluster2): 32x32 feature maps, images of half the size
    (faster_rcnn_adver_expansion_reweight_cluster.py:402-465)"""
    first_stage = LinUnsRes_cluster2
