"""torch.autograd.Function wrappers around the HIP kernels (scda_amd.native).

PyTorch is plumbing here: it owns device memory, the stream and the backward
graph walk; every forward/backward body below is one or more calls through the
C ABI of libscda_ops.so.  No Function has a CPU branch.
"""
import os

import torch
from torch.autograd import Function

from . import native as N

ACT_NONE, ACT_RELU, ACT_LEAKY = N.ACT_NONE, N.ACT_RELU, N.ACT_LEAKY


# Parity tests (scda_amd/probe.py): with a Probe carrying a `replay` object installed, each op that makes a non-differentiable
# selection -- ReLU / LeakyReLU sign masks, max-pool winners, RoI max-pool argmax -- asks it for the selection the CPU oracle made at
# the same site (matched by the output's shape and L1 norm) and differentiates through THAT instead of its own, so the gradient
# comparison measures the kernels' arithmetic, not tie-breaking.  _replay() is None outside a probed trainer step.
from .probe import replay as _replay


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _mask_src(ctx, y, kind="act"):
    """tensor whose sign the backward's act' reads: y itself, or the oracle's selection when a replay hook is installed"""
    r = _replay()
    if r is not None and any(ctx.needs_input_grad):
        return r.act(y)
    return y


def _sink(p):
    """If parameter `p` lives in a flat gradient bucket (scda_amd.flat), return its gradient view: the backward kernels
    then ACCUMULATE straight into the bucket (it was zeroed at the start of the phase) and hand autograd `None`,
    instead of materialising a gradient tensor that AccumulateGrad adds into the bucket with one more kernel
    (for FC6 that is a 411 MB temporary plus a 1.2 GB add)."""
    if p is not None and getattr(p, "_scda_flat", None) is not None and p.grad is not None and p.requires_grad:
        return p.grad
    return None


class Conv2dFn(Function):
    """conv (+bias) (+ReLU/LeakyReLU) in one MFMA kernel; backward = act' -> dgrad, wgrad, bias-grad."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, slope, fuse=(None, False), row_period=0):
        """fuse = (input_act, defer): static fusion plan of the owning model (scda_amd.layers.plan_act_fusion).
        input_act = (mode, slope) of the activation that produced x: THIS conv's data gradient applies that activation's
        gradient in its epilogue.  defer = True: the consumer of y does the same for this conv's own activation, so the
        backward here receives dy already multiplied by act'(y)."""
        x = _c(x)
        if not w.is_contiguous():
            w = w.contiguous()
        # row_period: x is a vertical stack of independent maps of that many rows (channel-major RoI head, scda_ops.h)
        y = N.conv2d_fwd(x, w, b, stride, pad, act, slope, row_period=row_period)
        in_act, defer = fuse if _replay() is None else (None, False)   # parity tests replay act masks: plain un-fused backward
        ctx.cfg = (stride, pad, act, slope, in_act, defer)
        ctx.row_period = row_period
        ctx.has_bias = b is not None
        ctx.bias_ref, ctx.w_ref = b, w   # the Parameter objects themselves (they carry the flat-bucket gradient views)
        ctx.save_for_backward(x, w, _mask_src(ctx, y) if act != ACT_NONE and not defer else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, act, slope, in_act, defer = ctx.cfg
        rp = ctx.row_period
        dy = _c(dy)
        if act != ACT_NONE and not defer:
            dy = N.act_bwd(dy, y, 0 if act == ACT_RELU else 1, slope)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if in_act is not None:
                dx = N.conv2d_dgrad(dy, w, x.shape, stride, pad, act_src=x, act_slope=0.0 if in_act[0] == ACT_RELU else in_act[1],
                                    row_period=rp)
            else:
                dx = N.conv2d_dgrad(dy, w, x.shape, stride, pad, row_period=rp)
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if want_w and want_b:   # one pass: the bias gradient is fused into the weight-gradient kernel where the shape allows
            sw, sb = _sink(ctx.w_ref), _sink(ctx.bias_ref)
            dw, db = N.conv2d_wgrad_bias(dy, x, w.shape, stride, pad, out=sw, db_out=sb, row_period=rp)
            dw = None if sw is not None else dw
            db = None if sb is not None else db
        elif want_w:
            sink = _sink(ctx.w_ref)
            dw = N.conv2d_wgrad(dy, x, w.shape, stride, pad, out=sink, row_period=rp)
            if sink is not None:
                dw = None
        elif want_b:
            sink = _sink(ctx.bias_ref)
            db = N.bias_grad_nchw(dy, out=sink)
            if sink is not None:
                db = None
        return dx, dw, db, None, None, None, None, None, None


class ConvPoolFn(Function):
    """conv3x3 (+bias) + ReLU + MaxPool2d(2, 2) as ONE launch (scda_conv2d_wino_pool_hip: a Winograd tile is a pooling window); the
    full-resolution map between them is never written.  Backward: the pool's scatter with the ReLU gradient applied through the pooled
    value (a window's winner is > 0 exactly when its maximum is), then the convolution's usual data / weight / bias gradients.  Same
    values, winners and gradients as Conv2dFn -> MaxPool2x2Fn on the fusion plan's deferred-ReLU path, bit for bit."""

    @staticmethod
    def forward(ctx, x, w, b, slope, in_act):
        x = _c(x)
        if isinstance(w, torch.nn.Parameter):
            w._scda_wino_used = True
        y, idx = N.conv2d_wino_pool(x, N.conv2d_wino_pack(w, False), b, w.shape[0], ACT_RELU, slope)
        ctx.in_act = in_act
        ctx.has_bias = b is not None
        ctx.bias_ref, ctx.w_ref = b, w
        ctx.save_for_backward(x, w, idx, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, idx, y = ctx.saved_tensors
        in_act = ctx.in_act
        dyc = N.maxpool2x2_bwd(_c(dy), idx, (x.shape[0], w.shape[0], x.shape[2], x.shape[3]), relu_y=y)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if in_act is not None:
                dx = N.conv2d_dgrad(dyc, w, x.shape, 1, 1, act_src=x, act_slope=0.0 if in_act[0] == ACT_RELU else in_act[1])
            else:
                dx = N.conv2d_dgrad(dyc, w, x.shape, 1, 1)
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if want_w and want_b:
            sw, sb = _sink(ctx.w_ref), _sink(ctx.bias_ref)
            dw, db = N.conv2d_wgrad_bias(dyc, x, w.shape, 1, 1, out=sw, db_out=sb)
            dw = None if sw is not None else dw
            db = None if sb is not None else db
        elif want_w:
            sink = _sink(ctx.w_ref)
            dw = N.conv2d_wgrad(dyc, x, w.shape, 1, 1, out=sink)
            if sink is not None:
                dw = None
        elif want_b:
            sink = _sink(ctx.bias_ref)
            db = N.bias_grad_nchw(dyc, out=sink)
            if sink is not None:
                db = None
        return dx, dw, db, None, None


class LinearFn(Function):
    @staticmethod
    def forward(ctx, x, w, b, act, defer=False):
        """defer: the consumer of y (a Dropout, see DropoutSeededFn) applies this layer's ReLU gradient"""
        x = _c(x); w = _c(w)
        y = N.linear_fwd(x, w, b, act)
        defer = defer and _replay() is None
        ctx.act = act if not defer else ACT_NONE
        ctx.has_bias = b is not None
        ctx.bias_ref, ctx.w_ref = b, w   # the Parameter objects themselves (they carry the flat-bucket gradient views)
        ctx.save_for_backward(x, w, _mask_src(ctx, y) if ctx.act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _c(dy)
        if ctx.act != ACT_NONE:
            dy = N.act_bwd(dy, y, 0, 0.0)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = N.linear_dgrad(dy, w)
        if ctx.needs_input_grad[1]:
            sink = _sink(ctx.w_ref)
            fresh = sink is not None and ctx.w_ref._scda_flat.take_fresh(ctx.w_ref)   # lazily-zeroed FC6 / FC7 slice: overwrite
            dw = N.linear_wgrad(dy, x, out=sink, accumulate=not fresh)
            if sink is not None:
                dw = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            sink = _sink(ctx.bias_ref)
            db = N.colsum(dy, out=sink)
            if sink is not None:
                db = None
        return dx, dw, db, None, None


class MaxPool2x2Fn(Function):
    @staticmethod
    def forward(ctx, x, relu_input=False):
        """relu_input: x is the output of a ReLU whose owner defers its gradient to this pool's backward (fusion plan)"""
        x = _c(x)
        y, idx = N.maxpool2x2_fwd(x)
        r = _replay()
        if r is not None and any(ctx.needs_input_grad):
            idx = r.pool(y, idx)
        fuse = relu_input and r is None
        ctx.save_for_backward(idx, y if fuse else None)
        ctx.xshape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        idx, y = ctx.saved_tensors
        return N.maxpool2x2_bwd(_c(dy), idx, ctx.xshape, relu_y=y), None


class ActFn(Function):
    """stand-alone ReLU / LeakyReLU / tanh / sigmoid (mode = N.ACT_MODE[...])"""

    @staticmethod
    def forward(ctx, x, mode, slope):
        y = N.act_fwd(_c(x), mode, slope)
        ctx.cfg = (mode, slope)
        ctx.save_for_backward(_mask_src(ctx, y) if mode in (0, 1) else y)   # tanh / sigmoid are smooth: nothing to replay
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        mode, slope = ctx.cfg
        return N.act_bwd(_c(dy), y, mode, slope), None, None


class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, mask, scale):
        ctx.save_for_backward(mask)
        ctx.scale = scale
        return N.dropout_apply(_c(x), mask, scale)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return N.dropout_apply(_c(dy), mask, ctx.scale), None, None


class DropoutSeededFn(Function):
    """nn.Dropout whose keep decisions are recomputed from a 64-bit seed in both passes (no mask tensor, one launch each way);
    relu_input: x is a ReLU output whose owner defers its gradient to this backward (fusion plan)"""

    @staticmethod
    def forward(ctx, x, p, seed, relu_input):
        x = _c(x)
        ctx.cfg = (p, seed)
        ctx.save_for_backward(x if relu_input else None)
        return N.dropout_seeded(x, p, seed, 1.0 / (1.0 - p))

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        p, seed = ctx.cfg
        return N.dropout_seeded(_c(dy), p, seed, 1.0 / (1.0 - p), relu_src=x), None, None, None


class AdvLossFn(Function):
    """scale * sum over groups of sum_c w[c] * mean_i BCE(sigmoid(logits[c, i]), labels[c or 0, i]) -> 0-dim loss.
    One launch per group in each direction (the reference: a sigmoid, a split and 2 x cluster_num BCE + add kernels per group,
    tools/faster_rcnn_train_val.py:567-600,642-690).  Called as AdvLossFn.apply(scale, n_groups, *logits, *labels, *weights);
    weights are constants (the per-cluster means of the patch discriminator are detached where the reference zeroes or never
    uses their gradient -- see scda_amd.train_step)."""

    @staticmethod
    def forward(ctx, scale, k, *args):
        logits, labels, weights = args[:k], args[k:2 * k], args[2 * k:3 * k]
        out, probs = None, []
        for x, t, w in zip(logits, labels, weights):
            out, pr = N.sigmoid_bce_rows_fwd(_c(x), _c(t), w, scale, out=out)
            probs.append(pr)
        ctx.k, ctx.scale = k, scale
        ctx.save_for_backward(*probs, *[_c(t) for t in labels], *[w if w is not None else torch.empty(0) for w in weights])
        ctx.has_w = [w is not None for w in weights]
        return out[0]

    @staticmethod
    def backward(ctx, g):
        k = ctx.k
        sv = ctx.saved_tensors
        probs, labels, weights = sv[:k], sv[k:2 * k], sv[2 * k:3 * k]
        g = _c(g).reshape(1)
        grads = [N.sigmoid_bce_rows_bwd(probs[i], labels[i], weights[i] if ctx.has_w[i] else None, ctx.scale, g)
                 if ctx.needs_input_grad[2 + i] else None for i in range(k)]
        return (None, None) + tuple(grads) + (None,) * (2 * k)


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return N.axpby(_c(a), _c(b), 1.0, 1.0)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class AddReluFn(Function):
    """relu(a + b) -- the residual join of a ResNet block; both inputs receive dy * (y > 0)"""

    @staticmethod
    def forward(ctx, a, b):
        y = N.add_relu(_c(a), _c(b))
        ctx.save_for_backward(_mask_src(ctx, y))
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        g = N.act_bwd(_c(dy), y, 0, 0.0)
        return (g if ctx.needs_input_grad[0] else None), (g if ctx.needs_input_grad[1] else None)


class MaxPool3x3s2Fn(Function):
    """nn.MaxPool2d(3, 2, 1), forward only (the ResNet stem in front of it is frozen: nothing is differentiated through it)"""

    @staticmethod
    def forward(ctx, x):
        return N.maxpool3x3s2_fwd(_c(x))

    @staticmethod
    def backward(ctx, dy):
        raise NotImplementedError("MaxPool3x3s2 has no backward: the reference freezes the stem (models/mask_rcnn/resnet.py:230-238)")


class RoIPoolFn(Function):
    """extensions/_roi_pooling/functions/roi_pool.py:6-42 (differentiable w.r.t. features only)"""

    @staticmethod
    def forward(ctx, features, rois, ph, pw, scale):
        if not features.is_contiguous() or not rois.is_contiguous():
            raise AssertionError("RoIPool needs contiguous features and rois")  # roi_pool.py:25-26
        out, arg = N.roi_pool_fwd(features, rois, ph, pw, scale)
        r = _replay()
        if r is not None and any(ctx.needs_input_grad):
            arg = r.roi(out, arg)
        ctx.save_for_backward(rois, arg)
        ctx.cfg = (tuple(features.shape), ph, pw, scale)
        return out

    @staticmethod
    def backward(ctx, dy):
        rois, arg = ctx.saved_tensors
        shape, ph, pw, scale = ctx.cfg
        return N.roi_pool_bwd(_c(dy), arg, rois, shape, ph, pw, scale), None, None, None, None


class RoIAlignFn(Function):
    """extensions/_roi_align/functions/roi_align.py:7-51"""

    @staticmethod
    def forward(ctx, features, rois, ah, aw, scale, channel_major=False):
        """channel_major: the pooled maps come out as [C,R,ah,aw] (RoI-head layout of the ResNet-C4 detector)"""
        if not features.is_contiguous() or not rois.is_contiguous():
            raise AssertionError("RoIAlign needs contiguous features and rois")
        ctx.save_for_backward(rois)
        ctx.cfg = (tuple(features.shape), ah, aw, scale, channel_major)
        return N.roi_align_fwd(features, rois, ah, aw, scale, channel_major)

    @staticmethod
    def backward(ctx, dy):
        (rois,) = ctx.saved_tensors
        shape, ah, aw, scale, channel_major = ctx.cfg
        return N.roi_align_bwd(_c(dy), rois, shape, ah, aw, scale, channel_major), None, None, None, None, None


class Avg2x2S1Fn(Function):
    """F.avg_pool2d(x, kernel_size=2, stride=1) -- the pooling half of RoIAlignAvg, one kernel each way"""

    @staticmethod
    def forward(ctx, x):
        return N.avg2x2s1_fwd(_c(x))

    @staticmethod
    def backward(ctx, dy):
        return N.avg2x2s1_bwd(_c(dy))


class SoftmaxCEFn(Function):
    """F.cross_entropy(logits, targets, ignore_index) -> 0-dim loss"""

    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        out2, probs = N.softmax_ce_fwd(_c(logits), _c(targets), ignore_index)
        ctx.save_for_backward(probs, targets, out2)
        ctx.ignore = ignore_index
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        probs, targets, out2 = ctx.saved_tensors
        return N.softmax_ce_bwd(probs, targets, out2, _c(g).reshape(1), ctx.ignore), None, None


class SmoothL1Fn(Function):
    """smooth_l1_loss_with_sigma(pred*mask, target, sigma) * scale -> 0-dim loss"""

    @staticmethod
    def forward(ctx, pred, mask, target, sigma, scale):
        pred = _c(pred)
        ctx.save_for_backward(pred, mask, target)
        ctx.cfg = (sigma, scale)
        return N.smooth_l1_fwd(pred, mask, target, sigma, scale)[0]

    @staticmethod
    def backward(ctx, g):
        pred, mask, target = ctx.saved_tensors
        sigma, scale = ctx.cfg
        return N.smooth_l1_bwd(pred, mask, target, sigma, scale, _c(g).reshape(1)), None, None, None, None


class InstanceNormFn(Function):
    @staticmethod
    def forward(ctx, x, eps, act, slope):
        x = _c(x)
        y, mean, rstd = N.instnorm_fwd(x, eps, act, slope)
        ctx.save_for_backward(x, mean, rstd)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        act, slope = ctx.cfg
        return N.instnorm_bwd(_c(dy), x, mean, rstd, act, slope), None, None, None


class InstNormDropAddFn(Function):
    """residual + Dropout_p(InstanceNorm(x)) -- the tail of INSResBlock (common_net.py:59-80) as one launch each way; the keep
    decisions come from the 64-bit seed in both passes (as DropoutSeededFn).  Bit-identical to InstanceNormFn -> DropoutSeededFn ->
    AddFn."""

    @staticmethod
    def forward(ctx, x, residual, eps, p, seed):
        x = _c(x)
        y, mean, rstd = N.instnorm_drop_add_fwd(x, _c(residual), eps, p, seed)
        ctx.save_for_backward(x, mean, rstd)
        ctx.cfg = (p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        p, seed = ctx.cfg
        dy = _c(dy)
        dx = N.instnorm_drop_bwd(dy, x, mean, rstd, p, seed) if ctx.needs_input_grad[0] else None
        return dx, (dy if ctx.needs_input_grad[1] else None), None, None, None


class InstanceNormUpFn(Function):
    """Upsample2x(act(InstanceNorm(x))) as ONE launch each way (N.instnorm_up2_fwd / instnorm_up2_bwd): the small normalised plane
    never reaches memory, its gradient passes from the bilinear gather to the norm's backward in registers (which needs x / mean /
    rstd only).  Bit-identical to InstanceNormFn -> Upsample2xFn in both directions."""

    @staticmethod
    def forward(ctx, x, eps, act, slope):
        x = _c(x)
        y2, mean, rstd = N.instnorm_up2_fwd(x, eps, act, slope)
        ctx.save_for_backward(x, mean, rstd)
        ctx.cfg = (act, slope)
        return y2

    @staticmethod
    def backward(ctx, dy2):
        x, mean, rstd = ctx.saved_tensors
        act, slope = ctx.cfg
        dy2 = _c(dy2)
        if N.aligned16(dy2) and not os.environ.get("SCDA_NO_NORM_UP_BWD_FUSION"):
            return N.instnorm_up2_bwd(dy2, x, mean, rstd, act, slope), None, None, None      # the gather and the norm's gradient in one launch
        return N.instnorm_bwd(N.upsample2x_bwd(dy2), x, mean, rstd, act, slope), None, None, None


class InstNormDropAddUpFn(Function):
    """Upsample2x(residual + Dropout_p(InstanceNorm(x))): the tail of the LAST INSResBlock of a decoder and the Interpolate of the
    up-sampling block behind it as one launch each way (the backward also writes the gathered gradient: it is the residual input's).
    Bit-identical to InstNormDropAddFn -> Upsample2xFn."""

    @staticmethod
    def forward(ctx, x, residual, eps, p, seed):
        x = _c(x)
        y2, mean, rstd = N.instnorm_drop_add_up2_fwd(x, _c(residual), eps, p, seed)
        ctx.save_for_backward(x, mean, rstd)
        ctx.cfg = (p, seed)
        return y2

    @staticmethod
    def backward(ctx, dy2):
        x, mean, rstd = ctx.saved_tensors
        p, seed = ctx.cfg
        dy2 = _c(dy2)
        if ctx.needs_input_grad[0] and N.aligned16(dy2) and not os.environ.get("SCDA_NO_NORM_UP_BWD_FUSION"):
            dx, dy = N.instnorm_drop_up2_bwd(dy2, x, mean, rstd, p, seed)
            return dx, (dy if ctx.needs_input_grad[1] else None), None, None, None
        dy = N.upsample2x_bwd(dy2)
        dx = N.instnorm_drop_bwd(dy, x, mean, rstd, p, seed) if ctx.needs_input_grad[0] else None
        return dx, (dy if ctx.needs_input_grad[1] else None), None, None, None


class BatchNormTrainFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, eps, momentum, act, slope):
        x = _c(x)
        y, mean, rstd = N.batchnorm_fwd(x, gamma, beta, run_mean, run_var, eps, momentum, act, slope)
        ctx.g_ref, ctx.b_ref = gamma, beta
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        act, slope = ctx.cfg
        sg, sb = _sink(ctx.g_ref), _sink(ctx.b_ref)
        both = sg is not None and sb is not None
        dx, dg, db = N.batchnorm_bwd(_c(dy), x, gamma, beta, mean, rstd, act, slope, need_dx=ctx.needs_input_grad[0],
                                     out=(sg, sb) if both else None)
        if both:
            dg = db = None
        return dx, dg, db, None, None, None, None, None, None


class BatchNormAddReluFn(Function):
    """relu(bn_train(x) + residual): bn3 and the residual join of a bottleneck (models/mask_rcnn/resnet.py:95-104) as one kernel each
    way.  Same arithmetic as BatchNormTrainFn followed by AddReluFn, without the normalised map's round trip through memory (forward)
    and without the separate gating pass (backward: the gated gradient is written once, for the residual branch)."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, run_mean, run_var, eps, momentum):
        x = _c(x)
        y, mean, rstd = N.batchnorm_add_relu_fwd(x, _c(residual), gamma, beta, run_mean, run_var, eps, momentum)
        ctx.g_ref, ctx.b_ref = gamma, beta
        ctx.save_for_backward(x, y, gamma, beta, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, mean, rstd = ctx.saved_tensors
        sg, sb = _sink(ctx.g_ref), _sink(ctx.b_ref)
        both = sg is not None and sb is not None
        dy = _c(dy)
        if not N.aligned16(dy):     # a contiguous storage-offset view: the kernel reads float4
            dy = dy.clone()
        dx, dres, dg, db = N.batchnorm_add_relu_bwd(dy, x, y, gamma, beta, mean, rstd, need_dx=ctx.needs_input_grad[0],
                                                    out=(sg, sb) if both else None)
        if both:
            dg = db = None
        return dx, (dres if ctx.needs_input_grad[1] else None), dg, db, None, None, None, None


class BatchNormEvalFn(Function):
    """eval-mode nn.BatchNorm2d (+ fused activation): running statistics and affine parameters are constants"""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, eps, act, slope):
        x = _c(x)
        ctx.save_for_backward(x, gamma, beta, run_mean, run_var)
        ctx.cfg = (eps, act, slope)
        return N.batchnorm_eval(x, gamma, beta, run_mean, run_var, eps, act, slope)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, run_mean, run_var = ctx.saved_tensors
        eps, act, slope = ctx.cfg
        return N.batchnorm_eval(x, gamma, beta, run_mean, run_var, eps, act, slope, dy=_c(dy)), None, None, None, None, None, None, None


class Upsample2xFn(Function):
    @staticmethod
    def forward(ctx, x):
        return N.upsample2x_fwd(_c(x))

    @staticmethod
    def backward(ctx, dy):
        return N.upsample2x_bwd(_c(dy))


class BCEFn(Function):
    """F.binary_cross_entropy(p, t) (mean) -> 0-dim loss; differentiable w.r.t. p only"""

    @staticmethod
    def forward(ctx, p, t):
        p = _c(p); t = _c(t)
        ctx.save_for_backward(p, t)
        return N.bce_fwd(p, t)[0]

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        return N.bce_bwd(p, t, _c(g).reshape(1)), None


class GlobalAvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.xshape = tuple(x.shape)
        return N.gap_fwd(_c(x))

    @staticmethod
    def backward(ctx, dy):
        return N.gap_bwd(_c(dy), ctx.xshape)


# functional front-ends -------------------------------------------------------
def conv2d(x, w, b=None, stride=1, padding=0, act=ACT_NONE, slope=0.01, fuse=(None, False), row_period=0):
    return Conv2dFn.apply(x, w, b, stride, padding, act, slope, fuse, row_period)


def linear(x, w, b=None, act=ACT_NONE, defer_act_bwd=False):
    return LinearFn.apply(x, w, b, act, defer_act_bwd)


def adversarial_loss(groups, scale=1.0):
    """groups: [(logits [C,n], labels [1,n] | [C,n], weights [C] | None), ...] -> scale * sum of the weighted per-row mean BCEs"""
    k = len(groups)
    return AdvLossFn.apply(scale, k, *[g[0] for g in groups], *[g[1] for g in groups], *[g[2] for g in groups])


def cross_entropy(logits, targets, ignore_index=-100):
    return SoftmaxCEFn.apply(logits, targets, ignore_index)


def smooth_l1_sum(pred, mask, target, sigma=3.0, scale=1.0):
    return SmoothL1Fn.apply(pred, mask, target, sigma, scale)


def binary_cross_entropy(p, t):
    return BCEFn.apply(p, t)


def sigmoid(x):
    return ActFn.apply(x, N.ACT_MODE["sigmoid"], 0.0)


def tanh(x):
    return ActFn.apply(x, N.ACT_MODE["tanh"], 0.0)
