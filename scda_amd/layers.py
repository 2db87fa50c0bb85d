"""nn.Module layers whose forward/backward run on the HIP kernels.

They subclass the torch.nn classes the reference uses so that parameter names,
shapes, `state_dict()` keys and initialisation are identical (checkpoints and
torchvision's vgg16-397923af.pth load unchanged), but none of them calls a
torch compute kernel: forward dispatches to scda_amd.autograd_ops.
"""

import os
import weakref

import torch
import torch.nn as nn

from . import autograd_ops as A
from . import native as N
from . import probe as P


class Conv2d(nn.Conv2d):
    """nn.Conv2d with an optional activation fused into the MFMA kernel's epilogue."""

    def __init__(self, *args, fused_act=A.ACT_NONE, slope=0.01, **kw):
        super().__init__(*args, **kw)
        if self.groups != 1 or self.dilation != (1, 1) or self.padding_mode != "zeros":
            raise NotImplementedError("scda_amd.Conv2d: groups/dilation/padding_mode are not part of the SCDA path")
        if self.stride[0] != self.stride[1] or self.padding[0] != self.padding[1]:
            raise NotImplementedError("scda_amd.Conv2d: anisotropic stride/padding")
        self.fused_act = fused_act
        self.slope = slope
        # static fusion plan (plan_act_fusion): act of the layer that feeds this conv / whether our consumer applies our act'
        self.input_act = None
        self.defer_act_bwd = False
        # > 0: the input is a vertical stack of independent maps of that many rows (the channel-major RoI-head layout
        # [1, C, R*7, 7] of the ResNet-C4 detector); set per call by the owner (dropin/models/mask_rcnn/resnet.py)
        self.row_period = 0
        # fusion plan: a MaxPool2x2 consumes this conv's ReLU output (plan_act_fusion): the pool can run in this conv's epilogue.
        # The hand-over is structural: the conv tells ITS pool module (a weak reference set by the plan) that the tensor it is about
        # to receive is already pooled, with the shape to expect -- not an attribute on the tensor, which a hook or wrapper that
        # returns a new tensor (detach, clone, checkpointing) would drop, pooling a second time without a word.
        self.pool_next = False
        self._pool_ref = None

    def _my_pool(self):
        """the MaxPool2x2 the fusion plan paired with THIS module object, or None.  The pairing is checked from both sides: a shallow
        copy of the model (nn.DataParallel replicas copy __dict__) carries the original's weak reference, but the original's pool names
        the original conv as its producer -- the copy then simply runs un-fused instead of signalling a pool it does not feed."""
        pool = self._pool_ref() if self._pool_ref is not None else None
        if pool is None or pool._producer is None or pool._producer() is not self:
            return None
        return pool

    def __getstate__(self):
        # copy.deepcopy / pickle / torch.save(model): weak references do not travel.  The copy runs un-fused until plan_act_fusion is
        # called on it (the model constructors do that; a deep copy of a planned model can call it again).
        state = dict(self.__dict__)
        state["_pool_ref"] = None
        state["pool_next"] = False
        return state

    def forward(self, x):
        pool = self._my_pool() if self.pool_next else None
        if pool is not None:
            pool._pooled_shape = None        # (an announcement left behind by a call that raised between this conv and its pool)
        if pool is not None and P.replay() is None and self.fused_act == A.ACT_RELU and self.defer_act_bwd:
            B, Cin, IH, IW = x.shape
            if x.is_cuda and N.conv_pool_fusable(B, Cin, IH, IW, self.out_channels, self.kernel_size[0], self.kernel_size[1],
                                                 self.stride[0], self.padding[0], self.row_period):
                # conv + ReLU + the 2x2 max-pool behind it in one launch; the pool module recognises the pooled tensor and passes it on
                y = A.ConvPoolFn.apply(x, self.weight, self.bias, self.slope, self.input_act)
                pool.expect_pooled(tuple(y.shape))
                return y
        return A.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.fused_act, self.slope,
                        (self.input_act, self.defer_act_bwd), self.row_period)

    def extra_repr(self):
        s = super().extra_repr()
        return s + (f", fused_act={self.fused_act}" if self.fused_act else "")


class ConvTranspose1x1(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(kernel_size=1, stride=1): a 1x1 conv with the [in,out,1,1] weight read transposed."""

    def __init__(self, cin, cout, **kw):
        super().__init__(cin, cout, kernel_size=1, stride=1, padding=0, **kw)

    def forward(self, x):
        w = self.weight.transpose(0, 1).contiguous()  # [out,in,1,1]; 96 floats -- autograd handles the permutation
        return A.conv2d(x, w, self.bias, 1, 0)


class ConvTranspose2x2s2(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(kernel_size=2, stride=2, padding=0) -- the up-sampling layer of the mask / keypoint branches
    (models/mask_rcnn/resnet.py:183-185).  Stride = kernel: every output pixel (2h+a, 2w+b) has exactly ONE tap,
    y[n, o, 2h+a, 2w+b] = sum_c x[n, c, h, w] W[c, o, a, b] + bias[o] -- a 1x1 convolution to 4*out channels on the MFMA kernel
    (channel o*4 + a*2 + b), then a pixel shuffle.  A ReLU behind it commutes with the shuffle and is fused into the conv."""

    def __init__(self, cin, cout, fused_act=A.ACT_NONE, slope=0.01, **kw):
        super().__init__(cin, cout, kernel_size=2, stride=2, padding=0, **kw)
        self.fused_act, self.slope = fused_act, slope

    def forward(self, x):
        cin, cout = self.weight.shape[0], self.weight.shape[1]
        w = self.weight.permute(1, 2, 3, 0).reshape(cout * 4, cin, 1, 1).contiguous()      # [o*4 + a*2 + b, c]
        b = self.bias.repeat_interleave(4) if self.bias is not None else None
        y = A.conv2d(x, w, b, 1, 0, self.fused_act, self.slope)                              # [n, 4*out, h, w]
        n, _, h, wd = y.shape
        return y.view(n, cout, 2, 2, h, wd).permute(0, 1, 4, 2, 5, 3).reshape(n, cout, 2 * h, 2 * wd)


class FusedAct(nn.Module):
    """Placeholder that keeps nn.Sequential indices where the reference has nn.ReLU(inplace=True) /
    nn.LeakyReLU(inplace=True) directly after a conv/linear/norm whose kernel already applied it."""

    def __init__(self, name="ReLU"):
        super().__init__()
        self.name = name

    def forward(self, x):
        return x

    def extra_repr(self):
        return f"{self.name} (fused into the producer's epilogue)"


class Linear(nn.Linear):
    def __init__(self, *args, fused_act=A.ACT_NONE, **kw):
        super().__init__(*args, **kw)
        self.fused_act = fused_act
        self.defer_act_bwd = False     # fusion plan: the Dropout behind this layer applies its ReLU gradient

    def forward(self, x):
        return A.linear(x, self.weight, self.bias, self.fused_act, self.defer_act_bwd)


class MaxPool2x2(nn.Module):
    relu_input = False             # fusion plan: this pool's backward also applies the ReLU gradient of the conv in front of it
    _pooled_shape = None           # set by the conv in front (Conv2d.forward) when IT pooled: the shape of the tensor to pass through
    _producer = None               # weak reference to that conv (plan_act_fusion): see Conv2d._my_pool

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_producer"] = None
        state["_pooled_shape"] = None
        return state

    def expect_pooled(self, shape):
        self._pooled_shape = shape

    def forward(self, x):
        if self._pooled_shape is not None:     # the producing conv already pooled (layers.Conv2d.forward / A.ConvPoolFn)
            want, self._pooled_shape = self._pooled_shape, None
            if tuple(x.shape) != want:
                raise RuntimeError("MaxPool2x2: the convolution in front pooled in its epilogue and announced %s, but a tensor of shape %s "
                                   "arrived (a hook between the two modules?)" % (want, tuple(x.shape)))
            return x
        return A.MaxPool2x2Fn.apply(x, self.relu_input)

    def extra_repr(self):
        return "kernel_size=2, stride=2"


class MaxPool3x3s2(nn.Module):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1), forward only"""

    def forward(self, x):
        return A.MaxPool3x3s2Fn.apply(x.detach())

    def extra_repr(self):
        return "kernel_size=3, stride=2, padding=1"


class Activation(nn.Module):
    def __init__(self, mode, slope=0.01):
        super().__init__()
        self.mode = N.ACT_MODE[mode]
        self.slope = slope

    def forward(self, x):
        return A.ActFn.apply(x, self.mode, self.slope)




class Dropout(nn.Module):
    """nn.Dropout(p).  The keep-mask comes from the library's counter-based generator; its 64-bit seed is drawn from
    torch's CPU generator, so runs are reproducible under torch.manual_seed() and resumable through torch.get_rng_state().
    Parity tests hand over masks through a Probe (scda_amd/probe.py: `dropout_masks`, callable(shape, p, device) -> uint8 mask) so that
    the CPU oracle and the device see the same Bernoulli draws."""

    relu_input = False  # fusion plan: the backward also applies the ReLU gradient of the Linear in front (eval mode: see forward)

    def __init__(self, p=0.5):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        fused = self.relu_input and P.replay() is None
        if not self.training or self.p == 0.0:
            # identity -- but a producer that deferred its ReLU gradient to us still needs it applied
            return A.ActFn.apply(x, N.ACT_MODE["relu"], 0.0) if fused and torch.is_grad_enabled() and x.requires_grad else x
        masks = P.dropout_masks()
        if masks is not None:
            mask = masks(tuple(x.shape), self.p, x.device)
            if fused:   # replayed masks (parity tests of the FUSED path): explicit mask, ReLU gradient through a no-op ReLU
                x = A.ActFn.apply(x, N.ACT_MODE["relu"], 0.0)
            return A.DropoutFn.apply(x, mask, 1.0 / (1.0 - self.p))
        from . import seeds
        if seeds.active is not None:
            # a hipGraph is being recorded: this launch takes its seed as a kernel ARGUMENT, which a recording would freeze -- every
            # replay would drop the same elements.  (The residual blocks' fused tail reads its seed from device memory; the trainer
            # only records when every dropout of the region is one of those: ScdaTrainer._gan_graph_ok.)
            raise RuntimeError("scda_amd.layers.Dropout cannot be recorded into a hipGraph (host-side seed)")
        seed = int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64))
        return A.DropoutSeededFn.apply(x, self.p, seed, fused)

    def extra_repr(self):
        return f"p={self.p}"


def plan_act_fusion(*sequentials):
    """Static fusion plan for PURE CHAINS (nn.Sequential whose intermediate tensors have exactly one consumer -- the VGG
    feature extractor, the FC6/FC7 classifier, each image-discriminator branch): the gradient of a fused ReLU / LeakyReLU is
    applied by the layer that CONSUMES the activation instead of by an elementwise pass of its own:
        Conv(act) -> Conv            the second conv's data-gradient epilogue multiplies by act'(its input)
        Conv(ReLU) -> MaxPool2x2     the pool's backward zeroes windows whose maximum is not positive
        Linear(ReLU) -> Dropout      the dropout's backward also tests its input > 0
    Both ends get a flag, so a producer never skips its act' unless its consumer applies it.  Removes 30 of the 44
    elementwise act' launches of an iteration.  Parity tests that replay activation masks run the un-fused backward."""
    n = 0
    for seq in sequentials:
        leaves = [m for m in seq.modules() if not list(m.children()) and not isinstance(m, FusedAct)]
        for a, b in zip(leaves, leaves[1:]):
            if isinstance(a, Conv2d) and a.fused_act != A.ACT_NONE and isinstance(b, Conv2d):
                a.defer_act_bwd, b.input_act = True, (a.fused_act, a.slope)
            elif isinstance(a, Conv2d) and a.fused_act == A.ACT_RELU and isinstance(b, MaxPool2x2):
                a.defer_act_bwd, b.relu_input = True, True
                a.pool_next = True
                a._pool_ref = weakref.ref(b)
                b._producer = weakref.ref(a)
            elif isinstance(a, Linear) and a.fused_act == A.ACT_RELU and isinstance(b, Dropout):
                a.defer_act_bwd, b.relu_input = True, True
            else:
                continue
            n += 1
    return n


class InstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(C) (affine=False, track_running_stats=False) with optional fused activation."""

    def __init__(self, num_features, eps=1e-5, fused_act=A.ACT_NONE, slope=0.01):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.fused_act = fused_act
        self.slope = slope
        self._up_ref = None        # weak reference to the Upsample2x that consumes this norm's output and nothing else does (pair_norm_upsample)

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_up_ref"] = None     # weak references do not travel (see Conv2d.__getstate__): a copy runs un-fused until paired again
        return state

    def forward(self, x):
        if P.replay() is not None and self.fused_act != A.ACT_NONE:   # parity tests: activation un-fused so its mask can be replayed
            y = A.InstanceNormFn.apply(x, self.eps, A.ACT_NONE, self.slope)
            return A.ActFn.apply(y, 0 if self.fused_act == A.ACT_RELU else 1, self.slope)
        up = my_upsample(self)
        if up is not None and N.instnorm_up2_ok(x):
            # the norm AND the bilinear x2 behind it in one launch; the Upsample2x module recognises the up-sampled tensor and passes it on
            y2 = A.InstanceNormUpFn.apply(x, self.eps, self.fused_act, self.slope)
            up.expect_upsampled(tuple(y2.shape))
            return y2
        return A.InstanceNormFn.apply(x, self.eps, self.fused_act, self.slope)


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d; training-mode statistics + running-stat update in one kernel, optional fused LeakyReLU."""

    def __init__(self, num_features, fused_act=A.ACT_NONE, slope=0.01, **kw):
        super().__init__(num_features, **kw)
        self.fused_act = fused_act
        self.slope = slope

    def forward(self, x):
        if not self.training:   # validation of vgg16_bn, dis_patch.eval(): running statistics, nothing is updated
            if self.running_mean is None:
                raise NotImplementedError("scda_amd.BatchNorm2d: eval mode without running statistics")
            return A.BatchNormEvalFn.apply(x, self.weight.detach(), self.bias.detach(), self.running_mean, self.running_var,
                                           self.eps, self.fused_act, self.slope)
        if self.momentum is None:   # cumulative moving average: needs the counter on the device at every call; no SCDA net uses it
            raise NotImplementedError("scda_amd.BatchNorm2d: momentum=None (cumulative average) is not implemented")
        if self.num_batches_tracked is not None:
            self._nbt_pending = getattr(self, "_nbt_pending", 0) + 1    # counted on the host, written into the buffer when it is read
        if P.replay() is not None and self.fused_act != A.ACT_NONE:   # parity tests: see InstanceNorm2d
            y = A.BatchNormTrainFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                         self.momentum, A.ACT_NONE, self.slope)
            return A.ActFn.apply(y, 0 if self.fused_act == A.ACT_RELU else 1, self.slope)
        return A.BatchNormTrainFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                        self.momentum, self.fused_act, self.slope)


    def forward_add_relu(self, x, residual):
        """relu(self(x) + residual) -- the residual join of a ResNet block.  One kernel in training mode on the maps the plane form
        serves (batch 1); the batch norm followed by the join kernel otherwise (eval mode, other shapes, a fused activation of its
        own, the parity tests' replay hook: they replay the join's mask)."""
        if (self.training and self.fused_act == A.ACT_NONE and P.replay() is None and self.momentum is not None and x.dim() == 4
                and x.shape == residual.shape and N.batchnorm_add_relu_ok(x) and x.is_contiguous() and residual.is_contiguous()
                and N.aligned16(x, residual) and not os.environ.get("SCDA_BN_NO_JOIN")):
            if self.num_batches_tracked is not None:
                self._nbt_pending = getattr(self, "_nbt_pending", 0) + 1
            return A.BatchNormAddReluFn.apply(x, residual, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                              self.momentum)
        return A.AddReluFn.apply(self(x), residual)


def _flush_nbt(module):
    n = getattr(module, "_nbt_pending", 0)
    if n and module.num_batches_tracked is not None:
        module.num_batches_tracked.add_(n)
    module._nbt_pending = 0


def _bn_state_dict(self, *args, **kw):
    _flush_nbt(self)
    return nn.BatchNorm2d.state_dict(self, *args, **kw)


def _bn_save(self, destination, prefix, keep_vars):
    _flush_nbt(self)            # num_batches_tracked += 1 per forward would be one tiny device kernel per BN layer and call
    return nn.BatchNorm2d._save_to_state_dict(self, destination, prefix, keep_vars)


def _bn_load(self, *args, **kw):
    self._nbt_pending = 0       # the loaded counter replaces everything counted so far, pending increments included
    return nn.BatchNorm2d._load_from_state_dict(self, *args, **kw)


BatchNorm2d._save_to_state_dict = _bn_save
BatchNorm2d.state_dict = _bn_state_dict
BatchNorm2d._load_from_state_dict = _bn_load


def flush_counters(module):
    """write the host-side batch counters of every BatchNorm2d below `module` into their `num_batches_tracked` buffers: call before
    reading those buffers directly (state_dict() does it by itself; broadcast_params() calls this)"""
    for m in module.modules():
        if isinstance(m, BatchNorm2d):
            _flush_nbt(m)


class Upsample2x(nn.Module):
    """Interpolate(scale_factor=2, mode='bilinear', align_corners=True).  In the decoders its input is the output of an instance norm
    that nothing else reads (pair_norm_upsample): that norm's launch then writes the up-sampled map itself and announces it here --
    the hand-over is structural, as Conv2d -> MaxPool2x2's (a shape announced by the producer, checked on arrival)."""
    _upsampled_shape = None        # set by the producer when IT up-sampled: the shape of the tensor to pass through
    _producer = None               # weak reference to that producer (an InstanceNorm2d, or the INSResBlock whose fused tail holds the norm)

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_producer"] = None
        state["_upsampled_shape"] = None
        return state

    def expect_upsampled(self, shape):
        self._upsampled_shape = shape

    def forward(self, x):
        if self._upsampled_shape is not None:
            want, self._upsampled_shape = self._upsampled_shape, None
            if tuple(x.shape) != want:
                raise RuntimeError("Upsample2x: the instance norm in front up-sampled in its own launch and announced %s, but a tensor of "
                                   "shape %s arrived (a hook between the two modules?)" % (want, tuple(x.shape)))
            return x
        return A.Upsample2xFn.apply(x)


def my_upsample(producer):
    """the Upsample2x paired with THIS producer object (pair_norm_upsample), or None; checked from both sides like Conv2d._my_pool, and
    a stale announcement (a call that raised between producer and consumer) is cleared"""
    ref = getattr(producer, "_up_ref", None)
    up = ref() if ref is not None else None
    if up is None or up._producer is None or up._producer() is not producer or os.environ.get("SCDA_NO_NORM_UP_FUSION"):
        return None
    up._upsampled_shape = None
    return up


def pair_norm_upsample(producer, up):
    """declare that `up` (an Upsample2x) is the ONLY consumer of `producer`'s output (an InstanceNorm2d, or a module that runs one in a
    fused tail): the producer may then write the up-sampled map itself"""
    producer._up_ref = weakref.ref(up)
    up._producer = weakref.ref(producer)


class GlobalAvgPool(nn.Module):
    def forward(self, x):
        return A.GlobalAvgPoolFn.apply(x)
