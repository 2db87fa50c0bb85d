"""Flat fp32 parameter / gradient buckets.

All parameters of one optimiser group (detector, decoder, discriminator, patch discriminator) are re-homed into ONE
contiguous fp32 buffer, and their gradients into a second one; every nn.Parameter becomes a view.  This gives
  * one fused Adam launch per model (scda_adam_hip) instead of ~40 small optimiser kernels,
  * one RCCL all-reduce per model on the flat gradient bucket (xGMI rings are per-link bound: few large
    messages, not 99 small ones),
  * 16-byte aligned segments so the optimiser kernel can use dwordx4 accesses.
"""
import os
import torch

from . import native as N


class FlatParams:
    ALIGN = 4  # floats (16 bytes)

    def __init__(self, module, keep_grads=False):
        """keep_grads: gradients that already exist (flattening a module AFTER a backward pass) are carried into the bucket"""
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        dev = params[0].device
        sizes = [(p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN for p in params]
        total = sum(sizes)
        self.data = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.params = params
        off = 0
        self.offset = {}     # id(parameter) -> first element of its slice in both buckets
        for p, sz in zip(params, sizes):
            n = p.numel()
            self.offset[id(p)] = off
            self.data[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.data[off:off + n].view_as(p.data)
            if keep_grads and p.grad is not None:
                self.grad[off:off + n].copy_(p.grad.reshape(-1))
            p.grad = self.grad[off:off + n].view_as(p.data)
            off += sz
        self.numel = total
        # Large dense weights (FC6: 411 MB, FC7: 67 MB) are not zero-filled before a backward pass: their one weight-gradient GEMM
        # per phase OVERWRITES its slice instead of zero-fill + read-modify-write (-0.9 GB of HBM traffic per detector phase).
        # `fresh` holds those whose slice still contains last phase's values; whoever writes first takes the token
        # (autograd_ops.LinearFn), and finalize_grads() zero-fills what nobody wrote before anything reads the bucket.
        self.lazy = [p for p in params if p.dim() == 2 and p.numel() >= (8 << 20)]
        self.fresh = set()
        self.epoch = 0   # bumped by the optimiser after each step: invalidates packed-weight caches of THESE parameters only
        for p in params:
            p._scda_flat = self
        module._scda_flat = self
        # conv weights whose GEMM-ready copies are rebuilt in one launch after every optimiser step (native.conv2d_pack_all)
        self.conv_weights = [m.weight for m in module.modules()
                             if isinstance(m, torch.nn.Conv2d) and m.weight.requires_grad and m.weight.is_cuda
                             and tuple(m.weight.shape[2:]) in ((3, 3), (1, 1))]

    def _inside(self, t, bucket):
        return t is not None and t.device == bucket.device and \
            bucket.data_ptr() <= t.data_ptr() < bucket.data_ptr() + self.numel * 4

    def check_aliases(self, span=None):
        """Every parameter must still be a view of `data` and its gradient a view of `grad`.  A gradient that was set to None
        or replaced by a fresh tensor (Module.zero_grad(set_to_none=True), an optimiser other than FlatAdam) is copied into the
        bucket and re-attached; a parameter that was re-homed (`module.cpu()`, `.to()`, `.half()`) cannot be repaired here.
        span = (lo, hi): only the parameters that lie inside that slice of the bucket."""
        for p in self.params:
            if span is not None:
                off = (p.data.data_ptr() - self.data.data_ptr()) // 4
                if off < span[0] or off + p.numel() > span[1]:
                    continue
            if not self._inside(p.data, self.data):
                raise RuntimeError("a parameter of this FlatParams bucket no longer lives in it (module.to()/.cpu()/.half() after "
                                   "flattening?): re-create FlatParams, or copy state_dict() instead of moving the module")
            if not self._inside(p.grad, self.grad):
                off = (p.data.data_ptr() - self.data.data_ptr()) // 4
                view = self.grad[off:off + p.numel()].view_as(p.data)
                if p.grad is not None:     # a complete gradient that autograd materialised outside the bucket (the kernels only
                    view.copy_(p.grad)     # accumulate into the bucket while p.grad IS the view): it replaces the stale slice
                else:
                    view.zero_()
                self.fresh.discard(id(p))  # the slice has just been written as a whole: finalize_grads() must not zero-fill it
                p.grad = view

    def take_fresh(self, p):
        """True exactly once per phase for a lazily-zeroed parameter: the caller must then OVERWRITE p.grad, not accumulate"""
        if id(p) in self.fresh:
            self.fresh.discard(id(p))
            return True
        return False

    def finalize_grads(self, span=None):
        """before the bucket (or its element range `span`) is read as a whole (all-reduce, optimiser step, tests): zero what no
        backward kernel wrote"""
        if self.fresh:
            for p in self.lazy:
                if id(p) in self.fresh and (span is None or span[0] <= self.offset[id(p)] < span[1]):
                    p.grad.zero_()
                    self.fresh.discard(id(p))

    def span_of(self, params):
        """-> (first, end) element range of the bucket that holds exactly `params` (their slices must be adjacent: no other
        parameter in between), or None"""
        ids = {id(p) for p in params}
        if not ids or not ids <= set(self.offset):
            return None
        inside = [p for p in self.params if id(p) in ids]
        lo = min(self.offset[id(p)] for p in inside)
        hi = max(self.offset[id(p)] + (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN for p in inside)
        if any(lo <= self.offset[id(p)] < hi for p in self.params if id(p) not in ids):
            return None
        return lo, hi

    def zero_grad(self):
        if self.lazy:
            base, edges = self.grad.data_ptr(), [0]
            for p in self.lazy:
                off = (p.grad.data_ptr() - base) // 4 if self._inside(p.grad, self.grad) else None
                if off is None:
                    edges = None
                    break
                edges += [off, off + p.numel()]
            if edges is None:
                self.grad.zero_()
                self.fresh = set()
            else:
                edges.append(self.numel)
                for a, b in zip(edges[0::2], edges[1::2]):
                    if b > a:
                        self.grad[a:b].zero_()
                self.fresh = {id(p) for p in self.lazy}
        else:
            self.grad.zero_()
        for p in self.params:  # re-attach views if something replaced .grad (e.g. zero_grad(set_to_none=True))
            if not self._inside(p.grad, self.grad):
                off = (p.data.data_ptr() - self.data.data_ptr()) // 4
                p.grad = self.grad[off:off + p.numel()].view_as(p.data)


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps=1e-8, weight_decay) semantics (L2 added to the gradient, bias-corrected moments)
    on a FlatParams bucket, as ONE fused kernel per step (scda_adam_hip), and a real `torch.optim.Optimizer`:

    * one param group whose single "parameter" is the whole bucket (a Parameter aliasing `flat.data`, `.grad` = `flat.grad`), so
      `param_groups[0]['lr']` is what the kernel reads and every scheduler that type-checks its optimiser drives it: the
      reference's `_IterLRScheduler` / `IterExponentialLR` (utils/lr_helper.py:6-8,33-49) and torch's `MultiStepLR`
      (tools/faster_rcnn_train_val.py:346-375);
    * `state_dict()` / `load_state_dict()` are the base class's: state[bucket] = {step, exp_avg, exp_avg_sq}, the layout
      torch.optim.Adam uses for a single flat parameter.

    Difference from torch.optim.Adam that cannot show on the SCDA nets: torch skips a parameter whose `.grad` is None, the
    bucket is updated as a whole (a parameter that never receives a gradient still gets weight decay).  Every parameter of
    the four SCDA nets receives a gradient in its phase.

    `Module.zero_grad()` (set_to_none), `.to()` / `.cpu()` on the module re-home parameter tensors away from the bucket; step()
    checks the aliasing first (re-attaching gradients that were merely set to None, refusing to step on moved parameters)
    instead of silently updating a bucket nobody writes gradients into."""

    def __init__(self, flat, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FlatAdam: invalid hyper-parameters")
        self.flat = flat
        self.max_blocks = int(os.environ.get('SCDA_ADAM_BLOCKS', '0'))     # > 0: explicit cap of the Adam launch (scripts/time_adam.py)
        self.bucket = torch.nn.Parameter(flat.data, requires_grad=True)   # aliases flat.data (no copy)
        self.bucket.grad = flat.grad
        super().__init__([self.bucket], dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.state[self.bucket] = {"step": torch.zeros((), dtype=torch.float32),
                                   "exp_avg": torch.zeros_like(flat.data), "exp_avg_sq": torch.zeros_like(flat.data)}

    # convenience views used by the trainer / checkpoints / tests
    @property
    def exp_avg(self):
        return self.state[self.bucket]["exp_avg"]

    @property
    def exp_avg_sq(self):
        return self.state[self.bucket]["exp_avg_sq"]

    @property
    def step_count(self):
        return int(self.state[self.bucket]["step"])

    @step_count.setter
    def step_count(self, n):
        self.state[self.bucket]["step"] = torch.tensor(float(n), dtype=torch.float32)

    lr = property(lambda self: self.param_groups[0]["lr"])
    betas = property(lambda self: self.param_groups[0]["betas"])
    eps = property(lambda self: self.param_groups[0]["eps"])
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"])

    def zero_grad(self, set_to_none=False):
        """always zero-fills (the backward kernels ACCUMULATE into the bucket; None would detach the views)"""
        self.flat.zero_grad()
        self.bucket.grad = self.flat.grad

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        st = self.state[self.bucket]          # the base class re-creates the tensors: keep them on the bucket's device
        for k in ("exp_avg", "exp_avg_sq"):
            if st[k].device != self.flat.data.device or st[k].numel() != self.flat.numel:
                if st[k].numel() != self.flat.numel:
                    raise ValueError("optimizer state has %d elements, the bucket %d" % (st[k].numel(), self.flat.numel))
                st[k] = st[k].to(self.flat.data.device)
        st["step"] = torch.as_tensor(st["step"], dtype=torch.float32).cpu()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.flat.check_aliases()
        self.flat.finalize_grads()
        g = self.param_groups[0]
        st = self.state[self.bucket]
        n = int(st["step"]) + 1
        st["step"] = torch.tensor(float(n), dtype=torch.float32)
        N.adam_step(self.flat.data, self.flat.grad, st["exp_avg"], st["exp_avg_sq"], float(g["lr"]), g["betas"][0], g["betas"][1],
                    g["eps"], g["weight_decay"], n, max_blocks=self.max_blocks)
        self.flat.epoch += 1   # the kernel wrote through raw pointers: packed-weight caches of this bucket are stale
        N.conv2d_pack_all(self.flat)   # ... and are rebuilt right here, all layers and both directions in one launch
        return loss
