"""Flat fp32 parameter / gradient buckets.

All parameters of one optimiser group (detector, decoder, discriminator, patch discriminator) are re-homed into ONE
contiguous fp32 buffer, and their gradients into a second one; every nn.Parameter becomes a view.  This gives
  * one fused Adam launch per model (scda_adam_hip) instead of ~40 small optimiser kernels,
  * one RCCL all-reduce per model on the flat gradient bucket (xGMI rings are per-link bound: few large
    messages, not 99 small ones),
  * 16-byte aligned segments so the optimiser kernel can use dwordx4 accesses.
"""
import torch

from . import native as N


class FlatParams:
    ALIGN = 4  # floats (16 bytes)

    def __init__(self, module):
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
        for p, sz in zip(params, sizes):
            n = p.numel()
            self.data[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.data[off:off + n].view_as(p.data)
            p.grad = self.grad[off:off + n].view_as(p.data)
            off += sz
        self.numel = total
        self.epoch = 0   # bumped by the optimiser after each step: invalidates packed-weight caches of THESE parameters only
        for p in params:
            p._scda_flat = self
        module._scda_flat = self
        # conv weights whose GEMM-ready copies are rebuilt in one launch after every optimiser step (native.conv2d_pack_all)
        self.conv_weights = [m.weight for m in module.modules()
                             if isinstance(m, torch.nn.Conv2d) and m.weight.requires_grad and m.weight.is_cuda
                             and tuple(m.weight.shape[2:]) in ((3, 3), (1, 1))]

    def zero_grad(self):
        self.grad.zero_()
        for p in self.params:  # re-attach views if something replaced .grad (e.g. zero_grad(set_to_none=True))
            if p.grad is None or p.grad.data_ptr() < self.grad.data_ptr() or \
                    p.grad.data_ptr() >= self.grad.data_ptr() + self.numel * 4:
                off = (p.data.data_ptr() - self.data.data_ptr()) // 4
                p.grad = self.grad[off:off + p.numel()].view_as(p.data)


class FlatAdam:
    """torch.optim.Adam(params, lr, betas, eps=1e-8, weight_decay) semantics on a FlatParams bucket: one kernel."""

    def __init__(self, flat, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.step_count = 0
        self.param_groups = [{"lr": lr, "initial_lr": lr}]  # so LR schedulers written against torch.optim can drive it

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self):
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        N.adam_step(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, lr, self.betas[0], self.betas[1],
                    self.eps, self.weight_decay, self.step_count)
        self.flat.epoch += 1   # the kernel wrote through raw pointers: packed-weight caches of this bucket are stale
        N.conv2d_pack_all(self.flat)   # ... and are rebuilt right here, all layers and both directions in one launch
