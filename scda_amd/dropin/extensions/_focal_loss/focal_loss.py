"""SigmoidFocalLossFunction / SoftmaxFocalLossFunction(gamma, alpha, num_classes)(preds, targets, weight_pos)
-- extensions/_focal_loss/focal_loss.py:7-142.  Callable instances (as the reference's old-style Functions were)
over static autograd.Functions; both return a 1-element tensor holding losses.sum()."""
import torch
from torch.autograd import Function

from scda_amd import native as N


def _check(preds, targets, num_classes):
    assert preds.size(0) == targets.size(0)
    assert preds.size(1) == num_classes
    assert preds.is_contiguous() and targets.is_contiguous()
    assert preds.is_cuda and targets.is_cuda


class _SigmoidFocal(Function):
    @staticmethod
    def forward(ctx, preds, targets, weight_pos, gamma, alpha, num_classes):
        targets = targets.int()
        losses = N.focal_sigmoid_fwd(preds, targets, weight_pos, gamma, alpha, num_classes)
        ctx.save_for_backward(preds, targets)
        ctx.cfg = (weight_pos, gamma, alpha, num_classes)
        return losses.sum().reshape(1)

    @staticmethod
    def backward(ctx, grad_output):
        preds, targets = ctx.saved_tensors
        g = N.focal_sigmoid_bwd(preds, targets, *ctx.cfg)
        return g * grad_output, None, None, None, None, None


class _SoftmaxFocal(Function):
    @staticmethod
    def forward(ctx, preds, targets, weight_pos, gamma, alpha, num_classes):
        targets = targets.int()
        losses, priors = N.focal_softmax_fwd(preds, targets, weight_pos, gamma, alpha, num_classes)
        ctx.save_for_backward(preds, targets, priors)
        ctx.cfg = (weight_pos, gamma, alpha, num_classes)
        return losses.sum().reshape(1)

    @staticmethod
    def backward(ctx, grad_output):
        preds, targets, priors = ctx.saved_tensors
        g = N.focal_softmax_bwd(preds, targets, priors, *ctx.cfg)
        return g * grad_output, None, None, None, None, None


class SigmoidFocalLossFunction(object):
    def __init__(self, gamma, alpha, num_classes):
        self.gamma, self.alpha, self.num_classes = gamma, alpha, num_classes

    def __call__(self, preds, targets, weight_pos):
        _check(preds, targets, self.num_classes)
        return _SigmoidFocal.apply(preds, targets, float(weight_pos[0]), self.gamma, self.alpha, self.num_classes)


class SoftmaxFocalLossFunction(object):
    def __init__(self, gamma, alpha, num_classes):
        self.gamma, self.alpha, self.num_classes = gamma, alpha, num_classes

    def __call__(self, preds, targets, weight_pos):
        _check(preds, targets, self.num_classes)
        return _SoftmaxFocal.apply(preds, targets, float(weight_pos[0]), self.gamma, self.alpha, self.num_classes)
