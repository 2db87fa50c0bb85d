"""Learning-rate schedule of the SCDA driver on the fused optimisers.

The reference drives four torch.optim.Adam instances with (tools/faster_rcnn_train_val.py:346-388, utils/lr_helper.py:33-49)
  * a per-ITERATION exponential warm-up during the first `warmup_epochs`: lr_i = base_lr * gamma**i with
    gamma = (world_size * batch_size) ** (1 / (warmup_iters - 1)), stepped at the top of every iteration (:511-514), so the
    last warm-up iteration runs at base_lr * world_size * batch_size; afterwards `initial_lr` is overwritten with that value;
  * torch's MultiStepLR(milestones=step_epochs, gamma=0.1), stepped at the top of every epoch (:380-383).
FlatAdam is a torch.optim.Optimizer, so the reference's own `IterExponentialLR` and torch's `MultiStepLR` drive it as they
are; this module holds the same warm-up rule for users of ScdaTrainer who do not have the reference tree on their path.
"""
from torch.optim import Optimizer


class IterExponentialLR:
    """lr = initial_lr * gamma ** last_iter, advanced once per step(); same state protocol as utils/lr_helper.py:6-49
    (initial_lr recorded in the param groups, constructor performs step 0)"""

    def __init__(self, optimizer, gamma, last_iter=-1):
        if not isinstance(optimizer, Optimizer):
            raise TypeError('{} is not an Optimizer'.format(type(optimizer).__name__))
        self.optimizer, self.gamma = optimizer, gamma
        if last_iter == -1:
            for group in optimizer.param_groups:
                group.setdefault('initial_lr', group['lr'])
        elif any('initial_lr' not in g for g in optimizer.param_groups):
            raise KeyError("param 'initial_lr' is not specified in param_groups when resuming an optimizer")
        self.base_lrs = [g['initial_lr'] for g in optimizer.param_groups]
        self.last_iter = last_iter
        self.step(last_iter + 1)
        self.last_iter = last_iter      # as the reference: construction leaves the counter where it was

    def get_lr(self):
        return [b * self.gamma ** self.last_iter for b in self.base_lrs]

    def step(self, iter=None):
        self.last_iter = self.last_iter + 1 if iter is None else iter
        for group, lr in zip(self.optimizer.param_groups, self.get_lr()):
            group['lr'] = lr


def warmup_gamma(world_size, batch_size, warmup_iters):
    """tools/faster_rcnn_train_val.py:352-355"""
    if warmup_iters <= 1:
        raise ValueError("warm-up needs more than one iteration")
    return float(world_size * batch_size) ** (1.0 / (warmup_iters - 1))
