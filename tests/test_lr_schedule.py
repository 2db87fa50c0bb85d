"""FlatAdam is a torch.optim.Optimizer: the reference's per-iteration warm-up scheduler (utils/lr_helper.py, imported from the
reference checkout where present), this repository's restatement of it, and torch's MultiStepLR all drive it; state_dict()
round-trips.  CPU only (no kernel is launched: schedulers and state handling are host logic)."""
import importlib.util
import os

import pytest
import torch
import torch.nn as nn

REF_LR = "/root/reference/utils/lr_helper.py"


def small_flat():
    from scda_amd.flat import FlatAdam, FlatParams
    net = nn.Sequential(nn.Conv2d(3, 4, 3), nn.Linear(5, 2))
    flat = FlatParams(net)
    return net, flat, FlatAdam(flat, 1e-3, betas=(0.9, 0.999), weight_decay=1e-4)


def test_flat_adam_is_a_torch_optimizer_and_multisteplr_drives_it():
    from torch.optim.lr_scheduler import MultiStepLR
    net, flat, opt = small_flat()
    assert isinstance(opt, torch.optim.Optimizer)
    assert len(opt.param_groups) == 1 and opt.param_groups[0]['params'][0].data_ptr() == flat.data.data_ptr()
    sched = MultiStepLR(opt, milestones=[2, 4], gamma=0.1, last_epoch=-1)
    lrs = []
    for _ in range(5):
        sched.step()                       # the reference steps at the top of every epoch (faster_rcnn_train_val.py:380)
        lrs.append(opt.param_groups[0]['lr'])
    assert lrs == pytest.approx([1e-3, 1e-4, 1e-4, 1e-5, 1e-5])
    assert opt.lr == pytest.approx(1e-5)


@pytest.mark.parametrize("which", ["product", "reference"])
def test_iter_exponential_warmup(which):
    """(world_size * batch)**(1/(n-1)) per iteration: iteration k of n runs at base * gamma**(k-1); then initial_lr <- lr"""
    from scda_amd.lr_schedule import warmup_gamma
    if which == "reference":
        if not os.path.exists(REF_LR):
            pytest.skip("reference checkout not present")
        spec = importlib.util.spec_from_file_location("ref_lr_helper", REF_LR)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        Sched = mod.IterExponentialLR       # type-checks `isinstance(optimizer, Optimizer)` (lr_helper.py:6-8)
    else:
        from scda_amd.lr_schedule import IterExponentialLR as Sched
    net, flat, opt = small_flat()
    gamma = warmup_gamma(8, 1, 4)
    s = Sched(opt, gamma)
    assert opt.param_groups[0]['lr'] == pytest.approx(1e-3) and opt.param_groups[0]['initial_lr'] == pytest.approx(1e-3)
    seen = []
    for _ in range(4):
        s.step()
        seen.append(opt.param_groups[0]['lr'])
    assert seen == pytest.approx([1e-3, 2e-3, 4e-3, 8e-3])
    with pytest.raises(TypeError):
        Sched(object(), gamma)


def test_state_dict_round_trip_and_alias_checks():
    from scda_amd.flat import FlatAdam, FlatParams
    net, flat, opt = small_flat()
    opt.exp_avg.normal_(); opt.exp_avg_sq.uniform_(); opt.step_count = 5
    opt.param_groups[0]['lr'] = 3e-4
    sd = opt.state_dict()
    assert set(sd) == {'state', 'param_groups'} and set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
    net2 = nn.Sequential(nn.Conv2d(3, 4, 3), nn.Linear(5, 2))
    flat2 = FlatParams(net2)
    opt2 = FlatAdam(flat2, 1e-3, weight_decay=1e-4)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 5 and opt2.lr == pytest.approx(3e-4)
    assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    # Module.zero_grad() sets gradients to None; a later backward then lands in fresh tensors -- folded back into the bucket
    net.zero_grad()
    assert all(p.grad is None for p in net.parameters())
    for p in net.parameters():
        p.grad = torch.ones_like(p)
    flat.check_aliases()
    assert float(flat.grad.sum()) == sum(p.numel() for p in net.parameters())
    assert all(flat._inside(p.grad, flat.grad) for p in net.parameters())
    # a re-homed parameter cannot be repaired: loud error instead of a silent no-op update
    net[1].weight.data = net[1].weight.data.clone()
    with pytest.raises(RuntimeError):
        flat.check_aliases()
