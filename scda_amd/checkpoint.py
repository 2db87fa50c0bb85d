"""Checkpoint / pre-training I/O for the SCDA trainer -- the contract of the reference's utils/load_helper.py:28-54 and of
the save in tools/faster_rcnn_train_val.py:397-408, plus what the reference leaves out.

  load_pretrain(model, path | state_dict)   ImageNet VGG16 (torchvision `vgg16-397923af.pth`: `features.N.*`,
      `classifier.{0,3}.*`) or a reference checkpoint (plain dict, or {'state_dict': ...} for *.tar; 'module.' prefixes
      stripped); non-strict like the reference: keys present on both sides with equal shapes are copied, at least one must be.
  save_checkpoint(trainer, path, epoch, ...) reference fields (epoch / arch / state_dict / best_recall / optimizer) so the
      file still loads in the reference, plus the three GAN nets and all four optimiser states, which the reference never
      saves (its resume restarts the discriminators and Adam moments from scratch; restore_from drops the optimizer,
      load_helper.py:52-53).
  restore(trainer, path)                     everything that is in the file; a reference-made file restores the detector only.

Loading copies INTO the existing parameter tensors: they are views of the flat fp32 buckets (scda_amd.flat) and stay so.
"""
import logging

import torch

logger = logging.getLogger('global')


def remove_prefix(state_dict, prefix='module.'):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state_dict.items()}


def _load_matching(model, state, what, allow_mismatch=False):
    """copy every entry of `state` whose key exists in the model.  A key present on both sides with a DIFFERENT shape is an
    error, as in the reference (`load_state_dict(strict=False)` raises on size mismatches, utils/load_helper.py:22), unless
    allow_mismatch=True (e.g. deliberately re-using a backbone under a head with another num_classes)."""
    own = model.state_dict()
    # first pass decides, second pass copies: a caller that catches the error keeps an untouched model (the parameters are views of
    # the flat buckets -- a half-loaded bucket cannot be told from a loaded one afterwards)
    used = [k for k, v in state.items() if k in own and tuple(own[k].shape) == tuple(v.shape)]
    skipped = [k for k, v in state.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
    if skipped and not allow_mismatch:
        raise ValueError('%s: shape mismatch for %s (pass allow_mismatch=True to skip them)' % (
            what, ', '.join('%s %s vs %s' % (k, tuple(state[k].shape), tuple(own[k].shape)) for k in skipped[:8])))
    assert used, 'load NONE from pretrained checkpoint'      # reference: utils/load_helper.py:17
    with torch.no_grad():
        for k in used:
            own[k].copy_(state[k].to(device=own[k].device, dtype=own[k].dtype))   # in place: flat-bucket views survive
    if skipped:
        logger.warning('%s: skipped %d keys with mismatching shapes: %s', what, len(skipped), skipped)
    missing = [k for k in own if k not in state]
    logger.info('%s: used keys:%d missing keys:%d unused checkpoint keys:%d shape mismatches:%d',
                what, len(used), len(missing), len(state) - len(used) - len(skipped), len(skipped))
    return used, missing


def _read(path_or_state):
    if isinstance(path_or_state, dict):
        return path_or_state
    return torch.load(path_or_state, map_location='cpu', weights_only=False)


def load_pretrain(model, path_or_state, allow_mismatch=False):
    """-> model (same object); see the module docstring"""
    d = _read(path_or_state)
    if isinstance(d.get('state_dict'), dict):
        d = d['state_dict']
    _load_matching(model, remove_prefix(d), 'load_pretrain', allow_mismatch)
    return model


def _adam_state(opt):
    g = opt.param_groups[0]
    return {'step': opt.step_count, 'exp_avg': opt.exp_avg.detach().cpu().clone(),
            'exp_avg_sq': opt.exp_avg_sq.detach().cpu().clone(), 'lr': g['lr'], 'initial_lr': g.get('initial_lr', g['lr']),
            'betas': tuple(g['betas']), 'eps': g['eps'], 'weight_decay': g['weight_decay']}


def _load_adam(opt, st):
    if st['exp_avg'].numel() != opt.exp_avg.numel():
        raise ValueError('optimizer state has %d elements, the bucket %d' % (st['exp_avg'].numel(), opt.exp_avg.numel()))
    g = opt.param_groups[0]
    for k in ('betas', 'eps', 'weight_decay'):       # a checkpoint made with other hyper-parameters must not load silently
        if k in st and tuple(map(float, st[k] if k == 'betas' else (st[k],))) != tuple(map(float, g[k] if k == 'betas' else (g[k],))):
            raise ValueError('optimizer %s differs: checkpoint %r, live optimiser %r' % (k, st[k], g[k]))
    opt.step_count = int(st['step'])
    opt.exp_avg.copy_(st['exp_avg'].to(opt.exp_avg.device))
    opt.exp_avg_sq.copy_(st['exp_avg_sq'].to(opt.exp_avg_sq.device))
    g['lr'] = st['lr']
    if 'initial_lr' in st:
        g['initial_lr'] = st['initial_lr']


def _cpu_state(m):
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


def save_checkpoint(trainer, path, epoch, best_recall=0.0, arch='vgg16_FasterRCNN'):
    ck = {'epoch': epoch, 'arch': arch, 'best_recall': best_recall,
          'state_dict': _cpu_state(trainer.model),                       # the reference's fields ...
          'optimizer': _adam_state(trainer.opt['det']),
          'gan': {'dec': _cpu_state(trainer.dec), 'dis': _cpu_state(trainer.dis),   # ... and what it leaves out
                  'dis_patch': _cpu_state(trainer.dis_patch)},
          'optimizers': {k: _adam_state(trainer.opt[k]) for k in ('dec', 'dis', 'dis_patch')}}
    torch.save(ck, path)
    return ck


def restore(trainer, path_or_state, allow_mismatch=False):
    """-> (epoch, best_recall, arch)"""
    ck = _read(path_or_state)
    _load_matching(trainer.model, remove_prefix(ck['state_dict']), 'restore detector', allow_mismatch)
    for name, module in (('dec', trainer.dec), ('dis', trainer.dis), ('dis_patch', trainer.dis_patch)):
        if 'gan' in ck and name in ck['gan']:
            _load_matching(module, ck['gan'][name], 'restore ' + name, allow_mismatch)
    if isinstance(ck.get('optimizer'), dict) and 'exp_avg' in ck['optimizer']:
        _load_adam(trainer.opt['det'], ck['optimizer'])
    for name, st in ck.get('optimizers', {}).items():
        _load_adam(trainer.opt[name], st)
    from . import native
    native.WEIGHT_EPOCH[0] += 1            # weights changed behind the pack cache's back
    for f in getattr(trainer, 'flat', {}).values():
        f.epoch += 1
    return ck.get('epoch', 0), ck.get('best_recall', 0.0), ck.get('arch')
