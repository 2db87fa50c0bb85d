"""dist_init / average_gradients / broadcast_params -- the API of utils/distributed_utils.py:9-53, re-designed for
8 MI355X on one xGMI node (backend 'nccl' IS RCCL on ROCm).

* average_gradients(model): the reference issues one blocking all-reduce per parameter tensor (99 per iteration,
  571 MB).  Here each model's gradients live in ONE flat fp32 bucket (scda_amd.flat.FlatParams); the call is a single
  all-reduce on that bucket.  Sum semantics are unchanged (losses are pre-divided by world_size by the caller).
  With async_op=True the collective runs on RCCL's stream and overlaps the next phase; call .wait() before the step.
* broadcast_params(model): one broadcast of the flat parameter bucket + one per buffer (BN running stats, int64
  counters), instead of one per state_dict entry.
* dist_init(port, backend): SLURM variables as in the reference, with a RANK/WORLD_SIZE/LOCAL_RANK (torchrun)
  fallback; always rendezvous on 127.0.0.1 unless MASTER_ADDR is set.
"""
import logging
import os

import torch
import torch.distributed as dist

logger = logging.getLogger('global')


def _flat_of(model):
    return getattr(model, "_scda_flat", None)


def average_gradients(model, async_op=False):
    """SUM-all-reduce every parameter gradient of `model` across ranks."""
    flat = _flat_of(model)
    if flat is not None:
        return dist.all_reduce(flat.grad, async_op=async_op)
    # un-flattened module: bucket on the fly (coalesced, still a single collective)
    grads = [p.grad.data for p in model.parameters() if p.requires_grad and p.grad is not None]
    if not grads:
        return None
    bucket = torch.cat([g.reshape(-1) for g in grads])
    work = dist.all_reduce(bucket, async_op=False)
    off = 0
    for g in grads:
        g.copy_(bucket[off:off + g.numel()].view_as(g))
        off += g.numel()
    return work


def broadcast_params(model):
    """rank 0's parameters and buffers to everyone"""
    flat = _flat_of(model)
    if flat is not None:
        dist.broadcast(flat.data, 0)
        for b in model.buffers():
            dist.broadcast(b, 0)
        return
    for p in model.state_dict().values():
        dist.broadcast(p, 0)


def dist_init(port, backend='nccl'):
    if 'SLURM_PROCID' in os.environ:
        rank = int(os.environ['SLURM_PROCID'])
        world = int(os.environ['SLURM_NTASKS'])
        local = rank % max(torch.cuda.device_count(), 1)
    else:
        rank = int(os.environ.get('RANK', 0))
        world = int(os.environ.get('WORLD_SIZE', 1))
        local = int(os.environ.get('LOCAL_RANK', rank % max(torch.cuda.device_count(), 1)))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(port))
    os.environ['WORLD_SIZE'] = str(world)
    os.environ['RANK'] = str(rank)
    if backend == 'nccl':
        torch.cuda.set_device(local)
        dist.init_process_group(backend='nccl', rank=rank, world_size=world)
    else:
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    return dist.get_rank(), dist.get_world_size()
