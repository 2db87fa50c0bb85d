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
    """SUM-all-reduce every parameter gradient of `model` across ranks (utils/distributed_utils.py:9-19): ONE collective on
    the model's flat gradient bucket, in place.

    A module that is not flattened yet -- the reference's own driver builds plain modules and torch.optim.Adam
    (tools/faster_rcnn_train_val.py:305-316) -- is flattened HERE, on its first call: parameters and the gradients just
    computed move into one bucket each (FlatParams(keep_grads=True); the Parameter objects, and therefore the optimiser's
    state, stay the same), and from then on this is the zero-copy path: no temporary, no pack / unpack kernels, and
    async_op=True really overlaps.  Gradients that the optimiser's zero_grad() set to None and autograd re-created outside the
    bucket are copied back into it first (FlatParams.check_aliases: one copy per tensor, no allocation)."""
    flat = _flat_of(model)
    if flat is None:
        from scda_amd.flat import FlatParams
        if not any(p.requires_grad and p.grad is not None for p in model.parameters()):
            return None
        flat = FlatParams(model, keep_grads=True)
    else:
        flat.check_aliases()
        flat.finalize_grads()
    return dist.all_reduce(flat.grad, async_op=async_op)


class SegmentedReduce:
    """The SUM all-reduce of one flat gradient bucket in pieces, so that it can START inside the backward pass that fills it
    (the 547 MB detector reduction overlapped with the VGG-body backward).  Parameters enter the bucket in
    module order, so the detector's classifier + heads (FC6 / FC7 / cls / loc: 480 of the 547 MB) are one contiguous slice whose
    gradients are final as soon as FC6's weight gradient has been enqueued -- `launch_early()` is called from a tensor hook on the
    RoI-pooled features (scda_amd.train_step), `launch_rest()` after backward() for what lies in front of / behind the slice, and
    `wait()` before the optimiser step.  Element-wise the result is the single collective's (a sum over ranks per element).
    `average_gradients(model)` -- the reference's one call -- stays the single collective."""

    def __init__(self, flat, early):
        self.flat, self.early, self.works, self.early_done = flat, early, [], False

    def launch_early(self):
        lo, hi = self.early
        # (a head gradient that autograd -- or a foreign optimiser's zero_grad(set_to_none=True) -- left OUTSIDE the bucket is moved
        #  into it first: launch_rest()'s check would otherwise overwrite the reduced slice with the local gradient afterwards)
        self.flat.check_aliases((lo, hi))
        self.flat.finalize_grads((lo, hi))
        self.works.append(dist.all_reduce(self.flat.grad[lo:hi], async_op=True))
        self.early_done = True

    def launch_rest(self):
        self.flat.check_aliases()
        self.flat.finalize_grads()
        lo, hi = self.early if self.early_done else (0, 0)     # (the hook did not fire: nothing differentiable reached the slice)
        for a, b in ((0, lo), (hi, self.flat.numel)):
            if b > a:
                self.works.append(dist.all_reduce(self.flat.grad[a:b], async_op=True))
        return self

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []


def broadcast_params(model):
    """rank 0's parameters and buffers to everyone"""
    from scda_amd.layers import flush_counters
    flush_counters(model)              # BatchNorm batch counters are kept on the host between reads of the buffer
    flat = _flat_of(model)
    if flat is not None:
        dist.broadcast(flat.data, 0)
        for b in model.buffers():
            dist.broadcast(b, 0)
        return
    for p in model.state_dict().values():
        dist.broadcast(p, 0)


def _first_slurm_host(nodelist):
    """'node[12-15,20],other3' -> 'node12' ; 'gpu-a,gpu-b' -> 'gpu-a'"""
    import re
    m = re.match(r'^([^,\[]+)(?:\[([^\]]+)\])?', nodelist.strip())
    if not m:
        return None
    if m.group(2) is None:
        return m.group(1)
    first = re.split(r'[,-]', m.group(2))[0]
    return m.group(1) + first


def dist_init(port, backend='nccl'):
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC (the only form the MI355X hosts' driver supports); harmless
    # when HIP is already initialised with it exported, as the launch environment does
    _world = int(os.environ.get('SLURM_NTASKS', os.environ.get('WORLD_SIZE', 1)))
    if backend == 'nccl' and _world > 1:
        # eight hardware queues for a rank's five streams (scda_amd.hostenv.data_parallel_env: 26.0 -> 18.3 ms per iteration).  The
        # reference's driver calls dist_init first thing in main() (tools/faster_rcnn_train_val.py:129), before anything touches the GPU
        from scda_amd.hostenv import data_parallel_env
        if torch.cuda.is_initialized() and 'GPU_MAX_HW_QUEUES' not in os.environ:
            logger.warning('dist_init: HIP is already initialised, GPU_MAX_HW_QUEUES=8 comes too late for this process '
                           '(export it, or call dist_init before the first torch.cuda call)')
        data_parallel_env(_world)
    if 'SLURM_PROCID' in os.environ:
        rank = int(os.environ['SLURM_PROCID'])
        world = int(os.environ['SLURM_NTASKS'])
        local = rank % max(torch.cuda.device_count(), 1)
    else:
        rank = int(os.environ.get('RANK', 0))
        world = int(os.environ.get('WORLD_SIZE', 1))
        local = int(os.environ.get('LOCAL_RANK', rank % max(torch.cuda.device_count(), 1)))
    if 'MASTER_ADDR' not in os.environ:
        # single node: loop-back.  Several SLURM nodes: the first host of the job's node list, as the reference derives it
        # (utils/distributed_utils.py:27-35); never let every node rendezvous with itself.
        nnodes = int(os.environ.get('SLURM_NNODES', os.environ.get('SLURM_JOB_NUM_NODES', 1)))
        if 'SLURM_PROCID' in os.environ and nnodes > 1:
            # rank 0 (SLURM_PROCID 0) runs on the FIRST host of the step's node list.  SLURM_LAUNCH_NODE_IPADDR is where `srun` was
            # typed -- a login node when the job is started as the reference's scripts do (`srun -p ...` from the front end,
            # examples/faster-rcnn/cityscapes/vgg/4cluster.sh:13) -- and nobody listens there.
            addr = _first_slurm_host(os.environ.get('SLURM_STEP_NODELIST') or os.environ.get('SLURM_NODELIST', ''))
            if not addr:
                raise RuntimeError('dist_init: %d SLURM nodes but neither MASTER_ADDR nor a usable SLURM_STEP_NODELIST / '
                                   'SLURM_NODELIST is set' % nnodes)
            os.environ['MASTER_ADDR'] = addr
        else:
            os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ.setdefault('MASTER_PORT', str(port))
    import torch.multiprocessing as mp
    if mp.get_start_method(allow_none=True) is None:   # as the reference (:22-23): DataLoader workers must not fork a process
        mp.set_start_method('spawn')                    # that has already initialised HIP
    os.environ['WORLD_SIZE'] = str(world)
    os.environ['RANK'] = str(rank)
    if backend == 'nccl':
        torch.cuda.set_device(local)
        dist.init_process_group(backend='nccl', rank=rank, world_size=world)
    else:
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    return dist.get_rank(), dist.get_world_size()
