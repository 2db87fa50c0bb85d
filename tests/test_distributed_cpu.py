"""Data-parallel plumbing (scda_amd/dropin/utils/distributed_utils.py + scda_amd/flat.py) with world_size 2 on the
gloo backend, CPU tensors: flat-bucket gradient SUM all-reduce (sync and async), fallback for un-flattened modules,
initial broadcast including buffers, env:// rendezvous fallback of dist_init."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(5, 3)
        self.bn = nn.BatchNorm1d(3)
        self.b = nn.Linear(3, 2, bias=False)


def _worker(rank, world, port, results):
    for k in ("SLURM_PROCID", "SLURM_NTASKS", "SLURM_NODELIST"):
        os.environ.pop(k, None)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1")
    from scda_amd.dropin.utils.distributed_utils import average_gradients, broadcast_params, dist_init
    from scda_amd.flat import FlatParams
    r, w = dist_init(str(port), backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)           # different weights per rank before the broadcast
    m = Tiny()
    flat = FlatParams(m)
    m.bn.running_mean.fill_(float(rank + 1))
    broadcast_params(m)
    ok_bcast = all(float((p - q).abs().max()) == 0 for p, q in zip(m.state_dict().values(), _rank0_state().values()))
    ok_bn = float(m.bn.running_mean[0]) == 1.0
    # flat-bucket all-reduce: every gradient element = rank+1  -> 1+2 = 3
    for p in m.parameters():
        p.grad.fill_(float(rank + 1))
    average_gradients(m)
    ok_sum = all(bool((p.grad == 3).all()) for p in m.parameters())
    assert all(p.grad.data_ptr() >= flat.grad.data_ptr() for p in m.parameters())   # still views of the bucket
    for p in m.parameters():
        p.grad.fill_(float(10 * (rank + 1)))
    work = average_gradients(m, async_op=True)
    work.wait()
    ok_async = all(bool((p.grad == 30).all()) for p in m.parameters())
    # un-flattened module: coalesced on the fly
    m2 = Tiny()
    for p in m2.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    opt2 = torch.optim.Adam(m2.parameters(), 1e-2)     # built BEFORE the lazy flattening, as the reference's driver does
    before = [p.detach().clone() for p in m2.parameters()]
    average_gradients(m2)
    ok_plain = all(bool((p.grad == 3).all()) for p in m2.parameters())
    # ... which happened in place: same Parameter objects, now views of one bucket; the second round is the zero-copy async path,
    # also after the optimiser's zero_grad() (set_to_none) made autograd re-create the gradients outside the bucket
    flat2 = m2._scda_flat
    ok_plain &= all(flat2._inside(p.data, flat2.data) and flat2._inside(p.grad, flat2.grad) for p in m2.parameters())
    opt2.step()
    ok_plain &= all(not torch.equal(p, b) for p, b in zip(m2.parameters(), before))   # torch's Adam still owns these tensors
    opt2.zero_grad()
    ok_plain &= all(p.grad is None for p in m2.parameters())
    for p in m2.parameters():
        p.grad = torch.full_like(p, float(5 * (rank + 1)))
    work = average_gradients(m2, async_op=True)
    work.wait()
    ok_plain &= all(bool((p.grad == 15).all()) and flat2._inside(p.grad, flat2.grad) for p in m2.parameters())
    # the reduction of one bucket in pieces, the first launched from a hook INSIDE the backward pass (SegmentedReduce): the
    # parameters behind the hooked activation are one contiguous slice; same sums as the single collective on a twin
    from scda_amd.dropin.utils.distributed_utils import SegmentedReduce
    torch.manual_seed(7)
    m3, m4 = Tiny(), Tiny()
    m4.load_state_dict(m3.state_dict())
    f3, f4 = FlatParams(m3), FlatParams(m4)
    span = f3.span_of(list(m3.bn.parameters()) + list(m3.b.parameters()))
    ok_seg = span is not None and f3.span_of(list(m3.a.parameters()) + list(m3.b.parameters())) is None   # (a, b) are not adjacent
    seg = SegmentedReduce(f3, span)
    xin = torch.randn(6, 5, generator=torch.Generator().manual_seed(20 + rank))
    for mm, early in ((m3, seg), (m4, None)):
        hcur = mm.a(xin)
        if early is not None:
            state = {}
            def hook(g, early=early, state=state):
                state['head'] = [p.grad.clone() for p in list(m3.bn.parameters()) + list(m3.b.parameters())]
                early.launch_early()
            hcur.register_hook(hook)
        mm.b(mm.bn(hcur)).square().sum().backward()
    ok_seg &= all(bool(g.abs().sum() > 0) for g in state['head'])       # the slice's gradients existed when the hook fired
    seg.launch_rest().wait()
    average_gradients(m4)
    ok_seg &= torch.equal(f3.grad, f4.grad) and len(seg.works) == 0
    results[rank] = (ok_bcast, ok_bn, ok_sum, ok_async, ok_plain, ok_seg)
    dist.destroy_process_group()


def _rank0_state():
    torch.manual_seed(100)
    m = Tiny()
    m.bn.running_mean.fill_(1.0)
    return m.state_dict()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        results = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, results)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert dict(results) == {0: (True,) * 6, 1: (True,) * 6}, dict(results)


def test_flat_params_views_and_adam_bucket_alignment():
    from scda_amd.flat import FlatParams
    m = Tiny()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = FlatParams(m)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])                      # values preserved
    for p in m.parameters():
        assert p.data_ptr() % 16 == flat.data.data_ptr() % 16  # 16-byte aligned segments
        assert p.grad is not None and p.grad.shape == p.shape
    flat.grad.fill_(2.0)
    assert all(bool((p.grad == 2).all()) for p in m.parameters())
    flat.zero_grad()
    assert float(flat.grad.abs().sum()) == 0


def test_slurm_master_address_is_not_loopback_on_several_nodes(monkeypatch):
    from scda_amd.dropin.utils import distributed_utils as D
    assert D._first_slurm_host("node[12-15,20],other3") == "node12"
    assert D._first_slurm_host("gpu-a,gpu-b") == "gpu-a"
    assert D._first_slurm_host("SH-IDC1-10-5-30-[36-37]") == "SH-IDC1-10-5-30-36"
    seen = {}
    monkeypatch.setattr(D.dist, "init_process_group", lambda **kw: seen.update(kw))
    monkeypatch.setattr(D.dist, "get_rank", lambda: 3)
    monkeypatch.setattr(D.dist, "get_world_size", lambda: 16)
    for k, v in dict(SLURM_PROCID="3", SLURM_NTASKS="16", SLURM_NNODES="2", SLURM_NODELIST="cn[07-08]").items():
        monkeypatch.setenv(k, v)
    for k in ("MASTER_ADDR", "MASTER_PORT", "SLURM_LAUNCH_NODE_IPADDR"):
        monkeypatch.delenv(k, raising=False)
    assert D.dist_init("23456", backend="gloo") == (3, 16)
    assert os.environ["MASTER_ADDR"] == "cn07" and os.environ["MASTER_PORT"] == "23456" and seen["world_size"] == 16
    # started as the reference's scripts do -- `srun` typed on a login node (4cluster.sh:13): SLURM_LAUNCH_NODE_IPADDR is THAT host, where
    # no rank runs.  The rendezvous must go to the first host of the step (where SLURM_PROCID 0 is), as the reference derives it.
    monkeypatch.delenv("MASTER_ADDR")
    monkeypatch.setenv("SLURM_LAUNCH_NODE_IPADDR", "10.0.0.250")
    assert D.dist_init("23456", backend="gloo") == (3, 16) and os.environ["MASTER_ADDR"] == "cn07"
    monkeypatch.delenv("MASTER_ADDR")
    monkeypatch.setenv("SLURM_STEP_NODELIST", "cn08")           # a step on a subset of the allocation: its own list wins
    assert D.dist_init("23456", backend="gloo") == (3, 16) and os.environ["MASTER_ADDR"] == "cn08"
    monkeypatch.delenv("MASTER_ADDR")
    monkeypatch.delenv("SLURM_STEP_NODELIST")
    monkeypatch.setenv("SLURM_NODELIST", "")
    with pytest.raises(RuntimeError):
        D.dist_init("23456", backend="gloo")


def test_data_parallel_env_sets_only_defaults(monkeypatch):
    """scda_amd.hostenv.data_parallel_env: eight hardware queues for a rank of a multi-rank run; nothing for one rank; what the
    caller exported wins.  wants_blocking_sync: blocking waits only under a tight CPU quota, or when forced"""
    from scda_amd import hostenv
    for k in ("GPU_MAX_HW_QUEUES", "SCDA_BLOCKING_SYNC"):
        monkeypatch.delenv(k, raising=False)
    assert hostenv.data_parallel_env(1) == {} and "GPU_MAX_HW_QUEUES" not in os.environ
    try:
        assert hostenv.data_parallel_env(8) == {"GPU_MAX_HW_QUEUES": "8"}
        assert os.environ["GPU_MAX_HW_QUEUES"] == "8"
    finally:
        os.environ.pop("GPU_MAX_HW_QUEUES", None)        # (set directly in os.environ by the call: monkeypatch does not know it)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "16")
    assert hostenv.data_parallel_env(8) == {} and os.environ["GPU_MAX_HW_QUEUES"] == "16"
    monkeypatch.setattr(hostenv, "cpu_quota", lambda: 16)
    assert hostenv.wants_blocking_sync(8) and not hostenv.wants_blocking_sync(4) and not hostenv.wants_blocking_sync(1)
    monkeypatch.setattr(hostenv, "cpu_quota", lambda: None)
    assert not hostenv.wants_blocking_sync(8)
    monkeypatch.setenv("SCDA_BLOCKING_SYNC", "1")
    assert hostenv.wants_blocking_sync(1)
    monkeypatch.setenv("SCDA_BLOCKING_SYNC", "0")
    monkeypatch.setattr(hostenv, "cpu_quota", lambda: 2)
    assert not hostenv.wants_blocking_sync(8)
