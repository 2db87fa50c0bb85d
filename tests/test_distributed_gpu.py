"""Two data-parallel ranks of the REAL training step on one MI355X (gloo carries the collectives, both ranks compute on
cuda:0): the step's world_size > 1 plumbing -- initial broadcast, loss / world_size, the four asynchronous flat-bucket
all-reduces and their waits -- keeps the replicas bit-identical while they see different data.  (RCCL itself needs one GPU
per rank; the driver's multi-GPU run covers that.  What can go wrong in OUR code is covered here.)"""
import copy
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_functions import CFG
    from scda_amd.dropin.utils.distributed_utils import broadcast_params
    from scda_amd.train_step import ScdaTrainer
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    cfg = copy.deepcopy(CFG)
    cfg['shared']['gan_model_flag'] = 2
    H, W = 256, 512
    import model_common as mc
    from test_train_step_gpu import build_product
    torch.manual_seed(1)
    models = mc.seeded_models(build_product)       # the seeded weights of the parity tests (a sane RPN) ...
    if rank == 1:                                  # ... perturbed on rank 1: the broadcast has to undo that
        with torch.no_grad():
            for m in models:
                for p in m.parameters():
                    p.add_(0.01)
    tr = ScdaTrainer(cfg, dev, lr=1e-3, new_w=W, new_h=H, world_size=world, models=models)
    for m in (tr.model, tr.dec, tr.dis, tr.dis_patch):
        broadcast_params(m)
    g = torch.Generator().manual_seed(50 + rank)   # different data per rank
    np.random.seed(60 + rank)
    losses = []
    for it in range(2):
        src = torch.randn(1, 3, H, W, generator=g).clamp_(-1, 1).to(dev)
        tgt = torch.randn(1, 3, H, W, generator=g).clamp_(-1, 1).to(dev)
        x1, y1 = 20 + 40 * rank + 10 * it, 30 + 20 * rank
        gts = torch.tensor([[[x1, y1, x1 + 150., y1 + 120., 3.], [260., 60., 420., 210., 5.]]])
        out = tr.step(src, gts, torch.tensor([[H, W, 1.0]]), tgt)
        losses.append(float(out['loss']))
    torch.cuda.synchronize()
    sums = {k: [float(f.data.double().sum()), float(f.data.double().abs().sum())] for k, f in tr.flat.items()}
    bn = float(tr.dis_patch.state_dict()['model_A_patch.0.model.1.running_mean'].double().sum())
    span = tr._det_early_span()
    torch.save({'sums': sums, 'losses': losses, 'bn': bn, 'early_reduces': tr.early_reduces,
                'early_mb': 0.0 if span is None else (span[1] - span[0]) * 4 / 1e6, 'bucket_mb': tr.flat['det'].numel * 4 / 1e6},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_stay_identical(cuda, tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(str(tmp_path / ("rank%d.pt" % r))) for r in range(world))
    assert r0['sums'] == r1['sums'], (r0['sums'], r1['sums'])          # parameters bit-identical after two steps
    assert r0['losses'] != r1['losses']                                  # ... although the ranks saw different data
    assert all(np.isfinite(v) for v in r0['losses'] + r1['losses'])
    assert r0['bn'] != r1['bn']      # BN running statistics of the patch discriminator stay per-rank, as in the reference
    # the detector's reduction left in two pieces, the classifier + heads (most of the bucket) from inside each backward
    assert r0['early_reduces'] == r1['early_reduces'] == 2 and r0['early_mb'] > 0.8 * r0['bucket_mb'], r0


def test_segmented_detector_reduce_equals_single_collective(cuda, tmp_path, monkeypatch):
    """the detector's all-reduce started inside the backward (classifier + heads behind FC6's weight gradient, the conv body at the
    end: distributed_utils.SegmentedReduce) leaves every parameter bucket bit-identical to ONE collective after the backward"""
    world = 2
    res = {}
    for mode in ("1", "0"):
        d = tmp_path / ("seg" + mode)
        d.mkdir()
        monkeypatch.setenv("SCDA_SEGMENTED_REDUCE", mode)
        mp.spawn(_worker, args=(world, _free_port(), str(d)), nprocs=world, join=True)
        res[mode] = [torch.load(str(d / ("rank%d.pt" % r))) for r in range(world)]
    assert res["1"][0]['early_reduces'] == 2 and res["0"][0]['early_reduces'] == 0
    for r in range(world):
        assert res["1"][r]['sums'] == res["0"][r]['sums'], (r, res["1"][r]['sums'], res["0"][r]['sums'])
        assert res["1"][r]['losses'] == res["0"][r]['losses']


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import model_common as mc
    from scda_amd import layers as L
    from scda_amd.dropin.utils.distributed_utils import broadcast_params
    from scda_amd.train_step import ScdaTrainer
    from test_train_step_gpu import build_product
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    H, W = 256, 512
    torch.manual_seed(1)
    tr = ScdaTrainer(mc.CFG, dev, lr=1e-3, new_w=W, new_h=H, world_size=world, models=mc.seeded_models(build_product))
    for m in (tr.model, tr.dec, tr.dis, tr.dis_patch):
        broadcast_params(m)
    tr.capture = True
    src, tgt, gts, info = mc.seeded_inputs(H, W, sample=rank)
    tape = torch.load(os.path.join(out_dir, "masks%d.pt" % rank))
    # the oracle's dropout masks, and an identical proposal ranking on both sides (scda_amd/probe.py: rpn_output), through the trainer
    import types
    rec = types.SimpleNamespace(records=torch.load(os.path.join(out_dir, "rpn%d.pt" % rank)))
    tr.probe = mc.Probe(dropout_masks=lambda shape, p, device: tape.pop(0).to(device),
                        rpn_output=mc.ReplaySource(rec, torch.device("cpu")).rpn)
    np.random.seed(mc.SEEDS['numpy'])
    out = tr.step(src.to(dev), gts, info, tgt.to(dev))
    torch.cuda.synchronize()
    assert not tape
    torch.save({'reduced': {n: {k: v.cpu() for k, v in g.items()} for n, g in tr.trace_reduced.items()},
                'local': {n: {k: v.cpu() for k, v in g.items()} for n, g in tr.trace.items()},
                'loss': float(out['loss'])}, os.path.join(out_dir, "grad%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_reduced_gradient_equals_oracle_sum_of_samples(cuda, tmp_path):
    """average_gradients is a SUM over ranks of gradients of loss / world_size (utils/distributed_utils.py:9-19, losses divided
    at tools/faster_rcnn_train_val.py:604,630,690,736).  Two ranks, two DIFFERENT samples: the gradient every rank holds after
    the all-reduce must equal the ORACLE's gradient of sample 0 plus the oracle's gradient of sample 1 (each computed on CPU
    with world_size = 2) -- not merely the other rank's copy.  Compared for the three phases whose gradient does not depend on
    an earlier reduced update: detector, image discriminators, patch discriminator.  Tolerance: relative L2 per tensor as in
    test_iteration_matches_oracle (each side breaks its own ReLU ties); a missing or doubled rank would be off by ~0.5."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import model_common as mc
    world, H, W = 2, 256, 512
    want = None
    for r in range(world):
        ref, _, masks = mc.oracle_iteration(H, W, lr=1e-3, record_masks=True, capture=True, sample=r, world_size=world,
                                            record_selections=True)
        torch.save(list(masks), str(tmp_path / ("masks%d.pt" % r)))
        torch.save([rec for rec in ref['_selections'].records if rec[0].startswith("rpn_")], str(tmp_path / ("rpn%d.pt" % r)))
        tr = {n: {k: v.clone() for k, v in g.items()} for n, g in ref['_trace'].items() if n in ('det', 'dis', 'dis_patch')}
        if want is None:
            want = tr
        else:
            for n in want:
                for k in want[n]:
                    want[n][k] += tr[n][k]
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(str(tmp_path / ("grad%d.pt" % r))) for r in range(world)]
    for n in want:
        errs = []
        scale = float(max(t.abs().max() for t in want[n].values()))
        for k, w in want[n].items():
            assert torch.equal(got[0]['reduced'][n][k], got[1]['reduced'][n][k]), (n, k)     # every rank holds the same sum
            assert not torch.equal(got[0]['local'][n][k], got[1]['local'][n][k]) or float(w.abs().max()) == 0
            if float(w.abs().max()) < 1e-6 * scale:
                continue
            errs.append(float((got[0]['reduced'][n][k].double() - w.double()).norm() / w.double().norm()))
        errs.sort()
        assert errs[len(errs) // 2] < 5e-3 and errs[-1] < 5e-2, (n, errs[len(errs) // 2], errs[-1])


def _rccl_worker(rank, port, out_dir):
    """ONE rank, backend 'nccl' (= RCCL): everything the N-GPU run does except moving bytes over xGMI"""
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("SLURM_PROCID", None)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import model_common as mc
    from scda_amd.dropin.utils.distributed_utils import average_gradients, broadcast_params, dist_init
    from scda_amd.train_step import ScdaTrainer
    from test_train_step_gpu import build_product
    r, w = dist_init(port, backend='nccl')                 # utils/distributed_utils.py:21-53, the 'nccl' branch
    assert (r, w) == (0, 1) and dist.get_backend() == "nccl"
    dev = torch.device("cuda", 0)
    H, W = 256, 512
    res = {}
    for name, coll in (("rccl", True), ("plain", False)):
        torch.manual_seed(1)
        tr = ScdaTrainer(mc.CFG, dev, lr=1e-3, new_w=W, new_h=H, world_size=1, models=mc.seeded_models(build_product),
                         collectives=coll)
        if coll:
            for m in (tr.model, tr.dec, tr.dis, tr.dis_patch):
                broadcast_params(m)                        # one RCCL broadcast per flat bucket + buffers
        np.random.seed(7)
        torch.manual_seed(11)
        losses = []
        for it in range(3):
            src, tgt, gts, info = mc.seeded_inputs(H, W, sample=it % 2)
            losses.append(float(tr.step(src.to(dev), gts, info, tgt.to(dev))['loss']))
        torch.cuda.synchronize()
        res[name] = {'losses': losses,
                     'sums': {k: [float(f.data.double().sum()), float(f.data.double().abs().sum())] for k, f in tr.flat.items()}}
        if coll:     # the reference driver's call shape on an un-flattened module: flattened on first use, then ONE all-reduce
            lin = torch.nn.Linear(8, 4).to(dev)
            lin(torch.ones(2, 8, device=dev)).sum().backward()
            g0 = lin.weight.grad.clone()
            wk = average_gradients(lin, async_op=True)
            wk.wait()
            assert torch.equal(lin.weight.grad, g0) and getattr(lin, "_scda_flat", None) is not None
    torch.save(res, os.path.join(out_dir, "rccl.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_one_rank_group_matches_plain_step(cuda, tmp_path):
    """The RCCL code path -- dist_init(..., 'nccl'), broadcast_params on the flat buckets, the four asynchronous
    dist.all_reduce(flat.grad, async_op=True) launches of a step, their waits and the hand-over between RCCL's stream and the
    compute / side streams -- executed on real RCCL with a one-rank group (the test box has one GPU; RCCL wants one device per
    rank).  A sum over one rank is the identity, so three steps must leave every parameter bucket BIT-identical to the same
    three steps without collectives; a missing wait or a wrong stream dependency shows up as a difference or a hang."""
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    res = torch.load(str(tmp_path / "rccl.pt"))
    assert res['rccl']['losses'] == res['plain']['losses'], res
    assert res['rccl']['sums'] == res['plain']['sums'], res
    assert all(np.isfinite(v) for v in res['rccl']['losses'])
