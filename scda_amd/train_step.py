"""The SCDA training iteration (tools/faster_rcnn_train_val.py:507-750) on the MI355X.

One call of ScdaTrainer.step() = one source + one target image through the four optimiser phases of the
reference's train() loop body:
    (1) image discriminators   (:567-616)      (3) cluster decoders   (:642-704)
    (2) patch discriminator    (:623-635)      (4) detector           (:716-750)
The parameter updates are the reference's.  What is NOT replayed is work whose result the reference throws away
(SURVEY.md 3.1): gradients deposited into nets whose optimiser is not stepping in that phase are zeroed by the
owning phase's zero_grad() before use, so here those sub-graphs run under no_grad / on detached inputs.  The
numpy global RNG is consumed in exactly the reference's order (soft labels AND the "hard" labels, which the
reference also draws with np.random.uniform).

Each model's parameters and gradients live in one flat fp32 bucket (scda_amd.flat): one fused Adam kernel and
one RCCL all-reduce per phase.  With world_size > 1 the all-reduces of phases 1-3 are launched asynchronously and
waited for just before the corresponding optimiser step.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import autograd_ops as A
from . import native as N
from . import probe as P
from .flat import FlatAdam, FlatParams
from ._timing import mark
from .dropin.models.faster_rcnn.faster_rcnn_adver_expansion_reweight_cluster import (GAN_decoder_AE, GAN_dis_AE,
                                                                                     GAN_dis_AE_patch)
from .dropin.models.faster_rcnn.vgg_adver_expansion_cluster import vgg16
from .dropin.utils.distributed_utils import SegmentedReduce, average_gradients


def get_corner_from_center(center, recon_size, new_w, new_h):
    """recon_size x recon_size crop window around each cluster centre, pushed inside the new_w x new_h image;
    int() truncation as in tools/faster_rcnn_train_val.py:411-438 -> list of [x1, y1, x2, y2].
    recon_size may be (height, width) for a rectangular window (the ResNet configuration's 2048-d RoI features unfold to a
    32 x 64 map, reconstructed as 128 x 256); the reference only has the square case."""
    rh, rw = (recon_size, recon_size) if np.isscalar(recon_size) else recon_size
    half_w, half_h = rw // 2, rh // 2
    boxes = []
    for cx, cy in np.asarray(center)[:, :2]:
        x1 = max(int(cx) - half_w, 0)
        y1 = max(int(cy) - half_h, 0)
        if x1 == 0:
            x2 = rw
        else:
            x2 = min(int(cx) + half_w, new_w)
            if x2 == new_w:
                x1 = new_w - rw
        if y1 == 0:
            y2 = rh
        else:
            y2 = min(int(cy) + half_h, new_h)
            if y2 == new_h:
                y1 = new_h - rh
        boxes.append([x1, y1, x2, y2])
    return boxes


def builder_gan(cluster_num=4, threshold=128, recon_size=256, neww=64, newh=64):
    """hyper-parameters of the GAN part: tools/faster_rcnn_train_val.py:255-274"""
    size2layers = {256: 3, 512: 4, 128: 2}
    params_dec = {'ch': threshold, 'input_dim_a': 3, 'input_dim_b': 3, 'n_enc_front_blk': 5, 'n_enc_res_blk': 2,
                  'n_enc_shared_blk': 2, 'n_gen_shared_blk': 2, 'n_gen_res_blk': 3,
                  'n_gen_front_blk': size2layers[recon_size], 'res_dropout_ratio': 0.5, 'neww': neww, 'newh': newh,
                  'cluster_num': cluster_num, 'threshold': threshold}
    params_dis = {'input_dim_a': 3, 'input_dim_b': 3, 'ch': 32, 'n_gen_res_blk': 3, 'n_layer': size2layers[recon_size]}
    params_patch = {'n_in': threshold, 'n_out': threshold * 2, 'cluster_num': cluster_num, 'w': neww, 'h': newh}
    return GAN_dis_AE(params_dis), GAN_decoder_AE(params_dec), GAN_dis_AE_patch(params_patch)


def _soft(flag, shape, device):
    """generate_soft_label: U(0.8,1) for 1, U(0,0.3) for 0, drawn from numpy's global RNG (:440-448)"""
    lo, hi = (0.8, 1.0) if flag == 1 else (0.0, 0.3)
    return N.upload(np.random.uniform(lo, hi, size=tuple(shape)), device, torch.float32)


def _hard(flag, shape, device):
    """generate_hard_label: constant, but still one numpy draw per element (:450-458)"""
    v = 1.0 if flag == 1 else 0.0
    return N.upload(np.random.uniform(v, v, size=tuple(shape)), device, torch.float32)


def _labels(specs, device):
    """several label tensors in ONE host-to-device copy: specs = [(kind, flag, shape)], kind 's' (generate_soft_label) or 'h'
    (generate_hard_label), drawn in this order from numpy's global RNG exactly as the separate calls would (the reference's order of
    draws, :442-458).  Each upload is a tiny copy on the compute stream, and the GAN phases are a chain of tiny launches."""
    draws = []
    for kind, flag, shape in specs:
        if kind == 's':
            lo, hi = (0.8, 1.0) if flag == 1 else (0.0, 0.3)
        else:
            lo = hi = 1.0 if flag == 1 else 0.0
        draws.append(np.random.uniform(lo, hi, size=tuple(shape)).astype(np.float32))
    offs, total = [], 0
    for d in draws:
        offs.append(total)
        total += (d.size + 3) // 4 * 4              # 16-byte aligned slices
    host = np.zeros(total, dtype=np.float32)
    for d, o in zip(draws, offs):
        host[o:o + d.size] = d.ravel()
    flat = N.upload(host, device, torch.float32)
    return [flat[o:o + d.size].view(d.shape) for d, o in zip(draws, offs)]


def _crops(img, corners, recon):
    rh, rw = (recon, recon) if np.isscalar(recon) else recon
    out = []
    for x1, y1, x2, y2 in corners:
        assert x2 - x1 == rw and y2 - y1 == rh, "crop window does not match recon_size"
        out.append(img[:, :, y1:y2, x1:x2])
    return torch.cat(out, 0).contiguous()


class _Frozen:
    """context: parameters of `modules` temporarily do not require grad (their wgrad kernels are skipped)"""

    def __init__(self, *modules):
        self.params = [p for m in modules for p in m.parameters()]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *a):
        for p in self.params:
            p.requires_grad_(True)


class _GanGraphs:
    """hipGraphs of the three GAN regions of ScdaTrainer.step (A, B, C).  Two eager iterations first (planner caches, workspaces, packed
    weights, BN state), the third records each region -- `torch.cuda.graph` records the library's ctypes-launched kernels, the A / B
    fork-join on the branch stream and the autograd backward inside a region -- and replays it at once (recording does not execute);
    every later iteration copies the region's inputs into the recorded tensors, draws the dropout seeds of the region on the host in the
    eager modules' order (scda_amd/seeds.py) and replays.  All regions share one memory pool: B differentiates through the autograd graph
    region A's recording built, whose saved activations live in that pool."""
    KEYS = {'a': ('src_patch', 'tgt_patch', 'x_small', 't_small', 'score1', 'score0', 'score0p', 'score1p'),
            'b': ('x_small', 't_small', 'tgt_patch', 'one_t', 'zero_t', 'one_s', 'zero_s'),
            'c': ('src_patch', 'tgt_patch', 'ones_all', 'ones_row')}

    def __init__(self, trainer):
        from . import seeds
        self.tr, self.calls, self.reg, self.static = trainer, 0, {}, {}
        self.arena = seeds.SeedArena(trainer.device)
        self.pool = None

    def ready(self, name):
        return name in self.reg

    def recording(self):
        return self.calls >= 3

    def _bns(self):
        from . import layers as L
        return [m for m in self.tr.dis_patch.modules() if isinstance(m, L.BatchNorm2d)]

    def fits(self, t):
        """this iteration's tensors have the shapes the regions were recorded with (otherwise the iteration runs eagerly)"""
        return all(k not in self.static or self.static[k].shape == t[k].shape for k in ('src_patch', 'tgt_patch', 'x_small', 't_small'))

    def _load(self, name, t):
        """this iteration's inputs of region `name` into the recorded tensors -- each ONCE per iteration: B and C read what A loaded
        (and while B is being recorded, autograd still holds A's inputs as saved tensors: an in-place copy would invalidate them)"""
        if name == 'a':
            self._loaded = set()
        for k in self.KEYS[name]:
            if k in self._loaded:
                continue
            self._loaded.add(k)
            dst = self.static.get(k)
            if dst is None:
                self.static[k] = t[k].detach().clone()
            elif dst.data_ptr() != t[k].data_ptr():
                if dst.shape != t[k].shape:
                    raise RuntimeError("GAN hipGraph: input %s changed shape %s -> %s" % (k, tuple(dst.shape), tuple(t[k].shape)))
                dst.copy_(t[k], non_blocking=True)
        return {k: self.static[k] for k in self.KEYS[name]}

    def record(self, name, t, fn):
        """record region `name` (fn(static inputs) -> tuple of tensors / None) and run it once by replaying"""
        from . import seeds
        st = self._load(name, t)
        bns = self._bns()
        before = [getattr(m, "_nbt_pending", 0) for m in bns]
        lo = self.arena.n
        seeds.active = self.arena
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self.pool):
                out = fn(st)
        finally:
            seeds.active = None
        if self.pool is None:
            self.pool = g.pool()
        inc = [(m, getattr(m, "_nbt_pending", 0) - b0) for m, b0 in zip(bns, before)]
        for m, b0 in zip(bns, before):         # the recording launched nothing: undo its host-side counting
            m._nbt_pending = b0
        self.reg[name] = {"graph": g, "out": out, "bn_inc": inc, "seeds": (lo, self.arena.n)}
        return self._replay(name, redraw=False)

    def _replay(self, name, redraw=True):
        r = self.reg[name]
        lo, hi = r["seeds"]
        if redraw:
            self.arena.redraw(lo, hi)
        self.arena.upload(lo, hi)
        r["graph"].replay()
        for m, inc in r["bn_inc"]:
            m._nbt_pending = getattr(m, "_nbt_pending", 0) + inc
        return r["out"]

    def run(self, name, t):
        self._load(name, t)
        out = self._replay(name)
        if name == 'a':
            return out[0], out[1], out[2].clone(), out[3].clone()
        if name == 'b':
            return out[0].clone(), out[1].clone(), out[2]
        return out[0].clone(), out[1].clone()


class _recording:
    """`with _recording(graphs, name, t) as rec: outs = rec(fn)`: run fn eagerly, or -- from the third iteration of a trainer with
    hipGraphs enabled -- record it as region `name` and replay it"""

    def __init__(self, graphs, name, t):
        self.g, self.name, self.t = graphs, name, t

    def __enter__(self):
        def rec(fn):
            if self.g is not None and self.g.recording():
                out = self.g.record(self.name, self.t, fn)
                if self.name == 'a':
                    return out[0], out[1], out[2].clone(), out[3].clone(), None, None
                if self.name == 'b':
                    return out[0].clone(), out[1].clone(), out[2], None
                return out[0].clone(), out[1].clone()
            return fn(self.t)
        return rec

    def __exit__(self, *a):
        return False


class ScdaTrainer:
    def __init__(self, cfg, device, lr=1.25e-5, cluster_num=4, threshold=128, recon_size=256, new_w=1024, new_h=512,
                 weight_decay=1e-4, world_size=1, models=None, recon_hw=None, reference_style=False, collectives=None, probe=None):
        """recon_hw: (height, width) of the reconstructions / image crops when they are not recon_size x recon_size (a
        detector whose RoI feature does not unfold to a square map: see scda_amd/resnet_config.py).
        reference_style: run the iteration the way the reference's own driver would on top of the drop-in modules -- four
        `torch.optim.Adam` instances over plain parameter lists (tools/faster_rcnn_train_val.py:305-316), phases strictly in
        program order on one stream, detector backward last, `average_gradients(model)` per phase -- so that the cost of NOT
        using this repository's step (flat buckets + fused Adam, early backward, side streams) can be measured
        (scripts/bench_reference_style.py).
        collectives: issue the four per-phase all-reduces (default: world_size > 1).  True with a one-rank process group runs
        the whole RCCL path -- async all-reduce on the device buckets, waits, stream hand-over -- on a single GPU
        (tests/test_distributed_gpu.py::test_rccl_one_rank_group_matches_plain_step)."""
        # parity tests: a scda_amd.probe.Probe (replayed selections / RPN outputs / dropout masks of the CPU oracle), visible to the
        # product's modules while THIS trainer's step() runs and at no other time; None in production
        self.probe = probe
        self.cfg, self.device = cfg, device
        self.collectives = (world_size > 1) if collectives is None else bool(collectives)
        self.early_reduces = 0      # all-reduces launched from inside a detector backward (SegmentedReduce) so far
        if self.collectives and device.type == "cuda" and int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 6:
            import logging   # five streams on the runtime's default four hardware queues: 26 instead of 18.5 ms per iteration
            logging.getLogger('global').warning(
                "ScdaTrainer: data-parallel run with GPU_MAX_HW_QUEUES=%s; export GPU_MAX_HW_QUEUES=8 before the process touches the "
                "GPU (scda_amd.hostenv.data_parallel_env; utils.distributed_utils.dist_init sets it when called first)",
                os.environ.get("GPU_MAX_HW_QUEUES", "unset (4)"))
        from .hostenv import configure_host_threads
        configure_host_threads()
        self.cluster_num, self.threshold, self.recon = cluster_num, threshold, (recon_hw or recon_size)
        self.new_w, self.new_h, self.world_size = new_w, new_h, world_size
        if models is None:
            model = vgg16(pretrained=False, cfg=cfg['shared'])
            dis, dec, dis_patch = builder_gan(cluster_num, threshold, recon_size)
        else:
            model, dec, dis, dis_patch = models
        self.model, self.dec, self.dis, self.dis_patch = (m.to(device) for m in (model, dec, dis, dis_patch))
        for m in (self.model, self.dec, self.dis, self.dis_patch):
            m.train()
        named = (("det", self.model), ("dec", self.dec), ("dis", self.dis), ("dis_patch", self.dis_patch))
        if reference_style:
            self.flat = {}
            self.opt = {k: torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr, betas=(0.9, 0.999),
                                            weight_decay=weight_decay) for k, m in named}
        else:
            self.flat = {k: FlatParams(m) for k, m in named}
            self.opt = {k: FlatAdam(f, lr, betas=(0.9, 0.999), weight_decay=weight_decay) for k, f in self.flat.items()}
        self._warmup = None          # per-iteration schedulers while warming up (begin_warmup / end_warmup)
        self._epoch_sched = None     # per-epoch MultiStepLR (set_epoch_schedule / begin_epoch)
        self._in_graph = False   # a GAN region is being recorded / replayed: its all-reduces are issued by step(), behind it
        self.capture = False   # debugging / parity tests: keep a copy of each phase's gradients in self.trace
        self.trace = {}        # ... as computed by this rank, and in self.trace_reduced after the all-reduce (world_size > 1)
        self.trace_reduced = {}
        # scheduling: detector backward enqueued as soon as its losses exist; target branch on a high-priority side stream
        self.early_backward = not reference_style
        self.side = torch.cuda.Stream(device=device, priority=-1) if device.type == "cuda" else None
        if os.environ.get("SCDA_SIDE_STREAM", "1") == "0" or reference_style:
            self.side = None
        # the B halves of the decoder / image discriminator run beside their A halves (SCDA_AB_STREAMS=0: one stream)
        if device.type == "cuda" and os.environ.get("SCDA_AB_STREAMS", "1") != "0" and not reference_style:
            self.branch = torch.cuda.Stream(device=device)
            self.dec.branch_stream = self.branch
            self.dis.branch_stream = self.branch
        else:
            self.branch = None

    # ---- learning-rate schedule (tools/faster_rcnn_train_val.py:346-388) ------------------------------------------
    def begin_warmup(self, warmup_iters, batch_size=1, world_size=None):
        """exponential per-iteration warm-up of all four optimisers towards lr * world_size * batch_size (:346-364)"""
        from .lr_schedule import IterExponentialLR, warmup_gamma
        gamma = warmup_gamma(self.world_size if world_size is None else world_size, batch_size, warmup_iters)
        self._warmup = [IterExponentialLR(o, gamma) for o in self.opt.values()]
        return gamma

    def end_warmup(self):
        """:365-367: the magnified rate becomes the schedule's base rate"""
        self._warmup = None
        for o in self.opt.values():
            for g in o.param_groups:
                g['initial_lr'] = g['lr']

    def set_epoch_schedule(self, milestones, gamma=0.1, start_epoch=0):
        """torch MultiStepLR on each optimiser, as :370-376; advance it with begin_epoch() at the top of every epoch (:380-383)"""
        from torch.optim.lr_scheduler import MultiStepLR
        self._epoch_sched = [MultiStepLR(o, milestones=list(milestones), gamma=gamma, last_epoch=start_epoch - 1)
                             for o in self.opt.values()]

    def begin_epoch(self):
        import warnings
        with warnings.catch_warnings():   # the reference steps the schedule BEFORE the epoch's optimiser steps, on purpose
            warnings.simplefilter("ignore", UserWarning)
            for s in self._epoch_sched or ():
                s.step()
        return self.lr

    @property
    def lr(self):
        return self.opt['det'].param_groups[0]['lr']

    # ------------------------------------------------------------------
    def _join_branch(self):
        """The B halves' backward kernels accumulate straight into the flat gradient bucket on the branch stream; autograd
        sees no leaf gradient there (autograd_ops._sink hands it None), so ITS end-of-backward stream sync does not cover
        them: order the compute stream behind the branch stream before anything reads the bucket."""
        if self.branch is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.branch)

    def _grab_reduced(self, name, module):
        if self.capture:
            self.trace_reduced[name] = {k: p.grad.detach().clone() for k, p in module.named_parameters()}

    def _finish_grads(self, module):
        """device side of the end of a phase's backward (recordable into a hipGraph): the B halves joined, lazily-zeroed slices
        nobody wrote this phase filled (FlatParams.lazy)"""
        self._join_branch()
        flat = getattr(module, "_scda_flat", None)
        if flat is not None:
            flat.finalize_grads()

    def _all_reduce(self, module, async_op):
        """host side: gradient capture (tests) and the phase's all-reduce -- never inside a recorded region"""
        if self.capture:
            name = {id(self.model): 'det', id(self.dec): 'dec', id(self.dis): 'dis', id(self.dis_patch): 'dis_patch'}[id(module)]
            self.trace[name] = {k: p.grad.detach().clone() for k, p in module.named_parameters()}
        if self.collectives:
            return average_gradients(module, async_op=async_op)
        return None

    def _det_early_span(self):
        """element range of the detector bucket whose gradients are final behind the RoI head's backward (the model names the
        parameters: VGG.EARLY_REDUCE_PREFIXES), or None (SCDA_SEGMENTED_REDUCE=0, other detectors, a non-contiguous layout)"""
        if not hasattr(self, '_early_span'):
            self._early_span = None
            pre = getattr(self.model, 'EARLY_REDUCE_PREFIXES', None)
            flat = getattr(self.model, '_scda_flat', None)
            if pre and flat is not None and os.environ.get("SCDA_SEGMENTED_REDUCE", "1") != "0":
                self._early_span = flat.span_of([p for n, p in self.model.named_parameters() if n.startswith(tuple(pre)) and p.requires_grad])
        return self._early_span

    def _reduce(self, module, async_op):
        self._finish_grads(module)
        return self._all_reduce(module, async_op)

    # ---- phases 1 + 2 (image discriminators, patch discriminator): forward, losses, both backward passes ----------------------
    def _dis_out_len(self, crops):
        """elements per cluster row of the image discriminator's output for crops [C, 3, h, w] (n_layer stride-2 convs, 1x1 head)"""
        h, w = crops.shape[2], crops.shape[3]
        for m in self.dis.model_A:
            conv = m.model[0] if hasattr(m, "model") else m
            k, st, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            h, w = (h + 2 * pd - k) // st + 1, (w + 2 * pd - k) // st + 1
        return h * w

    def _dis_patch_out_len(self):
        return 2 * self.dis_patch.n_out

    def _phase12(self, dev_in):
        """device work of phases 1 and 2 given their inputs and label tensors (:560-640).  No host decision inside: this is what
        the hipGraph of _phase12_graph records."""
        src_fake, tgt_fake, x_small, t_small, tgt_patch, src_patch, score1, score0, score0p, score1p = dev_in
        ws = float(self.world_size)
        bce, adv = A.binary_cross_entropy, A.adversarial_loss
        self.opt['dis'].zero_grad()
        d_src_fake, d_tgt_fake = self.dis(src_fake, tgt_fake)
        d_src_real, d_tgt_real = self.dis(x_small, t_small)
        tgt_pro = self.dis_patch(tgt_patch)                          # [C, 512] in (0,1); also updates BN statistics
        w_tgt = N.row_mean(tgt_pro.detach().contiguous())            # per-cluster weight (its gradient is dead here)
        src_pro = self.dis_patch(src_patch)
        # The adversarial terms are the reference's sums over clusters of F.binary_cross_entropy(torch.sigmoid(d)[c], label) (one
        # mean per cluster row), each group evaluated by ONE fused kernel (A.adversarial_loss) on the logits.
        adloss = adv([(d_src_fake, score1, None), (d_src_real, score0, None),          # ad_src  (:584-588)
                      (d_tgt_fake, score0, w_tgt), (d_tgt_real, score1, None)],         # ad_tgt  (:596-600)
                     scale=1.0 / ws)
        adloss.backward()
        self._finish_grads(self.dis)
        # eager: the discriminators' all-reduce starts here, underneath phase 2; as a hipGraph region the collectives are issued by
        # the caller behind the replay (step): 0.75 + 15 MB, still underneath phase 3
        w1 = None if self._in_graph else self._all_reduce(self.dis, async_op=True)
        mark('phase1')
        self.opt['dis_patch'].zero_grad()
        dis_patch_loss = (bce(src_pro, score1p) + bce(tgt_pro, score0p)) / ws
        dis_patch_loss.backward()
        self._finish_grads(self.dis_patch)
        w2 = None if self._in_graph else self._all_reduce(self.dis_patch, async_op=True)
        return adloss.detach(), dis_patch_loss.detach(), w1, w2

    # ---- the GAN part of the iteration as three regions without a host decision inside --------------------------------------
    # A: decoder forward + phases 1 and 2        B: phase 3 (forward, losses, backward into the decoders)
    # C: the forward-only part of phase 4 (the logged adversarial term of the detector loss)
    # Between them sit the optimiser steps (and, data-parallel, the waits for the all-reduces).  The eager iteration calls them as
    # plain functions; with SCDA_GAN_GRAPH=1 each is recorded ONCE as a hipGraph and replayed (_GanGraphs below).
    def _region_a(self, t):
        src_recon, tgt_recon = self.dec(t['src_patch'], t['tgt_patch'])        # [C, 3, recon, recon]
        mark('crops+dec_fwd_enqueued')
        adloss, dis_patch_loss, w1, w2 = self._phase12((src_recon.detach(), tgt_recon.detach(), t['x_small'], t['t_small'], t['tgt_patch'],
                                                        t['src_patch'], t['score1'], t['score0'], t['score0p'], t['score1p']))
        return src_recon, tgt_recon, adloss, dis_patch_loss, w1, w2

    def _region_b(self, src_recon, tgt_recon, t):
        ws = float(self.world_size)
        adv = A.adversarial_loss
        self.opt['dec'].zero_grad()
        with _Frozen(self.dis):
            d_src_fake, d_tgt_fake = self.dis(src_recon, tgt_recon)   # gradient flows to the decoders only
            with torch.no_grad():
                d_src_real, d_tgt_real = self.dis(t['x_small'], t['t_small'])
                w_tgt2 = N.row_mean(self.dis_patch(t['tgt_patch']).contiguous())
            fake1_tgt = adv([(d_tgt_fake, t['one_t'], w_tgt2), (d_tgt_real, t['zero_t'], w_tgt2)])       # :675-681
            fake1_src = adv([(d_src_fake, t['one_s'], None), (d_src_real, t['zero_s'], None)])           # :683-687
            recon_loss = (fake1_src + fake1_tgt) / ws
            recon_loss.backward()
        self._finish_grads(self.dec)
        w3 = None if self._in_graph else self._all_reduce(self.dec, async_op=True)
        return recon_loss.detach(), fake1_src.detach(), w_tgt2, w3

    def _region_c(self, t, w_tgt2):
        adv = A.adversarial_loss
        with torch.no_grad():
            swap_src, swap_tgt = self.dec(t['tgt_patch'], t['src_patch'])
            q_src, q_tgt = self.dis(swap_src, swap_tgt)
            fake_loss_source = adv([(q_tgt, t['ones_all'], None)], scale=1.0 / q_tgt.shape[0])      # mean over all elements (:723)
            fake_loss_target = adv([(q_src, t['ones_row'], w_tgt2)])                                 # :725-732
        return fake_loss_source, fake_loss_target

    def _gan_graph_ok(self):
        """The three regions above as hipGraphs (~330 of the iteration's ~700 launches, fixed shapes; SCDA_GAN_GRAPH=0 turns them off).
        Not with gradient capture, the reference-style schedule or the parity tests' hooks; not when a dropout of the decoders would
        take its seed as a kernel argument (a recording would freeze it: only the residual blocks' fused tail reads a device seed).
        The all-reduces of a data-parallel run are host calls: issued behind each replay (step), never recorded."""
        from . import layers as L
        from .dropin.models.faster_rcnn.common_net import INSResBlock
        from .hostenv import blocking_sync_selected
        if (blocking_sync_selected() or self.collectives) and os.environ.get("SCDA_GAN_GRAPH") != "1":
            # several ranks per node (bench.py): blocking waits, and in that mode a graph launch costs device time; and with the
            # collectives' stream in the process the replays interact badly with it (one-rank RCCL group, scripts/onerank_rccl_cost.py:
            # 27.6 ms per iteration with the graphs, 19.7 without, 19.4 without collectives) -- data-parallel runs stay eager
            return False
        if not (os.environ.get("SCDA_GAN_GRAPH", "1") != "0" and self.device.type == "cuda" and not self.capture and self.early_backward
                and not P.active() and self.flat):
            return False
        ok = getattr(self, "_graphable_nets", None)
        if ok is None:
            fused = set()
            for blk in self.dec.modules():
                if isinstance(blk, INSResBlock) and blk.tail_fusable():
                    fused.add(id(blk.model[-1]))
            ok = all(id(m) in fused or m.p == 0.0 for net in (self.dec, self.dis, self.dis_patch) for m in net.modules()
                     if isinstance(m, L.Dropout))
            self._graphable_nets = ok
        return ok

    def _ones_row(self, row):
        c = getattr(self, "_ones_row_cache", None)
        if c is None or tuple(c.shape) != tuple(row):
            c = self._ones_row_cache = torch.ones(row, dtype=torch.float32, device=self.device)
        return c

    def _gan_graphs(self):
        g = getattr(self, "_gg", None)
        if g is None:
            g = self._gg = _GanGraphs(self)
        g.calls += 1
        return g

    def step(self, image, gts, image_info, target, gt_masks=None):
        """image/target [1,3,H,W] on the device; gts [1,G,5]; image_info [1,3]; gt_masks [1,G,H,W] binary (detectors with a mask
        branch, BASELINE configs[4]) -> dict of 0-dim loss tensors"""
        with P.installed(self.probe):
            return self._step(image, gts, image_info, target, gt_masks)

    def _step(self, image, gts, image_info, target, gt_masks=None):
        dev, ws, C = self.device, float(self.world_size), self.cluster_num
        x = {'cfg': self.cfg, 'image': image, 'image_info': image_info, 'ground_truth_bboxes': gts,
             'ignore_regions': None, 'cluster_num': self.cluster_num, 'threshold': self.threshold}
        if gt_masks is not None:
            x['ground_truth_masks'] = gt_masks
        pending = {}
        self._in_graph = False
        for sch in self._warmup or ():     # :510-514 -- the warm-up schedulers step at the top of the iteration
            sch.step()

        def detector_backward(losses):
            # Phase (4)'s gradient depends only on the four detector losses (the adversarial term of the reference's
            # detector loss carries no gradient into the detector), so its backward -- and, multi-GPU,
            # the 547 MB all-reduce -- starts here and runs underneath the rest of the iteration.
            total = losses[0] + losses[1] + losses[2] + losses[3]
            for extra in losses[4:]:          # branches beyond RPN + RCNN (the mask loss of configs[4])
                total = total + extra
            det_loss = total / ws
            self.opt['det'].zero_grad()
            det_loss.backward()
            pending['det_loss'] = det_loss.detach()
            seg = pending.get('seg')
            if seg is not None:          # the head's slice left from inside the backward; the conv body's (and what else is left) now
                self._finish_grads(self.model)
                pending['w4'] = seg.launch_rest()
            else:
                pending['w4'] = self._reduce(self.model, async_op=True)

        if self.early_backward:
            x['_after_source_losses'] = detector_backward
            x['_side_stream'] = self.side
        # data parallel: the detector's all-reduce in two pieces, the classifier + heads (480 of 547 MB, final behind FC6's weight
        # gradient) from a hook inside the backward, the rest behind it (not under gradient capture: the tests' per-rank traces
        # are taken from the whole bucket at the end of the backward)
        span = self._det_early_span() if (self.collectives and not self.capture) else None
        if span is not None:
            pending['seg'] = SegmentedReduce(self.model._scda_flat, span)

            def head_gradients_enqueued():
                pending['seg'].launch_early()
                self.early_reduces += 1
            x['_after_head_backward'] = head_gradients_enqueued
        mark('step_begin')
        outputs = self.model(x, target)
        ctr_s, ctr_t = outputs['cluster_centers']
        src_patch, tgt_patch = outputs['cluster_features']          # [C, threshold, 4096] leaves
        t = {'x_small': _crops(image, get_corner_from_center(ctr_s, self.recon, self.new_w, self.new_h), self.recon),
             't_small': _crops(target, get_corner_from_center(ctr_t, self.recon, self.new_w, self.new_h), self.recon),
             'src_patch': src_patch, 'tgt_patch': tgt_patch}
        graphs = self._gan_graphs() if self._gan_graph_ok() else None
        if graphs is not None and not graphs.fits(t):
            graphs = None              # other cluster / crop shapes than the recorded ones: this iteration runs eagerly
        in_graph = graphs is not None and graphs.recording()

        # The adversarial terms below are the reference's sums over clusters of F.binary_cross_entropy(torch.sigmoid(d)[c], label)
        # (one mean per cluster row), each group evaluated by ONE fused kernel (A.adversarial_loss) on the logits.
        # ---------------- decoder forward, (1) image discriminators, (2) patch discriminator ----------------
        # host side first: the four label draws in the reference's order (nothing else consumes numpy's RNG in between)
        n_dis = self._dis_out_len(t['x_small'])
        row = (1, n_dis)
        pro_shape = (src_patch.shape[0], self._dis_patch_out_len())
        t['score1'], t['score0'], t['score0p'], t['score1p'] = _labels(
            [('s', 1, row), ('s', 0, row), ('s', 0, pro_shape), ('s', 1, pro_shape)], dev)
        self._in_graph = in_graph      # (reset in a finally: an exception inside a region -- an OOM while recording, a shape error -- must
        try:                           # not leave the flag set, or the next eager step would skip every phase's all-reduce)
            if graphs is not None and graphs.ready('a'):
                src_recon, tgt_recon, adloss, dis_patch_loss = graphs.run('a', t)
                w1 = w2 = None
            else:
                with _recording(graphs, 'a', t) as rec:
                    src_recon, tgt_recon, adloss, dis_patch_loss, w1, w2 = rec(lambda tt: self._region_a(tt))
        finally:
            self._in_graph = False
        if in_graph:                   # the region's collectives, behind the replay
            w1 = self._all_reduce(self.dis, async_op=True)
            w2 = self._all_reduce(self.dis_patch, async_op=True)
        mark('phase2')
        if w1 is not None:
            w1.wait()
            self._grab_reduced('dis', self.dis)
        self.opt['dis'].step()
        if w2 is not None:
            w2.wait()
            self._grab_reduced('dis_patch', self.dis_patch)
        self.opt['dis_patch'].step()

        # ---------------- (3) decoders ----------------
        t['one_t'], t['zero_t'], t['one_s'], t['zero_s'] = _labels([('h', 1, row), ('h', 0, row), ('h', 1, row), ('h', 0, row)], dev)
        self._in_graph = in_graph
        try:
            if graphs is not None and graphs.ready('b'):
                recon_loss, fake1_src, w_tgt2 = graphs.run('b', t)
                w3 = None
            else:
                with _recording(graphs, 'b', t) as rec:
                    recon_loss, fake1_src, w_tgt2, w3 = rec(lambda tt: self._region_b(src_recon, tgt_recon, tt))
        finally:
            self._in_graph = False
        if in_graph:
            w3 = self._all_reduce(self.dec, async_op=True)
        mark('phase3')

        # ---------------- (4) detector ----------------
        # The swapped reconstruction of the reference's phase 4 only feeds the LOGGED loss: the cluster features are
        # leaves, so the 0.1*(fake_loss_source + fake_loss_target) term carries no gradient into the detector
        # (SURVEY.md 3.1).  The detector backward therefore started long ago (early backward) and overlaps the decoder
        # all-reduce; the logged term is evaluated afterwards, without an autograd graph, with the freshly stepped decoder as in
        # the reference (dec_optimizer.step() precedes it, :704 vs :716).
        rpn_cls, rpn_loc, rcnn_cls, rcnn_loc = outputs['losses'][:4]
        if not self.early_backward:
            detector_backward(outputs['losses'])
        det_loss, w4 = pending['det_loss'], pending['w4']
        if w3 is not None:
            w3.wait()
            self._grab_reduced('dec', self.dec)
        self.opt['dec'].step()
        t['ones_all'] = _hard(1, (src_patch.shape[0], n_dis), dev)
        t['ones_row'] = self._ones_row(row)
        if graphs is not None and graphs.ready('c'):
            fake_loss_source, fake_loss_target = graphs.run('c', t)
        else:
            with _recording(graphs, 'c', t) as rec:
                fake_loss_source, fake_loss_target = rec(lambda tt: self._region_c(tt, w_tgt2))
        loss = det_loss + 0.1 * (fake_loss_source + fake_loss_target) / ws
        if w4 is not None:
            w4.wait()
            self._grab_reduced('det', self.model)
        self.opt['det'].step()
        mark('phase4+det_step')

        self.last_num_proposals = outputs.get('num_proposals')     # post-NMS proposal counts (source, target) of this iteration
        extra_logged = {'mask_loss': outputs['losses'][4].detach()} if len(outputs['losses']) > 4 else {}
        return {'loss': loss.detach() * ws, 'rpn_cls': rpn_cls.detach(), 'rpn_loc': rpn_loc.detach(),
                'rcnn_cls': rcnn_cls.detach(), 'rcnn_loc': rcnn_loc.detach(), 'rpn_acc': outputs['accuracy'][0],
                'rcnn_acc': outputs['accuracy'][1], 'fake_loss_target': fake_loss_target, 'fake_loss_source': fake_loss_source,
                'recon_loss': recon_loss.detach(), 'adloss': adloss.detach(), 'dis_patch_loss': dis_patch_loss.detach(),
                'fake_loss1_source': fake1_src.detach(), **extra_logged}
