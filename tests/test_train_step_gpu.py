"""The four-phase SCDA iteration on the MI355X (scda_amd.train_step.ScdaTrainer, all compute through the C ABI)
against the CPU oracle (oracle/torch_ref.py, itself pinned to the reference's train()).

Same seeded weights / inputs / numpy RNG stream; the oracle's dropout keep-masks are replayed on the device.
Tolerances: losses 1e-4 relative (fp32, different summation order); gradients per tensor as relative L2
(median < 5e-3, worst < 5e-2 -- see the comment at the check for why not tighter)."""
import numpy as np
import pytest
import torch

import model_common as mc

pytestmark = pytest.mark.gpu


def build_product():
    from scda_amd.dropin.models.faster_rcnn.vgg_adver_expansion_cluster import vgg16
    from scda_amd.train_step import builder_gan
    det = vgg16(cfg=dict(mc.CFG['shared'], gan_model_flag=2))
    dis, dec, dis_patch = builder_gan()
    return det, dec, dis, dis_patch


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# "fp32, tol 1e-5 rel" (relative to max(1, |loss|)) -- asserted as stated.  Achieved (round 6, printed and
# logged by model_common.check_losses): worst 1.3e-6 free-running at 256x512 (fake_loss_target), 1.4e-7 / 1.1e-7 with the oracle's
# selections replayed at 256x512 / 512x1024 -- the Winograd layers' 3e-5 per-element differences average out in the loss means.
LOSS_REL = 1e-5
LOSS_KEYS = ('rpn_cls', 'rpn_loc', 'rcnn_cls', 'rcnn_loc', 'adloss', 'dis_patch_loss', 'recon_loss', 'fake_loss1_source',
             'fake_loss_target', 'fake_loss_source', 'loss')


def test_few_target_proposals_fall_back_to_source_clusters(cuda):
    """fewer than 512 target proposals: the reference reuses the SOURCE cluster features / centres for the target side
    (faster_rcnn_adver_expansion_reweight_cluster.py:255-262); the source RoIs are padded to 512 by resampling.
    Forced here with post_nms_top_n = 300; every logged loss must still match the oracle."""
    import copy
    from scda_amd import layers as L
    from scda_amd.train_step import ScdaTrainer
    H, W, lr = 256, 512, 1e-3
    cfg = copy.deepcopy(mc.CFG)
    cfg['train_rpn_proposal_cfg']['post_nms_top_n'] = 300
    ref, _, masks = mc.oracle_iteration(H, W, lr=lr, record_masks=True, cfg=cfg)
    torch.manual_seed(1)
    tr = ScdaTrainer(cfg, cuda, lr=lr, new_w=W, new_h=H, models=mc.seeded_models(build_product))
    src, tgt, gts, info = mc.seeded_inputs(H, W)
    tape = list(masks)
    with mc.probed(dropout_masks=lambda shape, p, device: tape.pop(0).to(device)):
        np.random.seed(mc.SEEDS['numpy'])
        out = tr.step(src.to(cuda), gts, info, tgt.to(cuda))
        torch.cuda.synchronize()
    assert not tape
    mc.check_losses(out, ref, LOSS_KEYS, LOSS_REL, "few_target_proposals[256x512]")


def test_two_iterations_track_oracle(cuda):
    """two consecutive iterations (optimiser state, BN statistics and RNG streams carried along): the logged losses of
    every step stay on the oracle's trajectory.  Step 1 agrees to 1e-4 (previous test); from step 2 on the tolerance is 2e-2:
    Adam's first step moves EVERY weight by ~lr whatever its gradient's magnitude, so the (rare) elements whose gradient is
    within round-off of zero step the other way on the two machines (bounded in the previous test: <= 2 % per tensor), and
    that one-off perturbation of the weights shows up as ~1e-2 in the next losses (measured: rcnn_cls 0.7939 vs 0.7874; by the
    third step the two trajectories are 2 % apart and the comparison stops saying anything about the kernels)."""
    from scda_amd import layers as L
    from scda_amd.train_step import ScdaTrainer
    H, W, lr = 256, 512, 1e-4
    ref, _, masks = mc.oracle_iteration(H, W, lr=lr, record_masks=True, steps=2)
    torch.manual_seed(1)
    tr = ScdaTrainer(mc.CFG, cuda, lr=lr, new_w=W, new_h=H, models=mc.seeded_models(build_product))
    src, tgt, gts, info = mc.seeded_inputs(H, W)
    src, tgt = src.to(cuda), tgt.to(cuda)
    tape = list(masks)
    with mc.probed(dropout_masks=lambda shape, p, device: tape.pop(0).to(device)):
        np.random.seed(mc.SEEDS['numpy'])
        outs = [tr.step(src, gts, info, tgt) for _ in range(2)]
        torch.cuda.synchronize()
    assert not tape
    for i, (out, want) in enumerate(zip(outs, ref['_history'])):
        for k in LOSS_KEYS:
            a, b = float(out[k]), float(want[k])
            assert abs(a - b) <= (1e-4 if i == 0 else 2e-2) * max(1.0, abs(b)), (i, k, a, b)
    assert float(outs[1]['rcnn_cls']) != float(outs[0]['rcnn_cls'])        # the weights did move


def test_three_warmup_iterations_track_oracle(cuda):
    """the reference's per-iteration exponential warm-up (tools/faster_rcnn_train_val.py:346-364, utils/lr_helper.py:33-49) on
    the fused optimisers: three iterations warming up to 8x (the 8-GPU, batch-1 run): the learning rate that reaches the Adam
    kernel follows the oracle's exactly (1, sqrt 8, 8) x base, the losses stay on the oracle's trajectory (tolerances as in
    test_two_iterations_track_oracle), and end_warmup() turns the magnified rate into the MultiStepLR base (:365-376)"""
    from scda_amd import layers as L
    from scda_amd.train_step import ScdaTrainer
    H, W, lr = 256, 512, 2e-5
    ref, _, masks = mc.oracle_iteration(H, W, lr=lr, record_masks=True, steps=3, warmup=(3, 8))
    torch.manual_seed(1)
    tr = ScdaTrainer(mc.CFG, cuda, lr=lr, new_w=W, new_h=H, models=mc.seeded_models(build_product))
    gamma = tr.begin_warmup(3, world_size=8)
    assert abs(gamma - 8 ** 0.5) < 1e-12
    src, tgt, gts, info = mc.seeded_inputs(H, W)
    src, tgt = src.to(cuda), tgt.to(cuda)
    tape = list(masks)
    outs, lrs = [], []
    with mc.probed(dropout_masks=lambda shape, p, device: tape.pop(0).to(device)):
        np.random.seed(mc.SEEDS['numpy'])
        for _ in range(3):
            outs.append(tr.step(src, gts, info, tgt))
            lrs.append([o.param_groups[0]['lr'] for o in tr.opt.values()])
        torch.cuda.synchronize()
    assert not tape
    for i, want in enumerate(ref['_history']):
        assert all(abs(v - want['_lr']) <= 1e-12 * want['_lr'] for v in lrs[i]), (i, lrs[i], want['_lr'])
        assert abs(want['_lr'] - lr * 8 ** (i / 2)) <= 1e-12
        for k in LOSS_KEYS:
            a, b = float(outs[i][k]), float(want[k])
            assert abs(a - b) <= (1e-4 if i == 0 else 2e-2) * max(1.0, abs(b)), (i, k, a, b)
    tr.end_warmup()
    tr.set_epoch_schedule([2, 3])
    assert [round(tr.begin_epoch() / lr, 6) for _ in range(3)] == [8.0, 0.8, 0.08]
    assert all(o.param_groups[0]['initial_lr'] == pytest.approx(8 * lr) for o in tr.opt.values())


def test_iteration_matches_oracle(cuda):
    """each side breaks its own ties (no replay of selections): possible at 256x512, where no two RPN scores of this seed sit
    within fp32 round-off of each other; at 512x1024 (30720 anchors) some do, one swapped pair changes the sampled RoIs
    (scripts/debug_fullsize_parity.py: 130 of 2000 proposals differ) -- the BASELINE-size comparison is
    test_gradients_with_replayed_selections, which pins the discrete decisions and checks everything else more tightly"""
    from scda_amd import layers as L
    from scda_amd.train_step import ScdaTrainer
    H, W, lr = 256, 512, 1e-3
    ref, ref_models, masks = mc.oracle_iteration(H, W, lr=lr, record_masks=True, capture=True)

    torch.manual_seed(1)
    models = mc.seeded_models(build_product)
    tr = ScdaTrainer(mc.CFG, cuda, lr=lr, new_w=W, new_h=H, models=models)
    tr.capture = True
    src, tgt, gts, info = mc.seeded_inputs(H, W)
    tape = list(masks)

    def replay(shape, p, device):
        m = tape.pop(0)
        assert tuple(m.shape) == tuple(shape), (m.shape, shape)
        return m.to(device)

    with mc.probed(dropout_masks=replay):
        np.random.seed(mc.SEEDS['numpy'])
        out = tr.step(src.to(cuda), gts, info, tgt.to(cuda))
        torch.cuda.synchronize()
    assert not tape, "the device consumed fewer dropout masks than the oracle drew"

    mc.check_losses(out, ref, LOSS_KEYS, LOSS_REL, "iteration_matches_oracle[256x512]")
    assert abs(float(out['rpn_acc'][0]) - float(ref['rpn_acc'][0])) < 0.5
    assert abs(float(out['rcnn_acc'][0]) - float(ref['rcnn_acc'][0])) < 0.5

    # per-phase gradients (the oracle's are captured right after the phase's backward).
    # Metric: relative L2 per tensor.  Two effects make an element-wise bound meaningless here and are accounted for:
    #  * ReLU / LeakyReLU / max-pool are not differentiable at ties: activations agree to ~1e-6 between the CPU and the
    #    MI355X, so a handful of pre-activations within round-off of 0 flip their mask and each flip perturbs a 3x3xC
    #    neighbourhood of the upstream gradient (measured: 1 flip in conv4_1 -> 1.5k affected elements, scripts/debug_layers.py)
    #    (LeakyReLU flips in one discriminator branch change a local slope by 100x: the branch with flips shows ~1e-3,
    #    the other ~1e-6, in both the discriminator's own gradients and the decoder gradients that pass through it)
    #  * conv biases directly in front of InstanceNorm have a mathematically ZERO gradient (|g| ~ 1e-12 round-off on both sides)
    def rel_l2(a, b):
        a = a.detach().double().cpu(); b = b.detach().double().cpu()
        return float((a - b).norm() / (b.norm() + 1e-30))

    report = {}
    for name in ('dis', 'dis_patch', 'dec', 'det'):
        rg, pg = ref['_trace'][name], tr.trace[name]
        assert set(rg) == set(pg)
        errs = []
        for k in rg:
            scale = float(max(t.abs().max() for t in rg.values()))
            if float(rg[k].abs().max()) < 1e-6 * scale:
                assert float(pg[k].abs().max()) < 1e-5 * scale, k      # zero gradient stays (numerically) zero
                continue
            errs.append(rel_l2(pg[k], rg[k]))
        errs.sort()
        report[name] = (errs[len(errs) // 2], errs[-1])
    for name, (median, worst) in report.items():
        assert median < 5e-3 and worst < 5e-2, report

    # parameters after the step: Adam's FIRST step moves every weight by lr*g/(|g|+eps) ~ lr*sign(g), so an element whose
    # gradient is within round-off of zero may step the other way (|diff| up to 2*lr); those must stay rare
    for pm, rm in zip((tr.model, tr.dec, tr.dis, tr.dis_patch), ref_models):
        rsd = rm.state_dict()
        for k, v in pm.state_dict().items():
            if v.dtype.is_floating_point:
                d = (v.detach().cpu().double() - rsd[k].double()).abs()
                assert float(d.max()) <= 2.001 * lr, (k, float(d.max()))
                flipped = int((d > 0.1 * lr).sum())
                assert flipped <= max(1, int(0.02 * d.numel())), (k, flipped, d.numel())
    dp = tr.dis_patch.state_dict()
    assert int(dp['model_A_patch.0.model.1.num_batches_tracked']) == 3


@pytest.mark.parametrize("H,W", [(256, 512), (512, 1024)])
def test_gradients_with_replayed_selections(cuda, H, W):
    """The same iteration with the oracle's non-differentiable SELECTIONS replayed on the device (ReLU / LeakyReLU sign masks,
    2x2 max-pool winners, RoI max-pool argmax -- a scda_amd.probe.Probe's `replay`) in addition to the dropout masks: what is left
    is the kernels' arithmetic.  Every gradient tensor of every phase then agrees with the oracle's to 1e-4 relative L2
    (test_iteration_matches_oracle, which lets each side break its own ties, needs 5e-2)."""
    from scda_amd import autograd_ops as A
    from scda_amd import layers as L
    from scda_amd.train_step import ScdaTrainer
    lr = 1e-3
    ref, ref_models, masks = mc.oracle_iteration(H, W, lr=lr, record_masks=True, capture=True, record_selections=True)
    torch.manual_seed(1)
    tr = ScdaTrainer(mc.CFG, cuda, lr=lr, new_w=W, new_h=H, models=mc.seeded_models(build_product))
    tr.capture = True
    src, tgt, gts, info = mc.seeded_inputs(H, W)
    tape = list(masks)
    rp = mc.ReplaySource(ref['_selections'], cuda)
    # the Probe goes in through the trainer (scda_amd/probe.py): selections, identical proposal ranking (rpn_output), dropout masks
    tr.probe = mc.Probe(replay=rp, rpn_output=rp.rpn, dropout_masks=lambda shape, p, device: tape.pop(0).to(device))
    np.random.seed(mc.SEEDS['numpy'])
    out = tr.step(src.to(cuda), gts, info, tgt.to(cuda))
    torch.cuda.synchronize()
    used = rp.used
    tr.probe = None
    assert not tape and used >= 40, used
    mc.check_losses(out, ref, LOSS_KEYS, LOSS_REL, "replayed_selections[%dx%d]" % (H, W))

    def rel_l2(a, b):
        a = a.detach().double().cpu(); b = b.detach().double().cpu()
        return float((a - b).norm() / (b.norm() + 1e-30))

    worst = {}
    for name in ('dis', 'dis_patch', 'dec', 'det'):
        rg, pg = ref['_trace'][name], tr.trace[name]
        scale = float(max(t.abs().max() for t in rg.values()))
        for k in rg:
            if float(rg[k].abs().max()) < 1e-6 * scale:     # conv biases in front of InstanceNorm: mathematically zero
                continue
            e = rel_l2(pg[k], rg[k])
            if e > worst.get(name, ('', 0.0))[1]:
                worst[name] = (k, e)
    print("worst gradient tensor per net (relative L2):", worst)     # pytest -s: the margin against the 1e-4 bound
    assert all(e <= 1e-4 for _, e in worst.values()), worst


def test_fullsize_free_running_iteration_tracks_oracle(cuda, monkeypatch):
    """512 x 1024 (the BASELINE size) with NOTHING replayed but the dropout masks: each side ranks its own RPN scores, breaks its own
    ties and samples its own RoIs.  Among 30720 fp32 objectness scores some pairs sit closer than the 1e-6 by which two correct
    implementations differ, so a few proposals swap places and the sampled RoI sets part ways (why the arithmetic is compared with the
    selections replayed, test_gradients_with_replayed_selections) -- but the iteration as a whole must stay on the oracle's:
    identical anchor labelling (positive / negative counts), >= 90 % of the 2000 source proposals identical, every loss within 2 %."""
    from scda_amd import layers as L
    from scda_amd.train_step import ScdaTrainer
    from scda_amd.dropin.functions import anchor_target as AT
    from scda_amd.dropin.models.faster_rcnn import faster_rcnn_adver_expansion_reweight_cluster as FRC
    H, W, lr = 512, 1024, 1e-3
    counts = []
    orig = AT.compute_anchor_targets

    def counting(*a, **kw):
        out = orig(*a, **kw)
        lab = out[0].detach().cpu()
        counts.append((int((lab == 1).sum()), int((lab == 0).sum()), int((lab == -1).sum()), float(out[3])))
        return out

    monkeypatch.setattr(AT, "compute_anchor_targets", counting)      # the oracle imports it at call time ...
    monkeypatch.setattr(FRC, "compute_anchor_targets", counting)     # ... the product's detector module holds its own reference
    ref, _, masks = mc.oracle_iteration(H, W, lr=lr, record_masks=True)
    n_ref = len(counts)
    assert n_ref == 1
    torch.manual_seed(1)
    tr = ScdaTrainer(mc.CFG, cuda, lr=lr, new_w=W, new_h=H, models=mc.seeded_models(build_product))
    src, tgt, gts, info = mc.seeded_inputs(H, W)
    grabbed = {}
    fwd = tr.model.forward

    def grabbing(x, target):
        out = fwd(x, target)
        grabbed.update(out)
        return out

    tr.model.forward = grabbing
    tape = list(masks)
    with mc.probed(dropout_masks=lambda shape, p, device: tape.pop(0).to(device)):
        np.random.seed(mc.SEEDS['numpy'])
        out = tr.step(src.to(cuda), gts, info, tgt.to(cuda))
        torch.cuda.synchronize()
    assert not tape
    # anchor labelling: a function of the ground truth and numpy's generator alone
    assert len(counts) == 2 and counts[0] == counts[1], counts
    assert counts[0][0] > 0 and counts[0][0] + counts[0][1] == 256           # the RPN batch
    # proposals
    p_ref, p_got = ref['_outputs']['predict'][0].numpy(), grabbed['predict'][0].cpu().numpy()
    assert p_ref.shape == p_got.shape == (2000, 6), (p_ref.shape, p_got.shape)
    same_place = int((np.abs(p_ref[:, :5] - p_got[:, :5]).max(1) <= 1e-3).sum())
    ref_rows = {tuple(np.round(r, 2)) for r in p_ref[:, 1:5]}
    anywhere = sum(tuple(np.round(r, 2)) in ref_rows for r in p_got[:, 1:5])
    assert anywhere >= 1800, (same_place, anywhere)          # >= 90 % of the oracle's proposals are in the device's list
    # losses
    for k in LOSS_KEYS:
        a, b = float(out[k]), float(ref[k])
        assert abs(a - b) <= 0.02 * max(abs(b), 1e-3), (k, a, b)
    assert abs(float(out['rpn_acc'][0]) - float(ref['rpn_acc'][0])) < 1.0
    assert abs(float(out['rcnn_acc'][0]) - float(ref['rcnn_acc'][0])) < 2.0


def test_vgg16_bn_detector_trains(cuda):
    """the batch-norm backbone variant (vgg16_bn, `--arch vgg16bn_FasterRCNN` in the reference driver): one full iteration
    through the same step; finite losses, BN statistics updated, torchvision's vgg16_bn key layout"""
    from scda_amd.dropin.models.faster_rcnn.vgg_adver_expansion_cluster import vgg16_bn
    from scda_amd.train_step import ScdaTrainer, builder_gan
    H, W = 256, 512
    torch.manual_seed(2)
    det = vgg16_bn(cfg=dict(mc.CFG['shared'], gan_model_flag=2))
    keys = list(det.state_dict().keys())
    assert keys[:7] == ['features.0.weight', 'features.0.bias', 'features.1.weight', 'features.1.bias',
                        'features.1.running_mean', 'features.1.running_var', 'features.1.num_batches_tracked']
    dis, dec, dis_patch = builder_gan()
    tr = ScdaTrainer(mc.CFG, cuda, lr=1e-4, new_w=W, new_h=H, models=(det, dec, dis, dis_patch))
    src, tgt, gts, info = mc.seeded_inputs(H, W)
    np.random.seed(3)
    out = tr.step(src.to(cuda), gts, info, tgt.to(cuda))
    torch.cuda.synchronize()
    for k in LOSS_KEYS:
        assert np.isfinite(float(out[k])), k
    sd = tr.model.state_dict()
    assert int(sd['features.1.num_batches_tracked']) == 2          # source and target pass
    assert float(sd['features.1.running_mean'].abs().sum()) > 0


def test_gan_phases_as_hipgraph_match_eager(cuda, monkeypatch):
    """The default (SCDA_GAN_GRAPH=0 turns it off): the GAN part of the iteration -- decoder forward + phases 1 and 2, phase 3 (backward into the decoders
    included), the forward-only part of phase 4: ~330 launches on two streams -- recorded once as three hipGraphs and replayed from
    the third iteration on, the decoders' dropout seeds read from device memory.  Same kernels, same order, same inputs, same random
    draws: every parameter bucket, the BN statistics / counters and ALL logged losses must be BIT-identical to the eager step over 52
    iterations (50 of them replays)."""
    from scda_amd.train_step import ScdaTrainer
    res = {}
    for name in ("eager", "graph"):
        if name == "graph":
            monkeypatch.delenv("SCDA_GAN_GRAPH", raising=False)
        else:
            monkeypatch.setenv("SCDA_GAN_GRAPH", "0")
        torch.manual_seed(1)
        tr = ScdaTrainer(mc.CFG, cuda, lr=1e-3, new_w=512, new_h=256, models=mc.seeded_models(build_product))
        np.random.seed(5)
        losses = []
        for it in range(52):
            src, tgt, gts, info = mc.seeded_inputs(256, 512, sample=it % 3)
            out = tr.step(src.to(cuda), gts, info, tgt.to(cuda))
            losses.append([float(out[k]) for k in ('loss', 'adloss', 'dis_patch_loss', 'recon_loss', 'fake_loss_source', 'fake_loss_target',
                                                   'fake_loss1_source')])
        torch.cuda.synchronize()
        if name == "eager":
            assert getattr(tr, "_gg", None) is None
        else:
            assert sorted(tr._gg.reg) == ['a', 'b', 'c'] and tr._gg.calls == 52 and tr._gg.arena.n == 12     # 2 x 6 dropout launches
        res_rng = (torch.rand(1).item(), np.random.rand())          # both generators end in the same state
        sd = tr.dis_patch.state_dict()
        res[name] = dict(losses=losses, rng=res_rng, sums={k: (float(f.data.double().sum()), float(f.data.double().abs().sum())) for k, f in tr.flat.items()},
                         bn={k: v.double().sum().item() for k, v in sd.items() if 'running' in k or 'num_batches' in k})
    assert res["graph"]["losses"] == res["eager"]["losses"]
    assert res["graph"]["rng"] == res["eager"]["rng"]
    assert res["graph"]["sums"] == res["eager"]["sums"]
    assert res["graph"]["bn"] == res["eager"]["bn"]
