"""Golden vectors for the model level: ONE iteration of the reference's own train()
(tools/faster_rcnn_train_val.py:461-770, imported unmodified) on seeded synthetic inputs, CPU.
Called from make_golden.py --only train.  Output: tests/golden/train_step_<H>x<W>.npz"""
import logging
import os
import re
import time

import numpy as np
import torch

import seeded_init as si

SEEDS = dict(det=11, dec=12, dis=13, dis_patch=14, images=21, gts=22, torch=31, numpy=32)


class _Sched:
    def __init__(self, opt, lr):
        self.optimizer, self._lr = opt, lr

    def get_lr(self):
        return [self._lr]

    def step(self):
        pass


class _Capture(logging.Handler):
    def __init__(self):
        super().__init__()
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def run_reference_iteration(ns, H, W, lr=1e-3, G=6):
    T = ns.T
    a = T.args
    a.new_h, a.new_w, a.dist, a.cluster_num, a.threshold, a.recon_size = H, W, 0, 4, 128, 256
    cfg = T.load_config(a.config)
    torch.manual_seed(1)
    model = ns.vgg.vgg16(pretrained=False, cfg=cfg['shared'])
    dis, dec, dis_patch = T.builder_gan(a)
    si.seeded_reinit(model, SEEDS['det'], 'det')
    si.seeded_reinit(dec, SEEDS['dec'], 'gan')
    si.seeded_reinit(dis, SEEDS['dis'], 'gan')
    si.seeded_reinit(dis_patch, SEEDS['dis_patch'], 'gan')
    mk = lambda m: torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr, betas=(0.9, 0.999), weight_decay=0.0001)  # noqa: E731
    s_det, s_dec, s_dis, s_patch = (_Sched(mk(m), lr) for m in (model, dec, dis, dis_patch))
    src, tgt = si.synth_images(SEEDS['images'], H, W)
    gts = si.synth_gts(G, SEEDS['gts'], H, W)
    info = torch.tensor([[H, W, 1.0]])
    train_loader = [(src, info, gts)]
    target_loader = [tgt]
    cap = _Capture()
    logging.getLogger('global').addHandler(cap)
    logging.getLogger('global').setLevel(logging.INFO)
    torch.manual_seed(SEEDS['torch'])
    np.random.seed(SEEDS['numpy'])
    t0 = time.time()
    T.train(train_loader, target_loader, None, model, dec, dis, dis_patch, s_det, s_dec, s_dis, s_patch, 1, cfg)
    dt = time.time() - t0
    logging.getLogger('global').removeHandler(cap)
    line = [l for l in cap.lines if l.startswith('Epoch:')][0]
    nums = dict(re.findall(r'(\w+):\s*([-0-9.eE]+)', line.replace('Loss:', 'loss:').replace('(', ' ')))
    out = {
        'H': np.int64(H), 'W': np.int64(W), 'G': np.int64(G), 'lr': np.float64(lr), 'seconds': np.float64(dt),
        'log_line': np.array(line),
        'logged': np.array([float(nums[k]) for k in ('loss', 'rpn_cls', 'rpn_loc', 'rpn_acc', 'rcnn_cls', 'rcnn_loc', 'rcnn_acc',
                                                     'fake_loss', 'dec_loss', 'dis_loss', 'fake_loss1')], dtype=np.float64),
        'logged_keys': np.array(['loss', 'rpn_cls', 'rpn_loc', 'rpn_acc', 'rcnn_cls', 'rcnn_loc', 'rcnn_acc', 'fake_loss_target',
                                 'recon_loss', 'adloss', 'fake_loss1_source']),
        'ck_det': si.checksum(model), 'ck_dec': si.checksum(dec), 'ck_dis': si.checksum(dis), 'ck_dis_patch': si.checksum(dis_patch),
    }
    # a few exactly-comparable tensors after the step
    sd = model.state_dict()
    out['det_fc_cls_bias'] = sd['fc_rcnn_cls.bias'].numpy().copy()
    out['det_conv1_1_bias'] = sd['features.0.bias'].numpy().copy()
    out['det_rpn_loc_bias'] = sd['rpn_head.conv_loc.bias'].numpy().copy()
    out['dis_last_bias'] = dis.state_dict()['model_A.3.bias'].numpy().copy()
    out['dec_last_bias_A'] = dec.state_dict()['decode_A.6.bias'].numpy().copy()
    dp = dis_patch.state_dict()
    out['dp_bn1_running_mean'] = dp['model_A_patch.0.model.1.running_mean'].numpy().copy()
    out['dp_bn1_running_var'] = dp['model_A_patch.0.model.1.running_var'].numpy().copy()
    out['dp_bn1_weight'] = dp['model_A_patch.0.model.1.weight'].numpy().copy()
    return out


def generate(ns, outdir):
    for (H, W) in ((256, 512), (512, 1024)):   # the second one is BASELINE.json's size (configs[1])
        out = run_reference_iteration(ns, H, W)
        np.savez_compressed(os.path.join(outdir, f"train_step_{H}x{W}.npz"), **out)
        print(f"train_step_{H}x{W}.npz written ({out['seconds']:.1f}s):", out['log_line'])
