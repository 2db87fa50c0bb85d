"""compare activations / activation-gradients of selected backbone layers, product (GPU) vs oracle (CPU)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import model_common as mc
from test_train_step_gpu import build_product
from scda_amd import layers as L
from scda_amd.train_step import ScdaTrainer
from oracle import torch_ref as R

cuda = torch.device("cuda:0")
H, W, lr = 256, 512, 1e-3
# post-activation outputs to watch: (oracle module index, product module index)
WATCH = {"conv3_3": (15, 14), "pool3": (16, 16), "conv4_1": (18, 17), "conv4_2": (20, 19), "conv4_3": (22, 21), "conv5_3": (29, 28)}
cap = {"o": {}, "p": {}}

def hook(side, name):
    def f(mod, inp, out):
        if name + "_act" in cap[side] or not out.requires_grad:
            return
        cap[side][name + "_act"] = out.detach().cpu().clone()
        out.register_hook(lambda g: cap[side].__setitem__(name + "_grad", g.detach().cpu().clone()))
    return f

# oracle
R.use_cpu_backend()
torch.manual_seed(1)
omodels = mc.seeded_models(lambda: R.build_models(mc.CFG))
for n, (oi, pi) in WATCH.items():
    omodels[0].features[oi].register_forward_hook(hook("o", n))
otr = R.RefTrainer(mc.CFG, omodels, lr=lr, new_w=W, new_h=H)
src, tgt, gts, info = mc.seeded_inputs(H, W)
R.RecordingDropout.tape = []
torch.manual_seed(mc.SEEDS['torch']); np.random.seed(mc.SEEDS['numpy'])
otr.step(src, gts, info, tgt)
masks = R.RecordingDropout.tape; R.RecordingDropout.tape = None
R.reset_backend()
# product
torch.manual_seed(1)
pmodels = mc.seeded_models(build_product)
for n, (oi, pi) in WATCH.items():
    pmodels[0].features[pi].register_forward_hook(hook("p", n))
ptr = ScdaTrainer(mc.CFG, cuda, lr=lr, new_w=W, new_h=H, models=pmodels)
tape = list(masks)
ptr.probe = mc.Probe(dropout_masks=lambda shape, p, device: tape.pop(0).to(device))
np.random.seed(mc.SEEDS['numpy'])
ptr.step(src.to(cuda), gts, info, tgt.to(cuda))
for n in WATCH:
    for kind in ("act", "grad"):
        a, b = cap["p"][n + "_" + kind].double(), cap["o"][n + "_" + kind].double()
        d = (a - b).abs()
        nz_mismatch = int(((a != 0) != (b != 0)).sum())
        big = int((d > 1e-3 * b.abs().max()).sum())
        print("%-8s %-5s max|ref| %.3e  max|diff| %.3e  rel %.2e  nonzero-pattern mismatches %d  elems>1e-3*max %d / %d" %
              (n, kind, b.abs().max(), d.max(), d.max() / b.abs().max(), nz_mismatch, big, a.numel()))
