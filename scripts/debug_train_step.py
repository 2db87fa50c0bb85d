"""per-tensor gradient comparison product (GPU) vs oracle (CPU) for one iteration"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import model_common as mc
from test_train_step_gpu import build_product, rel
from scda_amd import layers as L
from scda_amd.train_step import ScdaTrainer

cuda = torch.device("cuda:0")
H, W, lr = 256, 512, 1e-3
ref, ref_models, masks = mc.oracle_iteration(H, W, lr=lr, record_masks=True, capture=True)
torch.manual_seed(1)
models = mc.seeded_models(build_product)
tr = ScdaTrainer(mc.CFG, cuda, lr=lr, new_w=W, new_h=H, models=models); tr.capture = True
src, tgt, gts, info = mc.seeded_inputs(H, W)
tape = list(masks)
print("mask shapes:", [tuple(m.shape) for m in masks])
tr.probe = mc.Probe(dropout_masks=lambda shape, p, device: tape.pop(0).to(device))
np.random.seed(mc.SEEDS['numpy'])
out = tr.step(src.to(cuda), gts, info, tgt.to(cuda))
for name in ('dis', 'dis_patch', 'dec', 'det'):
    rg, pg = ref['_trace'][name], tr.trace[name]
    rows = sorted(((rel(pg[k], rg[k]), k, float(rg[k].abs().max()), float(pg[k].abs().max())) for k in rg), reverse=True)
    print("==", name)
    for r in rows[:40]:
        print("  %.3e  %-40s ref|max| %.3e  got|max| %.3e" % r)
