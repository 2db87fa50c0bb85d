"""where do the oracle (CPU) and the product (MI355X) part ways at 512x1024?  compares proposals, cluster features, centres"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import model_common as mc
from scda_amd import layers as L
from scda_amd.train_step import ScdaTrainer
from test_train_step_gpu import build_product
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)
dev = torch.device("cuda:0")
ref, ref_models, masks = mc.oracle_iteration(H, W, lr=1e-3, record_masks=True)
ro = ref['_outputs']
torch.manual_seed(1)
tr = ScdaTrainer(mc.CFG, dev, lr=1e-3, new_w=W, new_h=H, models=mc.seeded_models(build_product))
src, tgt, gts, info = mc.seeded_inputs(H, W)
tape = list(masks)
tr.probe = mc.Probe(dropout_masks=lambda shape, p, device: tape.pop(0).to(device))
grabbed = {}
orig = tr.model.forward
def fwd(x, target):
    out = orig(x, target)
    grabbed.update(out)
    return out
tr.model.forward = fwd
np.random.seed(mc.SEEDS['numpy'])
out = tr.step(src.to(dev), gts, info, tgt.to(dev))
torch.cuda.synchronize()
tr.probe = None
po = grabbed
p_ref, p_got = ro['predict'][0].numpy(), po['predict'][0].cpu().numpy()
print("source proposals", p_ref.shape, p_got.shape, "boxes equal:", np.array_equal(p_ref[:, :5], p_got[:, :5]) if p_ref.shape == p_got.shape else None)
if p_ref.shape == p_got.shape:
    d = np.abs(p_ref[:, :5] - p_got[:, :5]).max(1)
    print("  rows differing > 1e-3:", int((d > 1e-3).sum()), " max score diff", float(np.abs(p_ref[:, 5] - p_got[:, 5]).max()))
for i, nm in enumerate(("source", "target")):
    a, b = ro['cluster_features'][i].detach().numpy(), po['cluster_features'][i].detach().cpu().numpy()
    print(nm, "cluster features", a.shape, "max abs diff", float(np.abs(a - b).max()), "rel L2", float(np.linalg.norm(a - b) / np.linalg.norm(a)))
    rows = np.abs(a - b).reshape(-1, a.shape[-1]).max(1)
    print("   rows with diff > 1e-3:", int((rows > 1e-3).sum()), "of", rows.size)
    print("   centres", np.asarray(ro['cluster_centers'][i]).round(3).tolist(), np.asarray(po['cluster_centers'][i]).round(3).tolist())
for k in ('rpn_cls', 'rpn_loc', 'rcnn_cls', 'rcnn_loc', 'adloss', 'dis_patch_loss', 'recon_loss', 'loss'):
    print(k, float(out[k]), float(ref[k]))
