import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bench
from scda_amd.dropin.models.mask_rcnn.resnet import resnet50
from test_oracle_golden import rand_rois
cuda = torch.device("cuda:0")
torch.manual_seed(7)
det = resnet50(cfg=dict(bench.CFG['shared'], roi_align=True, gan_model_flag=2)).to(cuda).train()
rs = np.random.RandomState(9)
feat = (torch.randn(1, 1024, 24, 40, generator=torch.Generator().manual_seed(8)) * 0.5).to(cuda)
rois = torch.from_numpy(rand_rois(rs, 32, B=1, W=40 * 16, H=24 * 16)).to(cuda)
state = {k: v.clone() for k, v in det.state_dict().items()}
res = {}
for name in ("tall", "nchw", "tall_noplane"):
    det.load_state_dict(state); det.zero_grad(set_to_none=True)
    os.environ.pop("SCDA_RESNET_HEAD_NCHW", None); os.environ.pop("SCDA_BN_NO_PLANE", None)
    if name == "nchw": os.environ["SCDA_RESNET_HEAD_NCHW"] = "1"
    if name == "tall_noplane": os.environ["SCDA_BN_NO_PLANE"] = "1"
    f = feat.clone().requires_grad_()
    x_fea, cls, loc = det.rcnn(f, rois)
    g = torch.Generator().manual_seed(10)
    loss = (x_fea * torch.randn(x_fea.shape, generator=g).to(cuda)).sum() + (cls * torch.randn(cls.shape, generator=g).to(cuda)).sum() + (loc * torch.randn(loc.shape, generator=g).to(cuda)).sum()
    loss.backward()
    res[name] = dict(x_fea=x_fea.detach(), cls=cls.detach(), loc=loc.detach(), dfeat=f.grad, **{"g:" + k: p.grad.clone() for k, p in det.named_parameters() if p.grad is not None})
def err(a, b):
    a = a.double(); b = b.double(); return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
for other in ("tall", "tall_noplane"):
    worst = sorted(((err(res[other][k], res["nchw"][k]), k) for k in res["nchw"]), reverse=True)[:8]
    print(other, "vs nchw:", [(round(e, 7), k) for e, k in worst])
