import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from scda_amd import native as N
dev = torch.device("cuda:0")
def err(a, b): return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12)).item()
for (C, H, W, act) in [(512, 224, 7, 1), (2048, 224, 7, 0), (512, 224, 7, 0), (16, 50, 84, 1), (64, 3584, 7, 1)]:
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(1, C, H, W, generator=g) * 1.3 + 0.2).to(dev); dy = torch.randn(1, C, H, W, generator=g).to(dev)
    ga = (1 + 0.1 * torch.randn(C, generator=g)).to(dev); be = (0.1 * torch.randn(C, generator=g)).to(dev)
    res = {}
    for name in ("plane", "old"):
        if name == "old": os.environ["SCDA_BN_NO_PLANE"] = "1"
        else: os.environ.pop("SCDA_BN_NO_PLANE", None)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y, mean, rstd = N.batchnorm_fwd(x, ga, be, rm, rv, 1e-5, 0.1, act, 0.0)
        dx, dg, db = N.batchnorm_bwd(dy, x, ga, be, mean, rstd, act, 0.0)
        torch.cuda.synchronize()
        res[name] = dict(y=y, mean=mean, rstd=rstd, dx=dx, dg=dg, db=db, rm=rm, rv=rv)
    print((C, H, W, act), {k: "%.1e" % err(res["plane"][k], res["old"][k]) for k in res["old"]})
