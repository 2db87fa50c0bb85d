"""compare the captured per-phase gradients of one iteration with SCDA_AB_STREAMS on/off (run twice, then diff)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=512, new_h=256)
tr.capture = True
H, W = 256, 512
g = torch.Generator().manual_seed(5)
src = torch.randn(1, 3, H, W, generator=g).clamp_(-1, 1).to(dev); tgt = torch.randn(1, 3, H, W, generator=g).clamp_(-1, 1).to(dev)
gts = torch.tensor([[[30., 40., 200., 180., 3.], [250., 60., 400., 200., 5.]]]); info = torch.tensor([[H, W, 1.0]])
out = tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
res = {m + "/" + k: v.float().cpu().numpy() for m, d in tr.trace.items() for k, v in d.items()}
np.savez(sys.argv[1], **res)
print({k: float(v) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1})
