"""RoIPool forward/backward at the training shape (512 RoIs, conv5 map 512x32x64): time per launch (HIP events)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scda_amd import native as N
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
R, C, H, W = 512, 512, 32, 64
w = np.exp(rs.uniform(np.log(16), np.log(400), R)); h = np.exp(rs.uniform(np.log(16), np.log(300), R))
x1 = rs.uniform(0, W * 16 - 1 - w); y1 = rs.uniform(0, H * 16 - 1 - h)
rois = torch.tensor(np.stack([np.zeros(R), x1, y1, x1 + w, y1 + h], 1).astype(np.float32), device=dev)
feat = torch.randn(1, C, H, W, device=dev)
out, arg = N.roi_pool_fwd(feat, rois, 7, 7, 1 / 16.)
dy = torch.randn_like(out)
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
print("roi_pool fwd %.1f us   bwd %.1f us" % (t(lambda: N.roi_pool_fwd(feat, rois, 7, 7, 1 / 16.)), t(lambda: N.roi_pool_bwd(dy, arg, rois, feat.shape, 7, 7, 1 / 16.))))
