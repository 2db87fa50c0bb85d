"""RoIPool backward (ordered scatter) on [1,512,32,64] x 512 RoIs (us per call) for RoI populations of different sizes, and on the RoIs
a bench iteration really samples:  python scripts/micro/time_roipool_bwd.py [bench]
(the size classes are those of the round-6 experiment in DESIGN.md: RoIs whose bins are at least one / half a feature pixel -- the bins
that share a pixel then lie in a 2 x 2 / 3 x 3 block of bins -- and smaller ones)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scda_amd import native as N
dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters)
    return best * 1e3


def classes(rois, scale=1 / 16.):
    r = np.floor(np.abs(rois[:, 1:] * scale) + 0.5) * np.sign(rois[:, 1:])
    w = np.maximum(r[:, 2] - r[:, 0] + 1, 1); h = np.maximum(r[:, 3] - r[:, 1] + 1, 1)
    c2 = (w >= 7) & (h >= 7); c3 = ~c2 & (2 * w >= 7) & (2 * h >= 7)
    return c2.mean(), c3.mean(), 1 - c2.mean() - c3.mean()


def run(tag, feat, rois):
    shape = tuple(feat.shape)
    out, arg = N.roi_pool_fwd(feat, rois, 7, 7, 1 / 16.)
    top = torch.randn_like(out)
    fn = lambda: N.roi_pool_bwd(top, arg, rois, shape, 7, 7, 1 / 16.)
    t1 = timeit(fn)
    print("%-28s bins >= 1 px %.2f  >= 1/2 px %.2f  smaller %.2f | %.1f us" % ((tag,) + classes(rois.cpu().numpy()) + (t1,)), flush=True)


rs = np.random.RandomState(0)
feat = torch.randn(1, 512, 32, 64, device=dev).relu()
for lo, hi in ((16, 64), (16, 200), (64, 400), (112, 600)):
    R = 512
    w = rs.randint(lo, hi + 1, R); h = rs.randint(lo, hi + 1, R)
    x = rs.randint(0, 1024 - 16, R); y = rs.randint(0, 512 - 16, R)
    rois = np.stack([np.zeros(R), x, y, np.minimum(x + w, 1023), np.minimum(y + h, 511)], 1).astype(np.float32)
    run("sizes %d..%d px" % (lo, hi), feat, torch.from_numpy(rois).to(dev))
if "bench" in sys.argv:
    import bench
    from scda_amd.train_step import ScdaTrainer
    torch.manual_seed(0); np.random.seed(100)
    tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
    seen = []
    real = N.roi_pool_fwd
    N.roi_pool_fwd = lambda f, r, *a, **k: (seen.append((f.detach().clone(), r.detach().clone())), real(f, r, *a, **k))[1]
    src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
    for i in range(3): tr.step(src, gts, info, tgt)
    N.roi_pool_fwd = real
    for f, r in seen[-2:]:
        run("bench iteration, %d RoIs" % r.shape[0], f, r)
