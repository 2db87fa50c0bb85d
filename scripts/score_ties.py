"""How often do two RPN scores of one image coincide in fp32 inside the pre-NMS ranking (ranks 0 .. pre_nms_top_n)?  A ranking made on
the device reproduces the reference's np.argpartition / np.argsort only where no two equal scores meet (their tie order is numpy's);
this counts, over the bench's iterations, the images for which that holds."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd import device_boxes
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
stats = []
orig = device_boxes.rpn_proposals
def counting(prob_dev, loc_dev, cfg, image_info, scores_host=None, loc_host=None):
    sh = scores_host if scores_host is not None else prob_dev.detach().cpu()
    B, A2, fh, fw = sh.shape
    s = sh.permute(0, 2, 3, 1).contiguous().view(B, fh * fw * (A2 // 2), -1).numpy()[0, :, -1]
    top = cfg['pre_nms_top_n']
    srt = np.sort(s)[::-1][:top + 1]
    stats.append((int((srt[1:] == srt[:-1]).sum()), float(srt[0]), float(srt[-1]), int(np.unique(s).size)))
    return orig(prob_dev, loc_dev, cfg, image_info, scores_host, loc_host)
device_boxes.rpn_proposals = counting
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 45):
    tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
ties = np.array([t[0] for t in stats])
print("images ranked: %d; without a tie inside ranks 0..top_n: %d; ties per image: median %d, min %d, max %d" %
      (len(stats), int((ties == 0).sum()), int(np.median(ties)), ties.min(), ties.max()))
for i in (0, 1, len(stats) // 2, len(stats) - 2, len(stats) - 1):
    print("  image %3d: %4d equal neighbours among the top %d, scores %.4f .. %.4f, %d distinct values of 30720" % (i, stats[i][0], bench.CFG['train_rpn_proposal_cfg']['pre_nms_top_n'], stats[i][1], stats[i][2], stats[i][3]))
