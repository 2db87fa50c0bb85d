"""Deterministic, construction-order-independent weight initialisation and synthetic inputs shared by
make_golden_model.py (applied to the REFERENCE's modules), the CPU oracle tests and the GPU parity tests.
Values depend only on (state_dict key order, shapes, seed), so any two implementations with the same
state_dict layout get bit-identical weights."""
import math

import numpy as np
import torch


def seeded_reinit(module, seed, kind):
    """kind: 'det' (He-style conv, 0.01 linear) or 'gan' (0.02 conv)"""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    with torch.no_grad():
        for k in sorted(sd.keys()):
            t = sd[k]
            if not t.dtype.is_floating_point:
                continue
            r = torch.randn(t.shape, generator=g)
            if k.endswith("running_mean"):
                t.copy_(0.1 * r)
            elif k.endswith("running_var"):
                t.copy_(1.0 + 0.1 * r.abs())
            elif t.dim() == 4:
                if kind == "det":
                    t.copy_(r * math.sqrt(2.0 / (t.shape[2] * t.shape[3] * t.shape[0])))
                else:
                    t.copy_(r * 0.02)
            elif t.dim() == 2:
                t.copy_(r * 0.01)
            elif k.endswith("weight"):      # BatchNorm gamma
                t.copy_(1.0 + 0.1 * r)
            else:                            # biases, BatchNorm beta
                t.copy_(0.01 * r)


def synth_gts(G, seed, H=512, W=1024):
    r = np.random.RandomState(seed)
    w = np.exp(r.uniform(np.log(16), np.log(min(400, W / 2)), G))
    h = np.exp(r.uniform(np.log(16), np.log(min(300, H / 2)), G))
    x1 = r.uniform(0, W - 1 - w)
    y1 = r.uniform(0, H - 1 - h)
    box = np.stack([np.floor(x1), np.floor(y1), np.minimum(np.ceil(x1 + w), W - 1), np.minimum(np.ceil(y1 + h), H - 1)], 1)
    cls = r.randint(1, 9, G)
    return torch.from_numpy(np.concatenate([box, cls[:, None]], 1).astype(np.float32)[None])


def synth_images(seed, H=512, W=1024):
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(1, 3, H, W, generator=g).clamp_(-1, 1)
    tgt = torch.randn(1, 3, H, W, generator=g).clamp_(-1, 1)
    return src, tgt


def checksum(module):
    """order-independent fp64 fingerprints of a module's floating-point state"""
    s = a = q = 0.0
    for k, t in sorted(module.state_dict().items()):
        if t.dtype.is_floating_point:
            d = t.detach().double().cpu()
            s += float(d.sum()); a += float(d.abs().sum()); q += float((d * d).sum())
    return np.array([s, a, q], dtype=np.float64)
