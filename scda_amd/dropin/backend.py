"""Which implementation the host-side box logic calls for IoU and NMS.

Default: the HIP kernels (through the C ABI).  Tests and the CPU oracle may
plug the C oracle in with `use(...)` to exercise the host logic on a machine
without a GPU; the product never does.
"""
import os

import numpy as np
import torch

from scda_amd import native

_impl = {}
_aux = {}


def _aux_stream(dev):
    """High-priority stream for the box logic's small host->device->host round trips (IoU matrices, NMS).  Their
    inputs come from host memory, so they depend on nothing that is queued on the compute stream; issuing them there
    would make the host wait behind whatever the GPU is still working through (e.g. the two backbone passes)."""
    s = _aux.get(dev.index)
    if s is None:
        s = _aux[dev.index] = torch.cuda.Stream(device=dev, priority=-1)
    return s


def _pinned(a):
    t = torch.from_numpy(a)
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)
    return p


def _hip_bbox_overlaps(boxes, query):
    dev = torch.device("cuda", torch.cuda.current_device())
    aux = _aux_stream(dev)
    with torch.cuda.stream(aux):
        # pinned staging + async copies: a pageable hipMemcpy serialises with the compute stream
        b = _pinned(np.ascontiguousarray(boxes[:, :4], dtype=np.float32)).to(dev, non_blocking=True)
        q = _pinned(np.ascontiguousarray(query[:, :4], dtype=np.float32)).to(dev, non_blocking=True)
        out = native.bbox_overlaps(b, q)
        host = torch.empty(out.shape, dtype=torch.float32, pin_memory=True)
        host.copy_(out, non_blocking=True)
        aux.synchronize()
    return host.numpy().copy()


def _hip_nms(dets, thresh, max_keep=0):
    """dets: float tensor [N,5] sorted by score (any device) -> CPU LongTensor of kept indices."""
    dev = torch.device("cuda", torch.cuda.current_device())
    if dets.is_cuda:   # device-resident input: stay on the caller's stream (it may still be producing `dets`)
        d = dets.to(torch.float32).contiguous()
        keep, num = native.nms(d, float(thresh), max_keep)
        return keep[: int(num.item())].cpu().contiguous()
    aux = _aux_stream(dev)
    with torch.cuda.stream(aux):
        d = _pinned(np.ascontiguousarray(dets.numpy(), dtype=np.float32)).to(dev, non_blocking=True)
        keep, num = native.nms(d, float(thresh), max_keep)
        n = dets.shape[0]
        host = torch.empty(n + 1, dtype=torch.int64, pin_memory=True)   # [keep..., num] in one transfer
        host[:n].copy_(keep[:n], non_blocking=True)
        host[n:].copy_(num, non_blocking=True)
        aux.synchronize()
    return host[: int(host[n])].clone()


def _hip_nms_segments(lists, thresh):
    """lists: score-sorted float arrays [n_i, >=4] -> list of kept-index arrays, ONE upload, two launches, ONE download"""
    dev = torch.device("cuda", torch.cuda.current_device())
    lens = [int(a.shape[0]) for a in lists]
    rows, max_n = sum(lens), max(lens, default=0)
    if rows == 0:
        return [np.zeros(0, dtype=np.int64) for _ in lists]
    packed = np.zeros((rows, 5), dtype=np.float32)
    seg = np.zeros((len(lists), 3), dtype=np.int64)
    r = w = 0
    for i, (a, n) in enumerate(zip(lists, lens)):
        packed[r:r + n, :4] = a[:, :4]
        seg[i] = (r, n, w)
        r += n
        w += n * ((n + 63) // 64)
    aux = _aux_stream(dev)
    with torch.cuda.stream(aux):
        b = _pinned(packed).to(dev, non_blocking=True)
        sg = _pinned(seg).to(dev, non_blocking=True)
        keep, num = native.nms_segments(b, sg, max_n, float(thresh))
        host = torch.empty(rows + len(lists), dtype=torch.int64, pin_memory=True)
        host[:rows].copy_(keep[:rows], non_blocking=True)
        host[rows:].copy_(num[:len(lists)], non_blocking=True)
        aux.synchronize()
    h = host.numpy()
    return [h[o:o + int(h[rows + i])].copy() for i, (o, n) in enumerate(zip(seg[:, 0], lens))]


def nms_segments(lists, thresh):
    """NMS of several independent score-sorted lists at once (functions/predict_bbox.py:29-55 calls nms per class and image).  With a
    substituted nms hook (tests on the CPU): one call of it per list."""
    f = _impl.get("nms")
    if f is None:
        if os.environ.get("SCDA_NMS_UNBATCHED"):       # A/B knob (scripts/time_eval.py): one round trip per list, as before
            return [_hip_nms(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)), thresh).numpy() for a in lists]
        return _hip_nms_segments(lists, thresh)
    return [np.asarray(f(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)), thresh)) if a.shape[0] else np.zeros(0, dtype=np.int64)
            for a in lists]


def host_array(x, copy=False):
    """numpy view of `x` on the host.  Device tensors that were built from host data carry their host original along as
    `_scda_host` (ground-truth boxes, sampled RoIs): reading that costs nothing, whereas `.cpu()` is a synchronous copy on
    the compute stream -- the host would sit behind every kernel already queued (both backbone passes, ~8 ms)."""
    if x is None:
        return None
    host = getattr(x, "_scda_host", None)
    if host is not None:
        return host.copy() if copy else host
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    return np.array(x) if copy else x


def use(bbox_overlaps=None, nms=None):
    if bbox_overlaps is not None:
        _impl["bbox_overlaps"] = bbox_overlaps
    if nms is not None:
        _impl["nms"] = nms


def reset():
    _impl.clear()


def bbox_overlaps(boxes, query):
    return _impl.get("bbox_overlaps", _hip_bbox_overlaps)(boxes, query)


def nms(dets, thresh, max_keep=0):
    f = _impl.get("nms")
    if f is None:
        return _hip_nms(dets, thresh, max_keep)
    k = f(dets, thresh)
    return k[:max_keep] if max_keep and max_keep > 0 else k
