"""Cluster-region generator -- the contract of functions/mask.py:183-237: k-means over RoI centres, then per cluster
the first `threshold` member RoIs (or a with-replacement resample when the cluster is smaller) are stacked into
[N_cluster, threshold, F].  The k-means is sklearn's (third-party, as in the reference: KMeans(n_clusters,
random_state=0)); what changes here is the data path: the RoI features never leave the MI355X -- only the 512 RoI
boxes (already on the host) feed the clustering, and the gather runs on the device.

As in the reference the result is a NEW leaf tensor: no gradient flows back into the detector through it
(functions/mask.py:202-203,234 round-trip through numpy)."""
import numpy as np
import torch

from scda_amd.dropin import backend


def _np(x):
    return backend.host_array(x)


def proposals_to_centers(proposals):
    """[N,>=5] (b,x1,y1,x2,y2) -> [N,2] (cx, cy)"""
    return np.stack([(proposals[:, 3] + proposals[:, 1]) / 2.0, (proposals[:, 4] + proposals[:, 2]) / 2.0], axis=1)


_POOLS_LIMITED = []


def _limit_host_pools_once():
    """Pin the BLAS / OpenMP pools that numpy, scipy and sklearn bring along to ONE thread, once, for the life of the
    process (torch's own pool is left alone).  The host-side work of the SCDA step is 512-point k-means, 12 000-element
    sorts and a few small reductions: on a 256-thread host the default pools only hurt -- waking or re-spawning 255
    worker threads costs 30-60 ms per parallel region (measured with scripts/stall_sampler.py: stalls inside
    sklearn's Lloyd loop, np.einsum, np.var and threadpoolctl's own set_num_threads), and their spin-waiting preempts the
    thread that feeds the GPU.  Results do not depend on the thread count except for the last bits of the float32
    k-means centres (see tests/test_host_functions.py)."""
    if _POOLS_LIMITED:
        return
    import os
    if os.environ.get('SCDA_NO_POOL_LIMIT'):
        _POOLS_LIMITED.append(None)
        return
    try:
        from threadpoolctl import ThreadpoolController
        ctl = ThreadpoolController()
        ctl.lib_controllers = [c for c in ctl.lib_controllers if "torch" not in (c.filepath or "")]
        _POOLS_LIMITED.append(ctl.limit(limits=1))   # kept alive, never restored
    except Exception:  # threadpoolctl missing: correctness is unaffected
        _POOLS_LIMITED.append(None)


def cluster_indices(proposals_np, N_cluster=4, threshold=128):
    """-> (index int64 [N_cluster, threshold] into the RoI list, centres float64 [N_cluster, 2])"""
    from sklearn.cluster import KMeans
    _limit_host_pools_once()   # 512 two-dimensional points: one thread
    km = KMeans(n_clusters=N_cluster, random_state=0).fit(proposals_to_centers(proposals_np))
    rows = []
    for c in range(N_cluster):
        member = np.where(km.labels_[:] == c)[0]
        if member.shape[0] < threshold:
            member = member[np.random.choice(member.shape[0], threshold, replace=True)]
        else:
            member = member[0:threshold]
        rows.append(member)
    return np.stack(rows, axis=0).astype(np.int64), km.cluster_centers_


def compute_cluster_targets(proposals, features, N_cluster=4, threshold=128):
    """proposals [N,>=5], features [N,F] -> (cluster features [N_cluster, threshold, F] (leaf), centres [N_cluster,2])"""
    idx, centres = cluster_indices(_np(proposals), N_cluster, threshold)
    if features.is_cuda:
        from scda_amd import native
        flat = native.upload(idx.reshape(-1), features.device)
    else:
        flat = torch.from_numpy(idx.reshape(-1))
    with torch.no_grad():
        gathered = features.detach().index_select(0, flat).view(N_cluster, threshold, features.shape[1]).contiguous()
    return gathered.float(), centres


# ---------------------------------------------------------------------------------------------------------------------------
# Mask targets (functions/mask.py:21-179) -- BASELINE.json configs[4], the mask branch of models/mask_rcnn/resnet.py:146-149.
# Host numpy like the reference.  `cv2.resize` is third-party and absent from this image: `resize_linear_u8` restates OpenCV's
# published fixed-point INTER_LINEAR for 8-bit images (parity unpinned: no cv2 here to check it against).
# ---------------------------------------------------------------------------------------------------------------------------
def _linear_taps(src, dst):
    """OpenCV's INTER_LINEAR taps for one axis: indices [dst] / [dst] and 11-bit fixed-point weights [dst, 2] (int32)"""
    scale = src / float(dst)
    f = (np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5
    s = np.floor(f).astype(np.int64)
    f = (f - s).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    w = np.stack([np.rint((1.0 - f) * 2048.0), np.rint(f * 2048.0)], axis=1).astype(np.int32)   # saturate_cast<short>(cvRound)
    return s, np.minimum(s + 1, src - 1), w


def resize_linear_u8(img, dst_w, dst_h):
    """img [h, w] uint8 -> [dst_h, dst_w] uint8: cv2.resize(img, (dst_w, dst_h)) with the default INTER_LINEAR"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    x0, x1, a = _linear_taps(w, dst_w)
    y0, y1, b = _linear_taps(h, dst_h)
    a0, a1 = a[None, :, 0], a[None, :, 1]
    top, bot = img[y0].astype(np.int32), img[y1].astype(np.int32)                  # only the source rows some output row reads
    r0 = top[:, x0] * a0 + top[:, x1] * a1                                         # horizontal pass, 11 fractional bits
    r1 = bot[:, x0] * a0 + bot[:, x1] * a1
    out = (((b[:, 0, None] * (r0 >> 4)) >> 16) + ((b[:, 1, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def _linear_taps_batch(src, dst):
    """_linear_taps for N source lengths at once: src [N] -> indices [N, dst] x 2 (relative to the window), weights [N, dst, 2]"""
    src = src.astype(np.int64)[:, None]
    f = (np.arange(dst, dtype=np.float64)[None, :] + 0.5) * (src / float(dst)) - 0.5
    s = np.floor(f).astype(np.int64)
    f = (f - s).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s = np.where(lo, 0, s)
    hi = s >= src - 1
    f[hi] = 0.0
    s = np.where(hi, src - 1, s)
    w = np.stack([np.rint((1.0 - f) * 2048.0), np.rint(f * 2048.0)], axis=2).astype(np.int32)
    return s, np.minimum(s + 1, src - 1), w


def generate_mask_labels(rois, masks, mask_h, mask_w, index=None):
    """rois [N, >=4] (x1, y1, x2, y2), masks [N, H, W] binary -> [N, mask_h, mask_w] int32: each RoI's window of its mask,
    resized (functions/mask.py:51-71).  index [N]: RoI i reads masks[index[i]] -- the caller's gather `masks[pos_g_ix]` (:135) without
    materialising one full-image plane per RoI (64 RoIs x 800 x 1344 bytes per iteration).  All RoIs in one pass: the four source
    pixels of every output pixel are gathered straight from the planes (`resize_linear_u8` on each window, vectorised over RoIs)."""
    rois = rois.astype(np.int32)
    n = rois.shape[0]
    assert n == (masks.shape[0] if index is None else len(index))
    x1, y1, x2, y2 = (rois[:, c].astype(np.int64) for c in range(4))
    assert (x1 < x2).all() and (y1 < y2).all()
    if masks.dtype != np.uint8:
        masks = masks.astype(np.uint8)          # cv2.resize of an 8-bit window
    plane = (np.arange(n) if index is None else np.asarray(index, dtype=np.int64))[:, None, None]
    xa, xb, a = _linear_taps_batch(x2 - x1, mask_w)
    ya, yb, b = _linear_taps_batch(y2 - y1, mask_h)
    xa, xb = (xa + x1[:, None])[:, None, :], (xb + x1[:, None])[:, None, :]
    ya, yb = (ya + y1[:, None])[:, :, None], (yb + y1[:, None])[:, :, None]
    a0, a1 = a[:, None, :, 0], a[:, None, :, 1]
    r0 = masks[plane, ya, xa].astype(np.int32) * a0 + masks[plane, ya, xb].astype(np.int32) * a1
    r1 = masks[plane, yb, xa].astype(np.int32) * a0 + masks[plane, yb, xb].astype(np.int32) * a1
    out = (((b[:, :, 0, None] * (r0 >> 4)) >> 16) + ((b[:, :, 1, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.int32)


def compute_mask_targets(proposals, cfg, ground_truth_bboxes, ground_truth_masks, image_info, ignore_regions=None):
    """proposals [N, >=5] (b, x1, y1, x2, y2, ...), gts [B, G, >=5], gt masks [B, G, H, W] ->
    rois [R, 6] fp32 (b, x1, y1, x2, y2, class) and labels [R, num_classes, label_h, label_w] fp32 (-1 = ignore; the RoI's own
    class plane holds its 0/1 mask).  functions/mask.py:73-179: RoIs with IoU > positive_iou_thresh, integer-clipped, at most
    batch_size_per_image per image (np.random.choice without replacement -- the only RNG draw)."""
    from scda_amd.dropin.utils import bbox_helper
    dev = proposals.device if torch.is_tensor(proposals) else torch.device("cpu")
    proposals, gts_all, masks_all, image_info = (np.asarray(_np(v)) for v in (proposals, ground_truth_bboxes, ground_truth_masks, image_info))
    batch_rois, batch_labels = [], []
    for b_ix in range(gts_all.shape[0]):
        rois = proposals[proposals[:, 0] == b_ix][:, 1:5]
        gts, masks = gts_all[b_ix], masks_all[b_ix]
        keep = np.where(gts[:, 2] > gts[:, 1] + 1)[0]              # the reference's padded-gt filter (:108), as written
        if keep.size == 0:
            continue
        gts = gts[keep]                                            # (the mask planes stay where they are: `keep` indexes them)
        if cfg['append_gts']:
            rois = np.vstack([rois, gts[:, :4]])
        rois = bbox_helper.clip_bbox(rois.astype(np.int32), image_info[b_ix].astype(np.int32))
        if rois.shape[0] == 0 or gts.shape[0] == 0:
            continue
        overlaps = bbox_helper.bbox_iou_overlaps(rois, gts)
        arg, best = overlaps.argmax(axis=1), overlaps.max(axis=1)
        pos_r = np.where(best > cfg['positive_iou_thresh'])[0]
        pos_g = arg[pos_r]
        if pos_r.shape[0] == 0:
            continue
        if 0 < cfg['batch_size_per_image'] < pos_r.shape[0]:
            pick = np.random.choice(pos_r.shape[0], size=cfg['batch_size_per_image'], replace=False)
            pos_r, pos_g = pos_r[pick], pos_g[pick]
        pos_rois = rois[pos_r]
        classes = gts[pos_g][:, 4].astype(np.int64)
        n = pos_rois.shape[0]
        labels = -np.ones((n, cfg['num_classes'], cfg['label_h'], cfg['label_w']))
        labels[range(n), classes, ...] = generate_mask_labels(pos_rois, masks, cfg['label_h'], cfg['label_w'], index=keep[pos_g])
        batch_rois.append(np.hstack([np.full((n, 1), b_ix), pos_rois, classes[:, None]]))
        batch_labels.append(labels)
    if not batch_rois:                                             # no positive RoI at all: one all-ignore row (:150-153)
        rois_out = np.zeros((1, 5), dtype=np.float32)
        labels_out = -np.ones((1, cfg['num_classes'], cfg['label_h'], cfg['label_w']), dtype=np.float32)
    else:
        rois_out, labels_out = np.vstack(batch_rois), np.vstack(batch_labels)
    # (float32 on the numpy side: the same values as the reference's `.float()`, without waking torch's intra-op pool)
    return (torch.from_numpy(np.ascontiguousarray(rois_out, dtype=np.float32)).to(dev),
            torch.from_numpy(np.ascontiguousarray(labels_out, dtype=np.float32)).to(dev))


def predict_masks(rois, heatmap, image_info, cfg=None):
    """rois [R, >=7] (b, x1, y1, x2, y2, score, class), heatmap [R, num_classes, h, w] -> list of [image_h, image_w] fp32 maps:
    each RoI's class plane resized to the RoI (PIL bilinear, as functions/mask.py:21-49) and pasted at its place"""
    from PIL import Image
    rois, heatmap, image_info = (np.asarray(_np(v)) for v in (rois, heatmap, image_info))
    assert rois.shape[0] == heatmap.shape[0]
    out = []
    for r_ix in range(rois.shape[0]):
        b_ix, x1, y1, x2, y2, _, cls = map(int, rois[r_ix][:7])
        roi_w, roi_h = x2 - x1 + 1, y2 - y1 + 1
        image_h, image_w = map(int, image_info[b_ix][:2])
        plane = np.array(Image.fromarray(np.ascontiguousarray(heatmap[r_ix, cls], dtype=np.float32)).resize((roi_w, roi_h)))
        image = np.zeros((image_h, image_w), dtype=np.float32)
        image[y1:y1 + roi_h, x1:x1 + roi_w] = plane
        out.append(image)
    return out
