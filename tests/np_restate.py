"""Independent numpy restatements used to cross-check the C oracle (second statement of
each operator, written from the operator's definition, not from scda_oracle.c)."""
import numpy as np

f32 = np.float32


def iou_plus1(a, b):
    w = max(f32(min(a[2], b[2]) - max(a[0], b[0])) + f32(1), f32(0))
    h = max(f32(min(a[3], b[3]) - max(a[1], b[1])) + f32(1), f32(0))
    inter = f32(w * h)
    sa = f32(f32(a[2] - a[0] + f32(1)) * f32(a[3] - a[1] + f32(1)))
    sb = f32(f32(b[2] - b[0] + f32(1)) * f32(b[3] - b[1] + f32(1)))
    return f32(inter / f32(f32(sa + sb) - inter))


def greedy_nms(boxes, thresh):
    """boxes sorted by score; suppress j>i when IoU(+1) > thresh (GPU semantics)."""
    boxes = boxes.astype(f32)
    n = len(boxes)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = ((x2 - x1 + f32(1)) * (y2 - y1 + f32(1))).astype(f32)
    dead = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        j = np.arange(i + 1, n)
        w = np.maximum((np.minimum(x2[i], x2[j]) - np.maximum(x1[i], x1[j])).astype(f32) + f32(1), f32(0)).astype(f32)
        h = np.maximum((np.minimum(y2[i], y2[j]) - np.maximum(y1[i], y1[j])).astype(f32) + f32(1), f32(0)).astype(f32)
        inter = (w * h).astype(f32)
        iou = (inter / ((area[i] + area[j]).astype(f32) - inter).astype(f32)).astype(f32)
        dead[j[iou > f32(thresh)]] = True
    return np.array(keep, dtype=np.int64)


def round_half_away(x):
    return np.where(x >= 0, np.floor(x + f32(0.5)), np.ceil(x - f32(0.5))).astype(np.int64)


def roi_pool(feat, rois, PH, PW, scale):
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, PH, PW), f32)
    arg = np.full((R, C, PH, PW), -1, np.int32)
    for n in range(R):
        b = int(rois[n, 0])
        sw, sh, ew, eh = [int(v) for v in round_half_away((rois[n, 1:5].astype(f32) * f32(scale)).astype(f32))]
        rw, rh = max(ew - sw + 1, 1), max(eh - sh + 1, 1)
        bh, bw = f32(rh) / f32(PH), f32(rw) / f32(PW)
        for ph in range(PH):
            hs = min(max(int(np.floor(f32(ph) * bh)) + sh, 0), H)
            he = min(max(int(np.ceil(f32(ph + 1) * bh)) + sh, 0), H)
            for pw in range(PW):
                ws = min(max(int(np.floor(f32(pw) * bw)) + sw, 0), W)
                we = min(max(int(np.ceil(f32(pw + 1) * bw)) + sw, 0), W)
                if he <= hs or we <= ws:
                    continue
                win = feat[b, :, hs:he, ws:we].reshape(C, -1)
                k = win.argmax(1)  # first max in row-major scan order
                out[n, :, ph, pw] = win[np.arange(C), k]
                hh, ww = hs + k // (we - ws), ws + k % (we - ws)
                arg[n, :, ph, pw] = ((b * C + np.arange(C)) * H + hh) * W + ww
    return out, arg


def roi_pool_bwd_scatter(top, arg, feat_shape):
    """scatter in (roi, c, ph, pw) order == the reference's per-element (roi, ph, pw) order"""
    g = np.zeros(int(np.prod(feat_shape)), f32)
    a = arg.reshape(-1)
    t = top.reshape(-1).astype(f32)
    # sequential fp32 accumulation in index order
    for i in np.nonzero(a >= 0)[0]:
        g[a[i]] = f32(g[a[i]] + t[i])
    return g.reshape(feat_shape)
