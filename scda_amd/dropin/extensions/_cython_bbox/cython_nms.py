"""cython_nms.nms / soft_nms API (extensions/_cython_bbox/cython_nms.pyx:37-203).  Nothing on the SCDA path
calls them (the GPU nms is used everywhere); kept for import compatibility as plain host loops over <= a few
hundred boxes."""
import numpy as np

f32, f64 = np.float32, np.float64


def nms(dets, thresh):
    """greedy NMS, IoU(+1) >= thresh suppresses; returns np.where(suppressed == 0)[0] like the reference"""
    d = np.asarray(dets, dtype=np.float32)
    x1, y1, x2, y2, s = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = s.argsort()[::-1]
    dead = np.zeros(len(d), dtype=np.int64)
    for a in range(len(d)):
        i = order[a]
        if dead[i]:
            continue
        rest = order[a + 1:]
        rest = rest[dead[rest] == 0]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        dead[rest[ovr >= np.float32(thresh)]] = 1
    return np.where(dead == 0)[0]


def soft_nms(boxes_in, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    boxes = np.array(boxes_in, dtype=np.float32, copy=True)
    n = boxes.shape[0]
    inds = np.arange(n)
    i = 0
    while i < n:
        m = i + int(np.argmax(boxes[i:n, 4]))
        boxes[[i, m]] = boxes[[m, i]]
        inds[[i, m]] = inds[[m, i]]
        tx1, ty1, tx2, ty2 = boxes[i, :4]
        pos = i + 1
        while pos < n:
            x1, y1, x2, y2 = boxes[pos, :4]
            # Cython writes the source's "+ 1" next to a C float as "+ 1.0": those sums and the products around them are evaluated
            # in double and rounded once on assignment to the `cdef float` (cython_nms.pyx:163-170)
            iw = f32(f64(min(tx2, x2) - max(tx1, x1)) + 1.0)
            ih = f32(f64(min(ty2, y2) - max(ty1, y1)) + 1.0)
            if iw > 0 and ih > 0:
                area = f32((f64(x2 - x1) + 1.0) * (f64(y2 - y1) + 1.0))
                ua = f32((f64(tx2 - tx1) + 1.0) * (f64(ty2 - ty1) + 1.0) + f64(area) - f64(iw * ih))
                ov = (iw * ih) / ua
                if method == 1:
                    wgt = 1 - ov if ov > Nt else 1
                elif method == 2:
                    # float32 argument, double exp, rounded to the reference's `cdef float weight` (cython_nms.pyx:179)
                    wgt = np.float32(np.exp(np.float64(np.float32(-(ov * ov)) / np.float32(sigma))))
                else:
                    wgt = 0 if ov > Nt else 1
                boxes[pos, 4] *= wgt
                if boxes[pos, 4] < threshold:
                    boxes[pos] = boxes[n - 1]
                    inds[pos] = inds[n - 1]
                    n -= 1
                    pos -= 1
            pos += 1
        i += 1
    return boxes[:n], inds[:n]
