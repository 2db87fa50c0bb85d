"""Seeded NMS inputs shared by tests/golden/make_golden_nms.py (run under the image's python3.9, where the reference's
cython_nms.pyx builds) and by the tests that compare the oracle and the HIP kernels with the keep lists it produced.
Only RandomState.uniform / randint and IEEE +,-,*,/ , round, sort: the arrays are bit-identical under numpy 1.26 and 2.2
(the fixture stores a sha256 of every input, checked before use)."""
import hashlib

import numpy as np

W, H = 1024, 512

# (name, n, kind, thresh): sizes of the RPN path (pre-NMS 6000 test / 12000 train), the class-wise NMS of evaluation (300) and the
# 2000 that survive; thresholds of cfg rpn (0.7), evaluation (0.5 / 0.3)
CASES = [
    ("rpn_300_t07", 300, "rpn", 0.7),
    ("rpn_2000_t07", 2000, "rpn", 0.7),
    ("rpn_6000_t07", 6000, "rpn", 0.7),
    ("rpn_12000_t07", 12000, "rpn", 0.7),
    ("clustered_2000_t05", 2000, "clustered", 0.5),
    ("clustered_12000_t07", 12000, "clustered", 0.7),
    ("integer_6000_t03", 6000, "integer", 0.3),
    ("integer_700_t05", 700, "integer", 0.5),
    ("sparse_65_t07", 65, "rpn", 0.7),
    ("single_1_t07", 1, "rpn", 0.7),
]


def _rpn_like(rs, n):
    """what decode+clip of 9 anchors per cell looks like: centres on a stride-16 grid with regression jitter, three aspect
    ratios x three scales with jitter, clipped to the image"""
    cx = rs.randint(0, W // 16, n) * 16.0 + 8.0 + rs.uniform(-12, 12, n)
    cy = rs.randint(0, H // 16, n) * 16.0 + 8.0 + rs.uniform(-12, 12, n)
    scale = np.array([64.0, 128.0, 256.0, 512.0])[rs.randint(0, 4, n)] * rs.uniform(0.6, 1.5, n)
    ratio = np.array([0.5, 1.0, 2.0])[rs.randint(0, 3, n)] * rs.uniform(0.8, 1.25, n)
    w = scale / ratio
    h = scale * ratio / 1.4142135
    b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, W - 1)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, H - 1)
    return b


def _clustered(rs, n):
    """a few hundred objects, each with many near-duplicates: long suppression chains"""
    k = max(1, n // 40)
    x1 = rs.uniform(0, W - 120, k); y1 = rs.uniform(0, H - 90, k)
    w = rs.uniform(20, 300, k); h = rs.uniform(16, 200, k)
    base = np.stack([x1, y1, np.minimum(x1 + w, W - 1), np.minimum(y1 + h, H - 1)], 1)
    b = base[rs.randint(0, k, n)] + rs.uniform(-6, 6, (n, 4))
    b[:, 0::2] = np.clip(b[:, 0::2], 0, W - 1)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, H - 1)
    b[:, 2] = np.maximum(b[:, 2], b[:, 0]); b[:, 3] = np.maximum(b[:, 3], b[:, 1])
    return b


def make(name):
    """-> float32 [n, 5] (x1, y1, x2, y2, score), sorted by strictly decreasing score"""
    _, n, kind, _ = next(c for c in CASES if c[0] == name)
    rs = np.random.RandomState(int(hashlib.sha256(name.encode()).hexdigest()[:8], 16))
    if kind == "rpn":
        b = _rpn_like(rs, n)
    elif kind == "clustered":
        b = _clustered(rs, n)
    else:
        b = np.round(_clustered(rs, n))
    s = (n - np.arange(n)) / float(n + 1)                    # strictly decreasing, also after the cast to float32
    out = np.ascontiguousarray(np.concatenate([b, s[:, None]], 1).astype(np.float32))
    assert n < 2 or (np.diff(out[:, 4]) < 0).all(), "scores must be strictly decreasing in float32 (argsort order = identity)"
    thresh = next(c for c in CASES if c[0] == name)[3]
    for _ in range(50):                                      # integer boxes do hit IoU == 0.5 exactly: move the later box a quarter pixel
        pairs = tie_pairs(out, thresh)
        if not pairs:
            break
        for _, j in pairs:
            if out[j, 2] + 0.25 <= W - 1:
                out[j, 2] += np.float32(0.25)
            else:
                out[j, 0] -= np.float32(0.25)
    else:
        raise AssertionError("could not make %s tie-free" % name)
    return out


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def iou_rows_f32(boxes, i, js):
    """float32 IoU(+1) of box i against boxes js in the operation order of cython_nms.pyx:70-81 / nms_kernel.cu devIoU"""
    f = np.float32
    x1, y1, x2, y2 = (boxes[:, k] for k in range(4))
    area = (x2 - x1 + f(1)) * (y2 - y1 + f(1))
    w = np.maximum(f(0), np.minimum(x2[i], x2[js]) - np.maximum(x1[i], x1[js]) + f(1))
    h = np.maximum(f(0), np.minimum(y2[i], y2[js]) - np.maximum(y1[i], y1[js]) + f(1))
    inter = w * h
    return inter / (area[i] + area[js] - inter)


def tie_pairs(boxes, thresh):
    """pairs (i, j > i) whose float32 IoU equals the float32 threshold exactly: the only inputs on which the reference's two NMS
    implementations disagree (cython_nms.pyx:81 suppresses on >=, nms_kernel.cu:67 on >)"""
    t = np.float32(thresh)
    n = len(boxes)
    out = []
    for i in range(n - 1):
        js = np.arange(i + 1, n)
        hit = js[iou_rows_f32(boxes, i, js) == t]
        out += [(i, int(j)) for j in hit]
    return out
