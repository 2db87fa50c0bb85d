"""Seeded inputs of the mask-target cases (tests/golden/make_golden_mask_targets.py runs the reference on them, tests/test_mask_targets.py
the drop-in): RPN-like proposals jittered around ground-truth boxes plus background boxes, elliptical ground-truth masks."""
import numpy as np

CASES = {
    # name: (seed, H, W, gts per image (batch), proposals, cfg)
    "one_image_G5": (5, 200, 320, [5], 300,
                     {'positive_iou_thresh': 0.5, 'batch_size_per_image': 16, 'label_h': 28, 'label_w': 28, 'append_gts': True,
                      'num_classes': 9}),
    "two_images_ragged": (6, 160, 224, [3, 1], 200,
                          {'positive_iou_thresh': 0.4, 'batch_size_per_image': 8, 'label_h': 14, 'label_w': 20, 'append_gts': False,
                           'num_classes': 5}),
    "all_kept_no_sampling": (7, 128, 128, [2], 60,
                             {'positive_iou_thresh': 0.6, 'batch_size_per_image': -1, 'label_h': 7, 'label_w': 7, 'append_gts': True,
                              'num_classes': 3}),
    "no_positive": (8, 96, 96, [1], 10,
                    {'positive_iou_thresh': 0.99, 'batch_size_per_image': 4, 'label_h': 7, 'label_w': 7, 'append_gts': False,
                     'num_classes': 3}),
}


def ellipse_masks(gts, H, W):
    """[G, H, W] uint8: the ellipse inscribed in each box (zero planes for padded rows)"""
    yy, xx = np.mgrid[0:H, 0:W]
    out = np.zeros((gts.shape[0], H, W), dtype=np.uint8)
    for i, (x1, y1, x2, y2) in enumerate(gts[:, :4]):
        if x2 <= x1 or y2 <= y1:
            continue
        cx, cy, rx, ry = (x1 + x2) / 2.0, (y1 + y2) / 2.0, (x2 - x1) / 2.0, (y2 - y1) / 2.0
        out[i] = (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0).astype(np.uint8)
    return out


def make(name):
    seed, H, W, per_image, n_prop, cfg = CASES[name]
    rng = np.random.RandomState(seed)
    B, G = len(per_image), max(per_image)
    gts = np.zeros((B, G, 5), dtype=np.float32)                  # zero rows pad the ragged batch
    masks = np.zeros((B, G, H, W), dtype=np.uint8)
    props = []
    for b, g in enumerate(per_image):
        w = rng.randint(24, W // 2, size=g); h = rng.randint(24, H // 2, size=g)
        x1 = rng.randint(0, W - w); y1 = rng.randint(0, H - h)
        gts[b, :g] = np.stack([x1, y1, x1 + w, y1 + h, rng.randint(1, cfg['num_classes'], size=g)], axis=1)
        masks[b] = ellipse_masks(gts[b], H, W)
        n = n_prop // B
        src = gts[b, rng.randint(0, g, size=n), :4]
        jit = rng.uniform(-0.25, 0.25, size=(n, 4)) * np.stack([src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]] * 2, axis=1)
        boxes = src + jit
        far = rng.rand(n) < 0.3                                  # background boxes anywhere
        boxes[far] = np.stack([rng.uniform(0, W / 2, far.sum()), rng.uniform(0, H / 2, far.sum()),
                               rng.uniform(W / 2, W + 20, far.sum()), rng.uniform(H / 2, H + 20, far.sum())], axis=1)
        props.append(np.concatenate([np.full((n, 1), b), boxes, rng.rand(n, 1)], axis=1))
    props = np.concatenate(props).astype(np.float32)
    info = np.tile(np.array([[H, W, 1.0]], dtype=np.float32), (B, 1))
    return props, gts, masks, info, dict(cfg)


def predict_case():
    """rois [R, 7] (b, x1, y1, x2, y2, score, class) inside a 60 x 80 image, heat maps [R, 4, 14, 14], image_info [[60, 80, 1]]"""
    rng = np.random.RandomState(17)
    R = 6
    x1 = rng.randint(0, 40, R); y1 = rng.randint(0, 30, R)
    rois = np.stack([np.zeros(R), x1, y1, x1 + rng.randint(2, 39, R), y1 + rng.randint(2, 29, R), rng.rand(R), rng.randint(0, 4, R)], 1)
    heat = rng.rand(R, 4, 14, 14).astype(np.float32)
    return rois.astype(np.float32), heat, np.array([[60, 80, 1.0]], dtype=np.float32)
