"""Mask targets of BASELINE.json configs[4] (functions/mask.py:51-179): the drop-in against the fixture the REFERENCE's
compute_mask_targets produced on the same seeded inputs (tests/golden/make_golden_mask_targets.py), the OpenCV-resize restatement's
properties, and the transposed-convolution layer's index mapping."""
import os

import numpy as np
import pytest
import torch

import mask_cases as mcases



@pytest.fixture()
def cpu_backend():
    """the IoU matrix of the host functions on the C oracle (the product computes it on the MI355X)"""
    from oracle import native_ops as orc
    from scda_amd.dropin import backend
    backend.use(bbox_overlaps=lambda b, q: orc.bbox_overlaps(b[:, :4], q[:, :4]),
                nms=lambda d, t: torch.from_numpy(orc.nms(d.numpy(), t)))
    yield
    backend.reset()


GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_targets_ref.npz"))


@pytest.mark.parametrize("name", list(mcases.CASES))
def test_compute_mask_targets_equals_reference(name, cpu_backend):
    from scda_amd.dropin.functions.mask import compute_mask_targets
    props, gts, masks, info, cfg = mcases.make(name)
    np.random.seed(11)
    rois, labels = compute_mask_targets(torch.from_numpy(props), cfg, torch.from_numpy(gts), torch.from_numpy(masks),
                                        torch.from_numpy(info))
    assert rois.dtype == torch.float32 and labels.dtype == torch.float32
    assert np.array_equal(rois.numpy(), GOLD[name + "_rois"])
    assert np.array_equal(labels.numpy(), GOLD[name + "_labels"].astype(np.float32))
    assert int(np.random.randint(1 << 30)) == int(GOLD[name + "_rng_after"])     # same number of RNG draws
    if rois.shape[1] == 6:       # contract: the RoI's own class plane is binary, every other plane is -1
        lab, cls = labels.numpy(), rois[:, 5].long().numpy()
        for r in range(lab.shape[0]):
            assert set(np.unique(lab[r, cls[r]])) <= {0.0, 1.0}
            others = np.delete(lab[r], cls[r], axis=0)
            assert (others == -1).all()
    else:
        assert rois.shape == (1, 5) and (labels.numpy() == -1).all()


def test_mask_targets_accept_device_style_inputs(cpu_backend):
    """lists / tensors carrying a host copy (`_scda_host`) give the same result as plain tensors"""
    from scda_amd.dropin.functions.mask import compute_mask_targets
    props, gts, masks, info, cfg = mcases.make("one_image_G5")
    g = torch.from_numpy(gts.copy())
    g._scda_host = gts
    np.random.seed(11)
    rois, labels = compute_mask_targets(torch.from_numpy(props), cfg, g, masks, info.tolist())
    assert np.array_equal(rois.numpy(), GOLD["one_image_G5_rois"])
    assert np.array_equal(labels.numpy(), GOLD["one_image_G5_labels"].astype(np.float32))


def _bilinear_float(img, dw, dh):
    """half-pixel-centre bilinear in float64 (the definition cv2.INTER_LINEAR discretises)"""
    h, w = img.shape
    fx = np.clip((np.arange(dw) + 0.5) * w / dw - 0.5, 0, w - 1); fy = np.clip((np.arange(dh) + 0.5) * h / dh - 0.5, 0, h - 1)
    x0 = np.floor(fx).astype(int); y0 = np.floor(fy).astype(int)
    x1 = np.minimum(x0 + 1, w - 1); y1 = np.minimum(y0 + 1, h - 1)
    ax = fx - x0; ay = fy - y0
    im = img.astype(np.float64)
    top = im[y0][:, x0] * (1 - ax) + im[y0][:, x1] * ax
    bot = im[y1][:, x0] * (1 - ax) + im[y1][:, x1] * ax
    return top * (1 - ay)[:, None] + bot * ay[:, None]


def test_resize_linear_u8_properties():
    from scda_amd.dropin.functions.mask import resize_linear_u8
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, size=(37, 53)).astype(np.uint8)
    assert np.array_equal(resize_linear_u8(img, 53, 37), img)                          # same size: identity
    for dw, dh in ((28, 28), (106, 74), (7, 90), (200, 3)):
        got = resize_linear_u8(img, dw, dh).astype(np.float64)
        assert got.shape == (dh, dw)
        assert np.abs(got - _bilinear_float(img, dw, dh)).max() <= 1.0                 # 11-bit weights, (x + 2) >> 2 rounding
    binary = (rng.rand(61, 45) > 0.5).astype(np.uint8)
    out = resize_linear_u8(binary, 28, 28)
    assert set(np.unique(out)) <= {0, 1}
    exact = _bilinear_float(binary, 28, 28)
    # on 0/1 images the vertical pass keeps two fractional bits per term and truncates: the result is 1 from 0.75 up, 0 below 0.5
    assert (out[exact >= 0.75] == 1).all() and (out[exact < 0.5] == 0).all()
    # known answer: a 1 x 2 image doubled in width -- taps (0), (.75,.25), (.25,.75), (1)
    assert resize_linear_u8(np.array([[0, 200]], np.uint8), 4, 1).tolist() == [[0, 50, 150, 200]]


def test_conv_transpose_2x2_s2_index_mapping(monkeypatch):
    """the layer = 1x1 convolution to 4*out channels + pixel shuffle; checked against nn.ConvTranspose2d with the convolution itself
    done by torch (the HIP kernel's own arithmetic is checked on the device, tests/test_maskrcnn_gpu.py)"""
    import torch.nn.functional as F
    from scda_amd import autograd_ops as A, layers as L
    monkeypatch.setattr(A, "conv2d", lambda x, w, b, s, p, act=0, slope=0.01, *a, **k: F.conv2d(x, w, b, stride=s, padding=p))
    torch.manual_seed(0)
    layer = L.ConvTranspose2x2s2(6, 5)
    x = torch.randn(3, 6, 4, 7, requires_grad=True)
    want = F.conv_transpose2d(x, layer.weight, layer.bias, stride=2)
    got = layer(x)
    assert got.shape == want.shape == (3, 5, 8, 14)
    assert torch.allclose(got, want, atol=1e-6)
    g = torch.randn_like(want)
    gw = torch.autograd.grad(want, [x, layer.weight, layer.bias], g, retain_graph=True)
    gg = torch.autograd.grad(got, [x, layer.weight, layer.bias], g)
    for a, b in zip(gg, gw):
        assert torch.allclose(a, b, atol=1e-5)
    assert list(layer.state_dict()) == ["weight", "bias"] and tuple(layer.weight.shape) == (6, 5, 2, 2)


def test_generate_mask_labels_equals_per_window_resize():
    """the all-RoIs-at-once gather against `resize_linear_u8` on each RoI's window (what the reference's loop over cv2.resize does)"""
    from scda_amd.dropin.functions.mask import generate_mask_labels, resize_linear_u8
    rng = np.random.RandomState(9)
    masks = (rng.rand(5, 90, 130) > 0.4).astype(np.uint8) * rng.randint(1, 255, size=(5, 1, 1)).astype(np.uint8)
    x1 = rng.randint(0, 100, 40); y1 = rng.randint(0, 60, 40)
    rois = np.stack([x1, y1, x1 + rng.randint(1, 30, 40), y1 + rng.randint(1, 30, 40)], 1).astype(np.float32)
    rois[0] = [3, 4, 4, 5]                                     # a one-pixel window
    rois[1] = [0, 0, 130, 90]                                  # the whole plane
    index = rng.randint(0, 5, 40)
    for mh, mw in ((28, 28), (7, 20)):
        got = generate_mask_labels(rois, masks, mh, mw, index=index)
        assert got.dtype == np.int32 and got.shape == (40, mh, mw)
        for i, (a, b, c, d) in enumerate(rois.astype(np.int32)):
            assert np.array_equal(got[i], resize_linear_u8(masks[index[i]][b:d, a:c], mw, mh)), i
    assert np.array_equal(generate_mask_labels(rois[:5], masks, 14, 14), generate_mask_labels(rois[:5], masks, 14, 14, index=np.arange(5)))


def test_predict_masks_equals_reference():
    """functions/mask.py:21-49 -- each RoI's class plane resized to the RoI with PIL and pasted into the image (the reference's own
    code produced the fixture: PIL is available in the build container)"""
    from scda_amd.dropin.functions.mask import predict_masks
    rois, heat, info = mcases.predict_case()
    got = predict_masks(torch.from_numpy(rois), torch.from_numpy(heat), info)
    assert len(got) == rois.shape[0] and all(m.shape == (60, 80) and m.dtype == np.float32 for m in got)
    assert np.array_equal(np.stack(got), GOLD["predict_masks"])
