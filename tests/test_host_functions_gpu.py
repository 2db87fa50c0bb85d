"""The same golden-vector checks as tests/test_host_functions.py, with the host logic's IoU and NMS running on the
MI355X (default backend): anchor labels, proposal lists, sampled RoIs and predicted boxes must still be bit-exact."""
import pytest

from test_host_functions import check_l2, check_predict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G", [3, 12, 30])
def test_l2_functions_match_reference_on_device(cuda, golden_dir, G):
    from scda_amd.dropin import backend
    backend.reset()
    check_l2(golden_dir, G)


@pytest.mark.parametrize("G", [3, 12, 30])
def test_device_resident_box_logic_matches_reference(cuda, golden_dir, G):
    """anchor labelling (IoU, labels, sub-sampling with host-drawn indices, target maps), proposal generation (decode, clip,
    size test, NMS, gather) and RoI sampling (clip, IoU, best gt, threshold tests, ordered index lists, gather of the sampled rows,
    target / weight maps) on the MI355X -- scda_amd/device_boxes.py, box_ops.hip -- against the SAME vectors the reference's numpy
    code produced: labels, target values, proposal lists, sampled RoIs bit for bit"""
    from scda_amd import device_boxes
    from scda_amd.dropin import backend
    backend.reset()
    assert device_boxes.enabled()
    check_l2(golden_dir, G, device=cuda)


def test_nms_with_validity_flags_equals_nms_of_the_filtered_list(cuda):
    import numpy as np
    import torch
    from oracle import native_ops as orc
    from scda_amd import native
    from test_oracle_golden import rand_boxes
    rs = np.random.RandomState(3)
    b = rand_boxes(rs, 3000)
    b[1500:, :4] = b[:1500, :4] + rs.uniform(-3, 3, (1500, 4)).astype(np.float32)
    valid = (rs.uniform(size=3000) > 0.3)
    d = torch.from_numpy(b).to(cuda)
    n = b.shape[0]
    keep = torch.empty(n, dtype=torch.int64, device=cuda); num = torch.zeros(1, dtype=torch.int64, device=cuda)
    ws = torch.empty(native.lib().scda_nms_workspace_bytes(n), dtype=torch.uint8, device=cuda)
    v = torch.from_numpy(valid.astype(np.uint8)).to(cuda)
    import ctypes
    for max_keep in (0, 300):
        rc = native.lib().scda_nms_valid_hip(ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(v.data_ptr()), n, ctypes.c_float(0.7),
                                             ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(keep.data_ptr()),
                                             ctypes.c_void_p(num.data_ptr()), max_keep, None)
        assert rc == 0
        got = keep[: int(num)].cpu().numpy()
        want = np.nonzero(valid)[0][orc.nms(b[valid], 0.7)]
        assert np.array_equal(got, want[:max_keep] if max_keep else want)


def test_predict_bbox_matches_reference_on_device(cuda, golden_dir):
    from scda_amd.dropin import backend
    backend.reset()
    check_predict(golden_dir)


@pytest.mark.parametrize("case", ["one_gt", "few_candidates", "nothing_survives_min_size", "all_candidates", "tiny_gt_no_anchor"])
def test_device_box_logic_equals_numpy_path_on_edge_cases(cuda, case, monkeypatch):
    """the device-resident kernels against THIS repository's numpy path (itself pinned to the reference's vectors) where the
    fixtures do not reach: a single ground-truth box, fewer candidates than the post-NMS quota, no box passing the size test
    (empty result), pre_nms_top_n <= 0 (all 30720 anchors ranked), a gt so small that no anchor reaches IoU 0.1"""
    import copy
    import numpy as np
    import torch
    from test_host_functions import CFG, synth_rpn_outputs
    from scda_amd.dropin.functions.anchor_target import compute_anchor_targets
    from scda_amd.dropin.functions.rpn_proposal import compute_rpn_proposals
    cfg_a, cfg_p = copy.deepcopy(CFG["train_anchor_target_cfg"]), copy.deepcopy(CFG["train_rpn_proposal_cfg"])
    gts = np.array([[[100, 80, 400, 300, 3], [600, 200, 900, 480, 5], [30, 300, 200, 500, 1]]], dtype=np.float32)
    if case == "one_gt":
        gts = gts[:, :1]
    elif case == "tiny_gt_no_anchor":
        gts = np.array([[[500, 250, 503, 252, 2], [100, 80, 400, 300, 3]]], dtype=np.float32)
    elif case == "few_candidates":
        cfg_p["pre_nms_top_n"] = 50
    elif case == "nothing_survives_min_size":
        cfg_p["roi_min_size"] = 5000
    elif case == "all_candidates":
        cfg_p["pre_nms_top_n"] = 0
        cfg_p["post_nms_top_n"] = 300
    info = torch.tensor([[512, 1024, 1.0]])
    cls, loc = synth_rpn_outputs(77)
    out = {}
    for dev_boxes in ("1", "0"):
        monkeypatch.setenv("SCDA_DEVICE_BOXES", dev_boxes)
        g = torch.from_numpy(gts).to(cuda)
        g._scda_host = gts
        np.random.seed(5)
        ct, lt, lm, norm = compute_anchor_targets((1, 60, 32, 64), cfg_a, g, info, None)
        props = compute_rpn_proposals(cls.to(cuda), loc.to(cuda), cfg_p, info)
        out[dev_boxes] = (ct.cpu(), lt.cpu(), lm.cpu(), norm, props.cpu(), np.random.rand())
    a, b = out["1"], out["0"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3]
    assert a[4].shape == b[4].shape and torch.equal(a[4], b[4])
    assert a[5] == b[5]                       # the numpy generator ended in the same state: same draws were made
    if case == "nothing_survives_min_size":
        assert tuple(a[4].shape) == (0, 6)
    if case == "few_candidates":
        assert 0 < a[4].shape[0] <= 50


@pytest.mark.parametrize("case", ["pad_by_resampling", "one_gt", "no_foreground_surplus", "zero_padded_gts", "host_proposals"])
def test_device_roi_sampling_equals_numpy_path(cuda, case, monkeypatch):
    """device_boxes.proposal_targets against THIS repository's numpy path (pinned to the reference's vectors by the fixtures) where
    the fixtures do not reach: fewer candidates than the 512-RoI batch (the padding draw), a single gt, fewer foreground rows than
    the 25 % budget, zero-padded gt rows, proposals that exist on the host only."""
    import copy
    import numpy as np
    import torch
    from test_host_functions import CFG
    from scda_amd.dropin.functions.proposal_target import compute_proposal_targets
    cfg = copy.deepcopy(CFG["train_proposal_target_cfg"])
    rs = np.random.RandomState(5)
    gts = np.array([[[100, 80, 400, 300, 3], [600, 200, 900, 480, 5], [30, 300, 200, 500, 1]]], dtype=np.float32)
    n = 1500
    if case == "pad_by_resampling":
        n = 200
    elif case == "one_gt":
        gts = gts[:, :1]
    elif case == "zero_padded_gts":
        gts = np.concatenate([gts, np.zeros((1, 2, 5), np.float32)], 1)
    x1 = rs.uniform(-30, 1000, n); y1 = rs.uniform(-30, 500, n)
    b = np.stack([np.zeros(n), x1, y1, x1 + rs.uniform(8, 400, n), y1 + rs.uniform(8, 300, n), np.sort(rs.uniform(0, 1, n))[::-1]], 1)
    k = 60 if case != "no_foreground_surplus" else 6
    for j in range(k):      # some candidates close to a gt box: foreground
        g = gts[0, j % (1 if case == "one_gt" else 3)]
        b[(j * 7) % n, 1:5] = g[:4] + rs.uniform(-12, 12, 4)
    props = torch.from_numpy(b.astype(np.float32))
    info = torch.tensor([[512, 1024, 1.0]])
    gd = torch.from_numpy(gts).to(cuda)
    gd._scda_host = gts
    if case != "host_proposals":
        props._scda_dev = props.to(cuda)
    out = {}
    for dev_boxes in ("1", "0"):
        monkeypatch.setenv("SCDA_DEVICE_BOXES", dev_boxes)
        np.random.seed(11)
        r = compute_proposal_targets(props, cfg, gd, info, None)
        out[dev_boxes] = [t.cpu().numpy() for t in r] + [r[0]._scda_host, np.random.uniform()]     # ... and the RNG stream ends in the same state
    for a, c in zip(out["1"], out["0"]):
        np.testing.assert_array_equal(a, c)
    assert out["1"][0].shape == (512, 5) and out["1"][2].shape == (512, 36)
