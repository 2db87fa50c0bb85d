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
    """anchor labelling (IoU, labels, sub-sampling with host-drawn indices, target maps) and proposal generation (decode, clip,
    size test, NMS, gather) entirely on the MI355X -- scda_amd/device_boxes.py, box_ops.hip -- against the SAME vectors the
    reference's numpy code produced: labels, target values, proposal lists bit for bit"""
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
