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


def test_predict_bbox_matches_reference_on_device(cuda, golden_dir):
    from scda_amd.dropin import backend
    backend.reset()
    check_predict(golden_dir)
