# `utils` is a split package: distributed_utils / anchor_helper / bbox_helper / cal_mAP live here, the reference's
# lr_helper / log_helper / load_helper are picked up from the reference checkout further down sys.path.
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
