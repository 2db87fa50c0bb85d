"""Import harness for the *reference* Python tree (/root/reference), CPU only.

Used ONLY by tests/golden/make_golden.py, in the build container, to generate
the committed golden vectors.  It never runs on the GPU box (the reference does
not travel) and nothing under scda_amd/ imports it.

The reference is PyTorch-0.4-era code; the shims below are the minimum that
lets its unmodified modules import and run on torch 2.x / numpy 2.x
(SURVEY.md section 8c lists them):
  1. numpy aliases np.float / np.int            (utils/anchor_helper.py:46-53)
  2. Tensor.cuda()/Module.cuda() -> identity     (functions/anchor_target.py:111 ...)
  3. sys.modules['extensions'] pre-seeded with `nms` and `RoIPool` backed by the
     CPU oracle (oracle/liboracle.so) -- this is exactly the drop-in boundary --
     and sys.modules['extensions._cython_bbox'] exposing cython_bbox
  4. stub modules cv2, torchvision.transforms, datasets.pycocotools._mask
  5. nn.Sequential._modules made an OrderedDict for `popitem(last=True)`
     (models/faster_rcnn/vgg_adver_expansion_cluster.py:38)
"""
import collections
import ctypes
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("SCDA_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_oracle():
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return ctypes.CDLL(path)


def _fp(t):
    return ctypes.c_void_p(t.data_ptr())


def make_extensions_module(bbox_overlaps_fn):
    """Build the fake `extensions` package: oracle NMS + oracle RoIPool."""
    lib = load_oracle()

    def nms(dets, thresh):
        dets = dets.float().contiguous()
        n = dets.shape[0]
        keep = torch.zeros(max(n, 1), dtype=torch.int64)
        num = torch.zeros(1, dtype=torch.int64)
        lib.orc_nms(_fp(dets), ctypes.c_int(n), ctypes.c_float(thresh), _fp(keep), _fp(num))
        return keep[: int(num[0])].contiguous()

    class _RoIPoolFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, feat, rois, ph, pw, scale):
            feat = feat.contiguous()
            rois = rois.contiguous().float()
            B, C, H, W = feat.shape
            R = rois.shape[0]
            out = torch.zeros(R, C, ph, pw)
            arg = torch.zeros(R, C, ph, pw, dtype=torch.int32)
            lib.orc_roi_pool_fwd(_fp(feat), _fp(rois), R, C, H, W, ph, pw, ctypes.c_float(scale), _fp(out), _fp(arg))
            ctx.save_for_backward(rois, arg)
            ctx.cfg = (B, C, H, W, ph, pw, scale)
            return out

        @staticmethod
        def backward(ctx, g):
            rois, arg = ctx.saved_tensors
            B, C, H, W, ph, pw, scale = ctx.cfg
            g = g.contiguous()
            gi = torch.zeros(B, C, H, W)
            lib.orc_roi_pool_bwd(_fp(g), _fp(arg), _fp(rois), rois.shape[0], B, C, H, W, ph, pw,
                                 ctypes.c_float(scale), _fp(gi))
            return gi, None, None, None, None

    class RoIPool(nn.Module):
        def __init__(self, pooled_height, pooled_width, spatial_scale):
            super().__init__()
            self.pooled_width = int(pooled_width)
            self.pooled_height = int(pooled_height)
            self.spatial_scale = float(spatial_scale)

        def forward(self, features, rois):
            assert rois.shape[1] == 5
            return _RoIPoolFn.apply(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale)

    ext = types.ModuleType("extensions")
    ext.__path__ = []
    ext.nms = nms
    ext.RoIPool = RoIPool
    cb_pkg = types.ModuleType("extensions._cython_bbox")
    cb_pkg.__path__ = []
    cb = types.ModuleType("extensions._cython_bbox.cython_bbox")
    cb.bbox_overlaps = bbox_overlaps_fn
    cb_pkg.cython_bbox = cb
    sys.modules["extensions"] = ext
    sys.modules["extensions._cython_bbox"] = cb_pkg
    sys.modules["extensions._cython_bbox.cython_bbox"] = cb
    return ext


def build_reference_cython_bbox(workdir="/tmp/scda_ref_cython"):
    """Compile the reference's cython_bbox.pyx UNMODIFIED, outputs under /tmp only."""
    import subprocess
    os.makedirs(workdir, exist_ok=True)
    src = os.path.join(REF, "extensions", "_cython_bbox", "cython_bbox.pyx")
    so = [f for f in os.listdir(workdir) if f.startswith("cython_bbox") and f.endswith(".so")]
    if not so:
        setup = os.path.join(workdir, "setup_tmp.py")
        with open(setup, "w") as f:
            f.write(
                "from setuptools import setup, Extension\n"
                "from Cython.Build import cythonize\nimport numpy as np\n"
                f"ext = Extension('cython_bbox', [r'{os.path.join(workdir, 'cython_bbox.pyx')}'], include_dirs=[np.get_include()],"
                " extra_compile_args=['-O2'])\n"
                "setup(ext_modules=cythonize([ext], language_level=2))\n")
        # cythonize wants the .pyx beside the build dir; link, do not copy into the repo
        link = os.path.join(workdir, "cython_bbox.pyx")
        if not os.path.exists(link):
            os.symlink(src, link)
        subprocess.check_call([sys.executable, setup, "build_ext", "--inplace"], cwd=workdir,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, workdir)
    import cython_bbox  # noqa
    sys.path.pop(0)
    return cython_bbox


def install_shims():
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "int"):
        np.int = int
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    if not torch.cuda.is_available():
        torch.cuda.FloatTensor = torch.FloatTensor

    cv2 = types.ModuleType("cv2")
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda *_: None)
    cv2.setNumThreads = lambda *_: None
    sys.modules.setdefault("cv2", cv2)

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class _Id:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    tvt.Normalize = _Id
    tvt.ToTensor = _Id
    tvt.Compose = _Id
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)

    mk = types.ModuleType("datasets.pycocotools._mask")
    for a in ("iou", "merge", "frPyObjects", "encode", "decode", "area", "toBbox"):
        setattr(mk, a, lambda *x, **k: None)
    sys.modules["datasets.pycocotools._mask"] = mk

    # (5) `self.features._modules.popitem(last=True)` needs an OrderedDict
    _orig_seq_init = nn.Sequential.__init__

    def _seq_init(self, *args):
        _orig_seq_init(self, *args)
        if not isinstance(self._modules, collections.OrderedDict):
            object.__setattr__(self, "_modules", collections.OrderedDict(self._modules))

    nn.Sequential.__init__ = _seq_init


def import_reference():
    """Returns a namespace of reference modules (imported from REF, unmodified)."""
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}; golden vectors can only be generated in the build container")
    install_shims()
    cyb = build_reference_cython_bbox()
    make_extensions_module(cyb.bbox_overlaps)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import importlib
    ns = types.SimpleNamespace()
    ns.cython_bbox = cyb
    ns.anchor_helper = importlib.import_module("utils.anchor_helper")
    ns.bbox_helper = importlib.import_module("utils.bbox_helper")
    ns.anchor_target = importlib.import_module("functions.anchor_target")
    ns.rpn_proposal = importlib.import_module("functions.rpn_proposal")
    ns.proposal_target = importlib.import_module("functions.proposal_target")
    ns.predict_bbox = importlib.import_module("functions.predict_bbox")
    ns.mask = importlib.import_module("functions.mask")
    ns.vgg = importlib.import_module("models.faster_rcnn.vgg_adver_expansion_cluster")
    ns.frcnn = importlib.import_module("models.faster_rcnn.faster_rcnn_adver_expansion_reweight_cluster")
    ns.head = importlib.import_module("models.head")
    return ns


def import_reference_driver(argv):
    """Import tools/faster_rcnn_train_val.py and set its module-global `args`."""
    ns = import_reference()
    import importlib
    T = importlib.import_module("tools.faster_rcnn_train_val")
    T.args = T.parser.parse_args(argv)
    ns.T = T
    return ns


# ---------------------------------------------------------------------------
# The reference's own pure-Python RoIPool (extensions/_roi_pooling/modules/roi_pool_py.py:7-47),
# imported UNMODIFIED.  It is torch<=0.3 code; three harness-side legacy semantics make it run:
#   * Tensor.cuda() -> identity (install_shims)
#   * the module's `torch.max(x, dim)` keeps the reduced dimension, as torch<=0.3 did (:45-46
#     reduce [C,h,w] over dim 1 and then over dim 2)
#   * iterating `rois` yields rows whose integer index gives a 1-element tensor, so that
#     `roi[0].data[0]` (:20) works as it did on 0.3 Variables
# ---------------------------------------------------------------------------
class _LegacyTorch:
    """`torch` as the module sees it: everything forwarded, `max(x, dim)` with keepdim=True."""

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        return getattr(self._real, name)

    def max(self, x, dim=None, *a, **k):
        if dim is None:
            return self._real.max(x)
        return self._real.max(x, dim, keepdim=True)


class _LegacyRow:
    def __init__(self, t):
        self.t = t

    def __getitem__(self, i):
        return self.t[i:i + 1] if isinstance(i, int) else self.t[i]


class _LegacyRois:
    """[R,5] tensor that iterates as torch 0.3 Variables did."""

    def __init__(self, t):
        self.t = t

    def size(self):
        return self.t.size()

    def __iter__(self):
        return (_LegacyRow(r) for r in self.t)


def import_reference_roi_pool_py():
    """Returns f(features[B,C,H,W], rois[R,5], ph, pw, scale) -> out[R,C,ph,pw] computed by the
    reference's roi_pool_py.RoIPool.forward."""
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    install_shims()
    import importlib.util
    path = os.path.join(REF, "extensions", "_roi_pooling", "modules", "roi_pool_py.py")
    spec = importlib.util.spec_from_file_location("_ref_roi_pool_py", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.torch = _LegacyTorch(torch)

    def run(features, rois, ph, pw, scale):
        m = mod.RoIPool(ph, pw, scale)
        with torch.no_grad():
            return m.forward(features, _LegacyRois(rois))

    return run
