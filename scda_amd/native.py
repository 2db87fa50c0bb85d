"""ctypes binding of libscda_ops.so (C ABI declared in include/scda_ops.h).

Every wrapper takes torch CUDA tensors, checks device/dtype/contiguity, passes
raw device pointers + sizes + the current HIP stream, and raises on a non-zero
status.  There is no CPU fallback: operators raise if the library or a HIP
device is missing.
"""
import ctypes
import os
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCDA_OPS_LIB") or os.path.join(_HERE, "libscda_ops.so")   # (the env override: A/B builds of the library)

_lib = None


class ScdaNativeError(RuntimeError):
    pass


def lib():
    """Load libscda_ops.so once; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ScdaNativeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C scda_amd/csrc` (no CPU fallback exists)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.scda_last_error.restype = ctypes.c_char_p
        _lib.scda_nms_workspace_bytes.restype = ctypes.c_size_t
    return _lib


def _check(status, what):
    if status != 0:
        msg = lib().scda_last_error().decode("utf-8", "replace")
        raise ScdaNativeError(f"{what} failed with status {status}: {msg}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """current HIP stream of the current device as void* (one C call; this runs once per kernel launch)"""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _req(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise ScdaNativeError(f"{name} must live on the HIP device (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


f32 = ctypes.c_float
i32 = ctypes.c_int


# ----------------------------------------------------------------- NMS ------
def nms(boxes, thresh, max_keep=0):
    """boxes [n,5] fp32 CUDA, sorted by score desc -> (keep int64[n] CUDA, num_out int64[1] CUDA)."""
    _req(boxes, "boxes")
    n = boxes.shape[0]
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=boxes.device)
    num = torch.zeros(1, dtype=torch.int64, device=boxes.device)
    ws = torch.empty(max(lib().scda_nms_workspace_bytes(i32(n)), 8), dtype=torch.uint8, device=boxes.device)
    _check(lib().scda_nms_hip(_p(boxes), i32(n), f32(thresh), _p(ws), _p(keep), _p(num), i32(max_keep), _stream()),
           "scda_nms_hip")
    return keep, num


def nms_segments(boxes, seg, max_n, thresh):
    """S independent score-sorted lists in one mask + one sweep launch (scda_nms_segments_hip): boxes [rows,5] fp32 CUDA (the lists
    back to back), seg int64 [S,3] CUDA = (first row, length, first mask word) -> (keep int64 [rows] CUDA, num int64 [S] CUDA)"""
    _req(boxes, "boxes"); _req(seg, "seg", torch.int64)
    S, rows = seg.shape[0], boxes.shape[0]
    keep = torch.empty(max(rows, 1), dtype=torch.int64, device=boxes.device)
    num = torch.zeros(max(S, 1), dtype=torch.int64, device=boxes.device)
    words = S * max_n * ((max_n + 63) // 64)                  # upper bound of the lists' mask words
    ws = torch.empty(max(words, 1), dtype=torch.int64, device=boxes.device)
    _check(lib().scda_nms_segments_hip(_p(boxes), _p(seg), i32(S), i32(max_n), f32(thresh), _p(ws), _p(keep), _p(num), _stream()),
           "scda_nms_segments_hip")
    return keep, num


def nms_mask(boxes, thresh):
    _req(boxes, "boxes")
    n = boxes.shape[0]
    cb = (n + 63) // 64
    mask = torch.zeros(n, cb, dtype=torch.int64, device=boxes.device)
    _check(lib().scda_nms_mask_hip(_p(boxes), i32(n), f32(thresh), _p(mask), _stream()), "scda_nms_mask_hip")
    return mask


# ------------------------------------------------------------- RoIPool ------
def roi_pool_fwd(features, rois, ph, pw, scale, want_argmax=True):
    """-> (out [R,C,ph,pw], argmax int32 [R,C,ph,pw] | None).  want_argmax=False (nothing will be differentiated through the
    call, e.g. the target-domain branch) skips the second 51 MB output."""
    _req(features, "features"); _req(rois, "rois")
    if rois.dim() != 2 or rois.shape[1] != 5:
        raise ValueError("rois must be [R,5]")
    B, C, H, W = features.shape
    R = rois.shape[0]
    out = torch.empty(R, C, ph, pw, dtype=torch.float32, device=features.device)
    arg = torch.empty(R, C, ph, pw, dtype=torch.int32, device=features.device) if want_argmax else None
    _check(lib().scda_roi_pool_fwd_hip(_p(features), _p(rois), i32(R), i32(B), i32(C), i32(H), i32(W), i32(ph), i32(pw),
                                       f32(scale), _p(out), _p(arg), _stream()), "scda_roi_pool_fwd_hip")
    return out, arg


def roi_pool_bwd(top_grad, argmax, rois, feat_shape, ph, pw, scale):
    _req(top_grad, "top_grad"); _req(argmax, "argmax", torch.int32); _req(rois, "rois")
    B, C, H, W = feat_shape
    R = rois.shape[0]
    gi = torch.empty(B, C, H, W, dtype=torch.float32, device=top_grad.device)
    _check(lib().scda_roi_pool_bwd_hip(_p(top_grad), _p(argmax), _p(rois), i32(R), i32(B), i32(C), i32(H), i32(W),
                                       i32(ph), i32(pw), f32(scale), _p(gi), _stream()), "scda_roi_pool_bwd_hip")
    return gi


# ------------------------------------------------------------ RoIAlign ------
def roi_align_fwd(features, rois, ah, aw, scale, channel_major=False):
    """-> [R,C,ah,aw], or [C,R,ah,aw] with channel_major (the RoI-head layout of scda_amd/dropin/models/mask_rcnn/resnet.py)"""
    _req(features, "features"); _req(rois, "rois")
    if rois.dim() != 2 or rois.shape[1] != 5:
        raise ValueError("rois must be [R,5]")
    B, C, H, W = features.shape
    R = rois.shape[0]
    out = torch.empty((C, R, ah, aw) if channel_major else (R, C, ah, aw), dtype=torch.float32, device=features.device)
    fn = lib().scda_roi_align_cmajor_fwd_hip if channel_major else lib().scda_roi_align_fwd_hip
    _check(fn(_p(features), _p(rois), i32(R), i32(B), i32(C), i32(H), i32(W), i32(ah), i32(aw), f32(scale), _p(out), _stream()),
           "scda_roi_align_fwd_hip")
    return out


def roi_align_bwd(top_grad, rois, feat_shape, ah, aw, scale, channel_major=False):
    _req(top_grad, "top_grad"); _req(rois, "rois")
    B, C, H, W = feat_shape
    gi = torch.zeros(B, C, H, W, dtype=torch.float32, device=top_grad.device)
    fn = lib().scda_roi_align_cmajor_bwd_hip if channel_major else lib().scda_roi_align_bwd_hip
    _check(fn(_p(top_grad), _p(rois), i32(rois.shape[0]), i32(B), i32(C), i32(H), i32(W), i32(ah), i32(aw), f32(scale), _p(gi),
              _stream()), "scda_roi_align_bwd_hip")
    return gi


# ---------------------------------------------------------- focal loss ------
def focal_sigmoid_fwd(logits, targets, weight_pos, gamma, alpha, num_classes):
    _req(logits, "logits"); _req(targets, "targets", torch.int32)
    losses = torch.empty_like(logits)
    _check(lib().scda_focal_sigmoid_fwd_hip(i32(logits.numel()), _p(logits), _p(targets), f32(weight_pos), f32(gamma),
                                            f32(alpha), i32(num_classes), _p(losses), _stream()),
           "scda_focal_sigmoid_fwd_hip")
    return losses


def focal_sigmoid_bwd(logits, targets, weight_pos, gamma, alpha, num_classes):
    _req(logits, "logits"); _req(targets, "targets", torch.int32)
    dx = torch.empty_like(logits)
    _check(lib().scda_focal_sigmoid_bwd_hip(i32(logits.numel()), _p(logits), _p(targets), _p(dx), f32(weight_pos),
                                            f32(gamma), f32(alpha), i32(num_classes), _stream()),
           "scda_focal_sigmoid_bwd_hip")
    return dx


def focal_softmax_fwd(logits, targets, weight_pos, gamma, alpha, num_classes):
    _req(logits, "logits"); _req(targets, "targets", torch.int32)
    rows = logits.numel() // num_classes
    losses = torch.empty(rows, dtype=torch.float32, device=logits.device)
    priors = torch.empty_like(logits)
    _check(lib().scda_focal_softmax_fwd_hip(i32(logits.numel()), _p(logits), _p(targets), f32(weight_pos), f32(gamma),
                                            f32(alpha), i32(num_classes), _p(losses), _p(priors), _stream()),
           "scda_focal_softmax_fwd_hip")
    return losses, priors


def focal_softmax_bwd(logits, targets, priors, weight_pos, gamma, alpha, num_classes):
    _req(logits, "logits"); _req(targets, "targets", torch.int32); _req(priors, "priors")
    rows = logits.numel() // num_classes
    dx = torch.empty_like(logits)
    buff = torch.empty(rows, dtype=torch.float32, device=logits.device)
    _check(lib().scda_focal_softmax_bwd_hip(i32(logits.numel()), _p(logits), _p(targets), _p(dx), f32(weight_pos),
                                            f32(gamma), f32(alpha), i32(num_classes), _p(priors), _p(buff), _stream()),
           "scda_focal_softmax_bwd_hip")
    return dx


# -------------------------------------------------------- box overlaps ------
def iou_overlaps(b1, b2):
    _req(b1, "b1"); _req(b2, "b2")
    if b1.shape[1] != b2.shape[1]:
        raise ValueError("box widths differ")
    out = torch.empty(b1.shape[0], b2.shape[0], dtype=torch.float32, device=b1.device)
    _check(lib().scda_iou_overlaps_hip(_p(b1), _p(b2), i32(b1.shape[1]), i32(b1.shape[0]), i32(b2.shape[0]), _p(out),
                                       _stream()), "scda_iou_overlaps_hip")
    return out


def bbox_overlaps(boxes, query):
    _req(boxes, "boxes"); _req(query, "query")
    if boxes.shape[1] != 4 or query.shape[1] != 4:
        raise ValueError("bbox_overlaps takes [N,4] and [K,4]")
    out = torch.empty(boxes.shape[0], query.shape[0], dtype=torch.float32, device=boxes.device)
    _check(lib().scda_bbox_overlaps_hip(_p(boxes), i32(boxes.shape[0]), _p(query), i32(query.shape[0]), _p(out),
                                        _stream()), "scda_bbox_overlaps_hip")
    return out


# ------------------------------------------------ box logic on the device ----
def anchor_label(anchors32, gts, neg_thresh, pos_thresh, min_gt_best, bufs):
    """bufs: dict of caller-owned device buffers (best_iou, best_gt, gt_best, labels, pos_list, neg_list, counts)"""
    _req(anchors32, "anchors"); _req(gts, "gts")
    KA, G = anchors32.shape[0], gts.shape[0]
    _check(lib().scda_anchor_label_hip(_p(anchors32), i32(KA), _p(gts), i32(G), i32(gts.shape[1]), f32(neg_thresh), f32(pos_thresh),
                                       f32(min_gt_best), _p(bufs["best_iou"]), _p(bufs["best_gt"]), _p(bufs["gt_best"]),
                                       _p(bufs["labels"]), _p(bufs["pos_list"]), _p(bufs["neg_list"]), _p(bufs["counts"]), _stream()),
           "scda_anchor_label_hip")


def anchor_finalize(bufs, drop_pos, drop_neg, anchors64, gts, A, fh, fw):
    """-> cls_targets int64 [1,A,fh,fw], loc_targets, loc_masks fp32 [1,4A,fh,fw]"""
    dev = gts.device
    cls_t = torch.empty(1, A, fh, fw, dtype=torch.int64, device=dev)
    loc_t = torch.empty(1, 4 * A, fh, fw, dtype=torch.float32, device=dev)
    loc_m = torch.empty(1, 4 * A, fh, fw, dtype=torch.float32, device=dev)
    _check(lib().scda_anchor_finalize_hip(_p(bufs["labels"]), _p(bufs["best_gt"]), _p(bufs["pos_list"]), _p(drop_pos),
                                          i32(0 if drop_pos is None else drop_pos.numel()), _p(bufs["neg_list"]), _p(drop_neg),
                                          i32(0 if drop_neg is None else drop_neg.numel()), _p(anchors64), _p(gts), i32(gts.shape[1]),
                                          i32(A), i32(fh), i32(fw), _p(cls_t), _p(loc_t), _p(loc_m), _stream()),
           "scda_anchor_finalize_hip")
    return cls_t, loc_t, loc_m


def proposal_match(props, gts, img_h, img_w, pos_thresh, neg_hi, neg_lo, bufs):
    """props [n,>=5] fp32 rows (b,x1,y1,x2,y2,..), gts [G,>=5]; bufs: caller-owned device buffers rois [n+G,4], best_iou, best_gt,
    labels, pos_list, neg_list [n+G], counts [2] (scda_proposal_match_hip)"""
    _req(props, "props"); _req(gts, "gts")
    _check(lib().scda_proposal_match_hip(_p(props), i32(props.shape[0]), i32(props.shape[1]), _p(gts), i32(gts.shape[0]), i32(gts.shape[1]),
                                         f32(img_h), f32(img_w), f32(pos_thresh), f32(neg_hi), f32(neg_lo), _p(bufs["rois"]),
                                         _p(bufs["best_iou"]), _p(bufs["best_gt"]), _p(bufs["labels"]), _p(bufs["pos_list"]),
                                         _p(bufs["neg_list"]), _p(bufs["counts"]), _stream()), "scda_proposal_match_hip")


def proposal_finalize(cand_rois, sel, gt_of, enc, gts, num_classes, image_index):
    """-> rois [R,5] fp32, labels int64 [R], loc_targets, loc_weights fp32 [R, 4*num_classes] (scda_proposal_finalize_hip)"""
    _req(cand_rois, "cand_rois"); _req(sel, "sel", torch.int32); _req(gt_of, "gt_of", torch.int32); _req(enc, "enc"); _req(gts, "gts")
    R, dev = sel.numel(), gts.device
    rois = torch.empty(R, 5, dtype=torch.float32, device=dev)
    labels = torch.empty(R, dtype=torch.int64, device=dev)
    t = torch.empty(R, 4 * num_classes, dtype=torch.float32, device=dev)
    w = torch.empty(R, 4 * num_classes, dtype=torch.float32, device=dev)
    _check(lib().scda_proposal_finalize_hip(_p(cand_rois), _p(sel), _p(gt_of), _p(enc), _p(gts), i32(gts.shape[1]), i32(R),
                                            i32(num_classes), f32(image_index), _p(rois), _p(labels), _p(t), _p(w), _stream()),
           "scda_proposal_finalize_hip")
    return rois, labels, t, w


def proposals_from_ranking(order, exp_wh, anchors64, loc, prob, A, fh, fw, img_h, img_w, min_size, nms_thresh, max_keep, image_index):
    """order int32 [n], exp_wh f32 [n,2] (device) -> (out6 fp32 [rows,6], num int64 [1]) on the device: decode + clip + size test,
    NMS, gather"""
    _req(order, "order", torch.int32); _req(exp_wh, "exp_wh"); _req(loc, "loc"); _req(prob, "prob")
    n = order.numel()
    dev = loc.device
    rows = max_keep if max_keep > 0 else max(n, 1)
    out6 = torch.zeros(rows, 6, dtype=torch.float32, device=dev)
    num = torch.zeros(1, dtype=torch.int64, device=dev)
    if n == 0:
        return out6, num
    props = torch.empty(n, 5, dtype=torch.float32, device=dev)
    ok = torch.empty(n, dtype=torch.uint8, device=dev)
    L = lib()
    _check(L.scda_proposal_decode_hip(_p(order), _p(exp_wh), i32(n), _p(anchors64), _p(loc), _p(prob), i32(A), i32(fh), i32(fw),
                                      ctypes.c_double(img_h), ctypes.c_double(img_w), ctypes.c_double(min_size), _p(props), _p(ok),
                                      _stream()), "scda_proposal_decode_hip")
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    ws = torch.empty(max(L.scda_nms_workspace_bytes(i32(n)), 8), dtype=torch.uint8, device=dev)
    _check(L.scda_nms_valid_hip(_p(props), _p(ok), i32(n), f32(nms_thresh), _p(ws), _p(keep), _p(num), i32(max_keep), _stream()),
           "scda_nms_valid_hip")
    _check(L.scda_proposal_gather_hip(_p(props), _p(keep), _p(num), f32(image_index), i32(rows), _p(out6), _stream()),
           "scda_proposal_gather_hip")
    return out6, num


# ------------------------------------------------- convolution / GEMM -------
_WS = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer (split-K slabs), one per (device, stream): kernels on different HIP streams may run
    concurrently and must not share slabs.  Never freed during a run."""
    key = (device.index, _stream().value)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
_sz = ctypes.c_size_t


def _conv_ws(batch, cin, ih, iw, cout, kh, kw, s, p, device):
    L = lib()
    L.scda_conv2d_workspace_bytes.restype = ctypes.c_size_t
    n = L.scda_conv2d_workspace_bytes(i32(batch), i32(cin), i32(ih), i32(iw), i32(cout), i32(kh), i32(kw), i32(s), i32(p))
    return workspace(n, device), n


WEIGHT_EPOCH = [0]   # fallback epoch for weights that are not part of a FlatParams bucket (bump after raw-pointer updates)
_PACK_CACHE = {}


def conv2d_pack_weight(w, for_dgrad=False, cache=True):
    """[Cout,Cin,KH,KW] -> the library's GEMM-ready layout for the forward / data-gradient kernel (see scda_ops.h).
    Cached per (storage, direction) until the weight changes (tensor version or optimiser epoch)."""
    _req(w, "w")
    Cout, Cin, KH, KW = w.shape
    # only parameters are worth caching: a temporary (e.g. the transposed 1x1 weight of ConvTranspose1x1) gets a new address
    # on every call and would leave a dead entry behind each time
    cache = cache and isinstance(w, torch.nn.Parameter)
    key = (w.data_ptr(), for_dgrad)
    flat = getattr(w, "_scda_flat", None)
    tag = (w._version, flat.epoch if flat is not None else WEIGHT_EPOCH[0], tuple(w.shape))
    if cache:
        hit = _PACK_CACHE.get(key)
        # valid only for the very same tensor object (a freed temporary's address may be reused by another weight)
        if hit is not None and hit[0] == tag and hit[2]() is w:
            return hit[1]
    lib().scda_conv2d_packed_elems.restype = ctypes.c_size_t
    n = lib().scda_conv2d_packed_elems(i32(Cout), i32(Cin), i32(KH), i32(KW), i32(int(for_dgrad)))
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    _check(lib().scda_conv2d_pack_weight_hip(_p(w), _p(out), i32(Cout), i32(Cin), i32(KH), i32(KW), i32(int(for_dgrad)),
                                             _stream()), "scda_conv2d_pack_weight_hip")
    if cache:
        _PACK_CACHE[key] = (tag, out, weakref.ref(w))
    return out


def wino_enabled():
    """the stride-1 3x3 layers the Winograd kernel supports run on it, forward and data gradient (SCDA_WINOGRAD=0: every layer on
    the direct implicit-GEMM kernels)"""
    return os.environ.get("SCDA_WINOGRAD", "1") != "0"


def wino_min_channels():
    return int(os.environ.get("SCDA_WINOGRAD_MIN_C", "32"))


def wino_stacked(B, IH, IW, row_period):
    """the image is a stack of 7 x 7 maps (the ResNet-50 C4 detector's channel-major RoI head): -> number of maps, else 0"""
    if row_period == 7 and IW == 7 and B == 1 and IH % 7 == 0 and os.environ.get("SCDA_WINO_STACKED", "1") != "0":
        return IH // 7
    return 0


def wino_ok(B, Cin, IH, IW, Cout, KH, KW, stride, pad, row_period=0):
    """this convolution (forward: reduced channels Cin, output rows Cout; both at least wino_min_channels()) takes the Winograd path"""
    if not (wino_enabled() and KH == 3 and KW == 3 and stride == 1 and pad == 1):
        return False
    if min(Cin, Cout) < wino_min_channels():
        return False
    if row_period:
        maps = wino_stacked(B, IH, IW, row_period)
        return bool(maps and lib().scda_conv2d_wino_stacked_supported(i32(maps), i32(Cin), i32(Cout)))
    return bool(lib().scda_conv2d_wino_supported(i32(B), i32(Cin), i32(IH), i32(IW), i32(Cout)))


def wino_wgrad_ok(B, Cin, IH, IW, Cout, KH, KW, stride, pad, row_period=0):
    """this weight gradient takes the Winograd kernel (SCDA_WINOGRAD_WGRAD=0 keeps it on the direct one)"""
    if not (wino_enabled() and os.environ.get("SCDA_WINOGRAD_WGRAD", "1") != "0" and KH == 3 and KW == 3 and stride == 1 and pad == 1):
        return False
    if row_period:
        maps = wino_stacked(B, IH, IW, row_period)
        return bool(maps and lib().scda_conv2d_wino_wgrad_stacked_supported(i32(maps), i32(Cin), i32(Cout)))
    return bool(lib().scda_conv2d_wino_wgrad_supported(i32(B), i32(Cin), i32(IH), i32(IW), i32(Cout)))


def conv2d_wino_wgrad(dy, x, w_shape, out=None, db_out=None, want_bias=False, row_period=0):
    """(dw, db) of a stride-1 pad-1 3x3 convolution on the Winograd weight-gradient kernel; accumulates into out / db_out when given"""
    _req(dy, "dy"); _req(x, "x")
    B, Cin, IH, IW = x.shape
    Cout = w_shape[0]
    acc = dbacc = 0
    if out is None:
        out = torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device)
    else:
        _req(out, "out"); acc = 1
    db = None
    if want_bias or db_out is not None:
        if db_out is None:
            db = torch.empty(Cout, dtype=torch.float32, device=x.device)
        else:
            _req(db_out, "db_out"); db = db_out; dbacc = 1
    ws, n = _conv_ws(B, Cin, IH, IW, Cout, 3, 3, 1, 1, x.device)
    if row_period:
        maps = wino_stacked(B, IH, IW, row_period)
        if not maps:
            raise ValueError("conv2d_wino_wgrad: row_period %d on %s is not a stack of 7 x 7 maps" % (row_period, tuple(x.shape)))
        _check(lib().scda_conv2d_wino_wgrad_stacked_hip(_p(dy), _p(x), _p(out), _p(db), i32(maps), i32(Cin), i32(Cout), i32(acc), i32(dbacc),
                                                        _p(ws), _sz(n), _stream()), "scda_conv2d_wino_wgrad_stacked_hip")
        return out, db
    _check(lib().scda_conv2d_wino_wgrad_hip(_p(dy), _p(x), _p(out), _p(db), i32(B), i32(Cin), i32(IH), i32(IW), i32(Cout), i32(acc),
                                            i32(dbacc), _p(ws), _sz(n), _stream()), "scda_conv2d_wino_wgrad_hip")
    return out, db


def conv2d_wino_pack(w, for_dgrad=False, cache=True):
    """[Cout,Cin,3,3] -> the Winograd kernel's transformed filters (scda_ops.h), cached like conv2d_pack_weight's layouts"""
    _req(w, "w")
    Cout, Cin, KH, KW = w.shape
    cache = cache and isinstance(w, torch.nn.Parameter)
    key = (w.data_ptr(), 2 + int(for_dgrad))
    flat = getattr(w, "_scda_flat", None)
    tag = (w._version, flat.epoch if flat is not None else WEIGHT_EPOCH[0], tuple(w.shape))
    if cache:
        hit = _PACK_CACHE.get(key)
        if hit is not None and hit[0] == tag and hit[2]() is w:
            return hit[1]
    L = lib()
    L.scda_conv2d_wino_packed_elems.restype = ctypes.c_size_t
    n = L.scda_conv2d_wino_packed_elems(i32(Cout), i32(Cin), i32(int(for_dgrad)))
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    _check(L.scda_conv2d_wino_pack_hip(_p(w.contiguous()), _p(out), i32(Cout), i32(Cin), i32(int(for_dgrad)), _stream()), "scda_conv2d_wino_pack_hip")
    if cache:
        _PACK_CACHE[key] = (tag, out, weakref.ref(w))
    return out


def conv2d_wino(x, u, bias, M, act=ACT_NONE, slope=0.01, mask_src=None, mask_slope=0.0, for_dgrad=False, row_period=0):
    """one Winograd launch: y [B, M, H, W] from x [B, C, H, W] and packed filters u (forward, or the data gradient with x = dy);
    row_period = 7 on a [1, C, R * 7, 7] tensor: a stack of R independent 7 x 7 maps"""
    B, C, H, W = x.shape
    y = torch.empty(B, M, H, W, dtype=torch.float32, device=x.device)
    ws, n = _conv_ws(B, C, H, W, M, 3, 3, 1, 1, x.device)
    if row_period:
        maps = wino_stacked(B, H, W, row_period)
        if not maps:
            raise ValueError("conv2d_wino: row_period %d on %s is not a stack of 7 x 7 maps" % (row_period, tuple(x.shape)))
        _check(lib().scda_conv2d_wino_stacked_hip(_p(x), _p(u), _p(bias), _p(y), i32(maps), i32(C), i32(M), i32(act), f32(slope), _p(mask_src),
                                                  f32(mask_slope), i32(int(for_dgrad)), _p(ws), _sz(n), _stream()), "scda_conv2d_wino_stacked_hip")
        return y
    _check(lib().scda_conv2d_wino_hip(_p(x), _p(u), _p(bias), _p(y), i32(B), i32(C), i32(H), i32(W), i32(M), i32(act), f32(slope),
                                      _p(mask_src), f32(mask_slope), i32(int(for_dgrad)), _p(ws), _sz(n), _stream()), "scda_conv2d_wino_hip")
    return y


def conv_pool_fusable(B, Cin, IH, IW, Cout, KH, KW, stride, pad, row_period=0):
    """conv3x3 + activation + 2x2 max-pool can run as one Winograd launch: an eligible layer on an even map with enough tiles to fill the
    chip without split-K (the fused epilogue needs finished values).  SCDA_CONV_POOL_FUSE=0 keeps the pool a launch of its own."""
    if row_period or os.environ.get("SCDA_CONV_POOL_FUSE", "1") == "0" or not wino_ok(B, Cin, IH, IW, Cout, KH, KW, stride, pad, row_period):
        return False
    return ((Cout + 63) // 64) * B * ((IH + 7) // 8) * ((IW + 31) // 32) >= 200


def conv2d_wino_pool(x, u, bias, M, act=ACT_NONE, slope=0.01):
    """-> (pooled [B, M, H/2, W/2], winner uint8 [B, M, H/2, W/2]) of maxpool2x2(act(conv3x3(x) + bias)), one launch"""
    B, C, H, W = x.shape
    y = torch.empty(B, M, H // 2, W // 2, dtype=torch.float32, device=x.device)
    idx = torch.empty(B, M, H // 2, W // 2, dtype=torch.uint8, device=x.device)
    _check(lib().scda_conv2d_wino_pool_hip(_p(x), _p(u), _p(bias), _p(y), _p(idx), i32(B), i32(C), i32(H), i32(W), i32(M), i32(act),
                                           f32(slope), _stream()), "scda_conv2d_wino_pool_hip")
    return y, idx


def conv2d_pack_all(flat):
    """Re-pack every conv weight of a FlatParams bucket (forward and data-gradient layouts) with ONE launch and seed the
    pack cache with the results.  Called by FlatAdam.step(): the lazy per-layer path above then never misses in the
    training loop (it was 90 five-microsecond launches per iteration, each a dependent dispatch on the compute stream)."""
    ws = getattr(flat, "conv_weights", None)
    if not ws:
        return
    # which weights ran on the Winograd kernel in their last forward call (native.conv2d_fwd notes it): those get its transformed
    # filters (modes 2, 3) instead of the implicit-GEMM layouts (0, 1); a call that needs the other kind packs lazily
    wino = tuple(bool(getattr(w, "_scda_wino_used", False)) for w in ws) if wino_enabled() else None
    # one plan (descriptor table + output buffer) PER flag tuple, kept: with variable-size inputs a layer's eligibility flips with the
    # parity of the map size, and rebuilding the plan on every flip meant a descriptor upload and fresh buffers for every layout
    plans = flat.__dict__.setdefault("_scda_pack_plans", {})
    plan = plans.get(wino)
    L = lib()
    if plan is None:
        L.scda_conv2d_packed_elems.restype = ctypes.c_size_t
        L.scda_conv2d_pack_tiles.restype = ctypes.c_longlong
        rows, entries, off, tiles = [], [], 0, 0
        base = flat.data.data_ptr()
        for i, w in enumerate(ws):
            Cout, Cin, KH, KW = w.shape
            src = (w.data_ptr() - base) // 4
            use_wino = wino is not None and wino[i] and (KH, KW) == (3, 3) and Cin % 8 == 0 and Cout % 8 == 0
            for d in ((2, 3) if use_wino else (0, 1)):
                n = int(L.scda_conv2d_packed_elems(i32(Cout), i32(Cin), i32(KH), i32(KW), i32(d)))
                rows.append([src, off, Cout, Cin, KH * KW, d, tiles])
                entries.append((w, d if d >= 2 else bool(d), off, n))
                off += n
                tiles += int(L.scda_conv2d_pack_tiles(i32(Cout), i32(Cin), i32(KH), i32(KW), i32(d)))
        desc = upload(torch.tensor(rows, dtype=torch.int64), flat.data.device)
        while len(plans) >= 4:       # least recently used first (dicts keep insertion order; a hit re-inserts below)
            plans.pop(next(iter(plans)))
        plan = (desc, off, entries, tiles, wino)
    else:
        plans.pop(wino)
    plans[wino] = plan
    desc, total, entries, tiles, _ = plan
    # ONE output buffer for every plan, sized for the largest layout: a plan is a descriptor table, not a copy of the packed weights
    # (eight kept plans were eight such buffers -- several hundred MB for VGG16 with variable-size inputs).  Sharing is safe: this
    # function runs behind an optimiser step, whose epoch bump has already invalidated every cache entry of the previous layout.
    out = flat.__dict__.get("_scda_pack_out")
    if out is None or out.numel() < total:
        out = flat.__dict__["_scda_pack_out"] = torch.empty(total, dtype=torch.float32, device=flat.data.device)
    _check(L.scda_conv2d_pack_weights_batched_hip(_p(flat.data), _p(out), _p(desc), i32(len(entries)),
                                                  ctypes.c_longlong(tiles), _stream()), "scda_conv2d_pack_weights_batched_hip")
    for w, d, off, n in entries:
        if w.data_ptr() < flat.data.data_ptr():   # parameter was re-homed: fall back to the lazy path for it
            continue
        _PACK_CACHE[(w.data_ptr(), d)] = ((w._version, flat.epoch, tuple(w.shape)), out[off:off + n], weakref.ref(w))


def conv2d_fwd(x, w, bias, stride, pad, act=ACT_NONE, slope=0.01, row_period=0):
    _req(x, "x"); _req(w, "w")
    if bias is not None:
        _req(bias, "bias")
    B, Cin, IH, IW = x.shape
    Cout, Cin2, KH, KW = w.shape
    if Cin2 != Cin:
        raise ValueError(f"conv2d: input has {Cin} channels, weight expects {Cin2}")
    OH = (IH + 2 * pad - KH) // stride + 1
    OW = (IW + 2 * pad - KW) // stride + 1
    use_wino = wino_ok(B, Cin, IH, IW, Cout, KH, KW, stride, pad, row_period)
    if isinstance(w, torch.nn.Parameter):
        w._scda_wino_used = use_wino       # conv2d_pack_all re-packs the layouts the layer's calls actually use
    if use_wino:
        return conv2d_wino(x, conv2d_wino_pack(w, False), bias, Cout, act, slope, row_period=row_period)
    wp = conv2d_pack_weight(w, False)
    y = torch.empty(B, Cout, OH, OW, dtype=torch.float32, device=x.device)
    ws, n = _conv_ws(B, Cin, IH, IW, Cout, KH, KW, stride, pad, x.device)
    _check(lib().scda_conv2d_fwd_hip(_p(x), _p(wp), _p(bias), _p(y), i32(B), i32(Cin), i32(IH), i32(IW), i32(Cout), i32(KH),
                                     i32(KW), i32(stride), i32(pad), i32(row_period), i32(act), f32(slope), _p(ws), _sz(n), _stream()),
           "scda_conv2d_fwd_hip")
    return y


def conv2d_dgrad(dy, w, x_shape, stride, pad, act_src=None, act_slope=0.0, row_period=0):
    """act_src (the conv's input x, a ReLU / LeakyReLU output): dx is additionally multiplied by x > 0 ? 1 : act_slope -- the
    activation gradient of the layer that produced x, folded into this kernel's epilogue"""
    _req(dy, "dy"); _req(w, "w")
    B, Cin, IH, IW = x_shape
    Cout, _, KH, KW = w.shape
    if act_src is not None:
        _req(act_src, "act_src")
        if tuple(act_src.shape) != tuple(x_shape):
            raise ValueError("act_src must have the shape of the conv input")
    if wino_ok(B, Cout, IH, IW, Cin, KH, KW, stride, pad, row_period):
        return conv2d_wino(dy, conv2d_wino_pack(w, True), None, Cin, ACT_NONE, 0.0, act_src, act_slope, for_dgrad=True, row_period=row_period)
    dx = torch.empty(B, Cin, IH, IW, dtype=torch.float32, device=dy.device)
    if act_src is None and Cin <= 4 and Cout * KH * KW * 16 <= 65536 and (KH, KW) in ((3, 3), (1, 1)):
        # image-side layer: 3 rows of a 64-row MFMA tile would be 95 % padding -- direct kernel, unpacked weights
        _check(lib().scda_conv2d_dgrad_small_cin_hip(_p(dy), _p(w.contiguous()), _p(dx), i32(B), i32(Cin), i32(IH), i32(IW), i32(Cout),
                                                     i32(KH), i32(KW), i32(stride), i32(pad), _stream()),
               "scda_conv2d_dgrad_small_cin_hip")
        return dx
    wt = conv2d_pack_weight(w, True)
    ws, n = _conv_ws(B, Cin, IH, IW, Cout, KH, KW, stride, pad, dy.device)
    _check(lib().scda_conv2d_dgrad_act_hip(_p(dy), _p(wt), _p(dx), i32(B), i32(Cin), i32(IH), i32(IW), i32(Cout), i32(KH),
                                           i32(KW), i32(stride), i32(pad), i32(row_period), _p(act_src), f32(act_slope), _p(ws), _sz(n), _stream()),
           "scda_conv2d_dgrad_act_hip")
    return dx


def conv2d_wgrad(dy, x, w_shape, stride, pad, out=None, row_period=0):
    """dw = wgrad(dy, x); with `out` given, accumulates into it."""
    _req(dy, "dy"); _req(x, "x")
    B, Cin, IH, IW = x.shape
    Cout, _, KH, KW = w_shape
    if wino_wgrad_ok(B, Cin, IH, IW, Cout, KH, KW, stride, pad, row_period):
        return conv2d_wino_wgrad(dy, x, w_shape, out=out, row_period=row_period)[0]
    acc = 0
    if out is None:
        out = torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device)
    else:
        _req(out, "out"); acc = 1
    ws, n = _conv_ws(B, Cin, IH, IW, Cout, KH, KW, stride, pad, x.device)
    _check(lib().scda_conv2d_wgrad_hip(_p(dy), _p(x), _p(out), i32(B), i32(Cin), i32(IH), i32(IW), i32(Cout), i32(KH),
                                       i32(KW), i32(stride), i32(pad), i32(row_period), i32(acc), _p(ws), _sz(n), _stream()),
           "scda_conv2d_wgrad_hip")
    return out


def conv2d_wgrad_bias(dy, x, w_shape, stride, pad, out=None, db_out=None, row_period=0):
    """(dw, db): weight and bias gradient of a conv layer.  One fused pass (the bias gradient rides on the weight-gradient
    GEMM's operand fragments) when the library says the shape allows it, otherwise the two separate kernels.  `out` /
    `db_out` given: accumulate into them."""
    _req(dy, "dy"); _req(x, "x")
    B, Cin, IH, IW = x.shape
    Cout, _, KH, KW = w_shape
    if wino_wgrad_ok(B, Cin, IH, IW, Cout, KH, KW, stride, pad, row_period):
        return conv2d_wino_wgrad(dy, x, w_shape, out=out, db_out=db_out, want_bias=True, row_period=row_period)
    L = lib()
    if not L.scda_conv2d_wgrad_bias_fusable(i32(B), i32(Cout), i32(dy.shape[2]), i32(dy.shape[3]), _p(dy)):
        return conv2d_wgrad(dy, x, w_shape, stride, pad, out=out, row_period=row_period), bias_grad_nchw(dy, out=db_out)
    acc = dbacc = 0
    if out is None:
        out = torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device)
    else:
        _req(out, "out"); acc = 1
    if db_out is None:
        db_out = torch.empty(Cout, dtype=torch.float32, device=x.device)
    else:
        _req(db_out, "db_out"); dbacc = 1
    ws, n = _conv_ws(B, Cin, IH, IW, Cout, KH, KW, stride, pad, x.device)
    _check(L.scda_conv2d_wgrad_bias_hip(_p(dy), _p(x), _p(out), _p(db_out), i32(B), i32(Cin), i32(IH), i32(IW), i32(Cout),
                                        i32(KH), i32(KW), i32(stride), i32(pad), i32(row_period), i32(acc), i32(dbacc), _p(ws), _sz(n),
                                        _stream()), "scda_conv2d_wgrad_bias_hip")
    return out, db_out


def gemm(a, b, M, N, K, lda, ldb, trans_a=False, trans_b=False, bias=None, bias_on_n=True, act=ACT_NONE, slope=0.01,
         out=None, accumulate=False):
    """C[M,N] (+)= op(A) op(B) (+bias) -> act.  See include/scda_ops.h for the operand layouts."""
    _req(a, "a"); _req(b, "b")
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs out")
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    else:
        _req(out, "out")
    L = lib()
    L.scda_gemm_workspace_bytes.restype = ctypes.c_size_t
    n = L.scda_gemm_workspace_bytes(i32(M), i32(N), i32(K))
    ws = workspace(n, a.device)
    _check(L.scda_gemm_hip(_p(a), _p(b), _p(out), i32(M), i32(N), i32(K), i32(lda), i32(ldb), i32(N), i32(int(trans_a)),
                           i32(int(trans_b)), _p(bias), i32(int(bias_on_n)), i32(act), f32(slope), i32(int(accumulate)),
                           _p(ws), _sz(n), _stream()), "scda_gemm_hip")
    return out


def linear_fwd(x, w, bias, act=ACT_NONE):
    """y[M,out] = x[M,in] @ w[out,in]^T + b"""
    M, K = x.shape
    N = w.shape[0]
    return gemm(x, w, M, N, K, K, K, False, False, bias, True, act)


def linear_dgrad(dy, w):
    """dx[M,in] = dy[M,out] @ w[out,in]"""
    M, K = dy.shape
    N = w.shape[1]
    return gemm(dy, w, M, N, K, K, N, False, True)


def linear_wgrad(dy, x, out=None, accumulate=True):
    """dw[out,in] (+)= dy[M,out]^T @ x[M,in]; with `out`: accumulates into it unless accumulate=False (overwrite)"""
    Kb, M = dy.shape
    N = x.shape[1]
    return gemm(dy, x, M, N, Kb, M, N, True, True, out=out, accumulate=out is not None and accumulate)


# ------------------------------------------------------ layer kernels -------
i64 = ctypes.c_longlong
u64 = ctypes.c_uint64


def maxpool2x2_fwd(x):
    _req(x, "x")
    B, C, H, W = x.shape
    y = torch.empty(B, C, H // 2, W // 2, dtype=torch.float32, device=x.device)
    idx = torch.empty(B, C, H // 2, W // 2, dtype=torch.uint8, device=x.device)
    _check(lib().scda_maxpool2x2_fwd_hip(_p(x), _p(y), _p(idx), i32(B * C), i32(H), i32(W), _stream()), "scda_maxpool2x2_fwd_hip")
    return y, idx


def maxpool2x2_bwd(dy, idx, x_shape, relu_y=None):
    """relu_y = the pool's output: additionally applies the gradient of a ReLU that produced the pool's input"""
    _req(dy, "dy"); _req(idx, "idx", torch.uint8)
    B, C, H, W = x_shape
    dx = torch.empty(B, C, H, W, dtype=torch.float32, device=dy.device)
    if relu_y is None:
        _check(lib().scda_maxpool2x2_bwd_hip(_p(dy), _p(idx), _p(dx), i32(B * C), i32(H), i32(W), _stream()), "scda_maxpool2x2_bwd_hip")
    else:
        _req(relu_y, "relu_y")
        _check(lib().scda_maxpool2x2_bwd_relu_hip(_p(dy), _p(idx), _p(relu_y), _p(dx), i32(B * C), i32(H), i32(W), _stream()),
               "scda_maxpool2x2_bwd_relu_hip")
    return dx


def maxpool3x3s2_fwd(x):
    _req(x, "x")
    B, C, H, W = x.shape
    y = torch.empty(B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, dtype=torch.float32, device=x.device)
    _check(lib().scda_maxpool3x3s2_fwd_hip(_p(x), _p(y), i32(B * C), i32(H), i32(W), _stream()), "scda_maxpool3x3s2_fwd_hip")
    return y


def add_relu(a, b):
    _req(a, "a"); _req(b, "b")
    if a.shape != b.shape:
        raise ValueError("add_relu: shape mismatch")
    y = torch.empty_like(a)
    _check(lib().scda_add_relu_hip(_p(a), _p(b), _p(y), i64(a.numel()), _stream()), "scda_add_relu_hip")
    return y


ACT_MODE = {"relu": 0, "leaky": 1, "tanh": 2, "sigmoid": 3}


def act_fwd(x, mode, slope=0.01):
    _req(x, "x")
    y = torch.empty_like(x)
    _check(lib().scda_act_fwd_hip(_p(x), _p(y), i64(x.numel()), i32(mode), f32(slope), _stream()), "scda_act_fwd_hip")
    return y


def act_bwd(dy, y, mode, slope=0.01):
    _req(dy, "dy"); _req(y, "y")
    dx = torch.empty_like(dy)
    _check(lib().scda_act_bwd_hip(_p(dy), _p(y), _p(dx), i64(dy.numel()), i32(mode), f32(slope), _stream()), "scda_act_bwd_hip")
    return dx


def axpby(a, b, alpha=1.0, beta=1.0):
    _req(a, "a")
    if b is not None:
        _req(b, "b")
    y = torch.empty_like(a)
    _check(lib().scda_axpby_hip(_p(a), _p(b), _p(y), i64(a.numel()), f32(alpha), f32(beta), _stream()), "scda_axpby_hip")
    return y


def dropout_mask(shape, p, seed, device):
    mask = torch.empty(shape, dtype=torch.uint8, device=device)
    _check(lib().scda_dropout_mask_hip(_p(mask), i64(mask.numel()), f32(p), u64(seed & 0xFFFFFFFFFFFFFFFF), _stream()),
           "scda_dropout_mask_hip")
    return mask


def dropout_apply(x, mask, scale):
    _req(x, "x"); _req(mask, "mask", torch.uint8)
    y = torch.empty_like(x)
    _check(lib().scda_dropout_apply_hip(_p(x), _p(mask), _p(y), i64(x.numel()), f32(scale), _stream()), "scda_dropout_apply_hip")
    return y


def dropout_seeded(x, p, seed, scale, relu_src=None):
    """y = keep(seed, i) ? x * scale : 0 (no mask tensor; the backward calls this again with dy); relu_src: see scda_ops.h"""
    _req(x, "x")
    if relu_src is not None:
        _req(relu_src, "relu_src")
    y = torch.empty_like(x)
    _check(lib().scda_dropout_seeded_hip(_p(x), _p(y), i64(x.numel()), f32(p), u64(seed & 0xFFFFFFFFFFFFFFFF), f32(scale),
                                         _p(relu_src), _stream()), "scda_dropout_seeded_hip")
    return y


def sigmoid_bce_rows_fwd(x, t, w, scale, out=None, want_prob=True):
    """out[0] (+)= scale * sum_c w[c] * mean_i BCE(sigmoid(x[c,i]), t[c or 0, i]); -> (out [1], prob [C,n] | None)"""
    _req(x, "x"); _req(t, "t")
    C, n = x.shape
    t_rows = t.numel() // n
    if t.numel() != t_rows * n or t_rows not in (1, C):
        raise ValueError("labels must be [1,n] or [C,n]")
    if w is not None:
        _req(w, "w")
    acc = out is not None
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=x.device)
    prob = torch.empty_like(x) if want_prob else None
    _check(lib().scda_sigmoid_bce_rows_fwd_hip(_p(x), _p(t), i32(t_rows), _p(w), i32(C), i32(n), f32(scale), i32(int(acc)), _p(prob),
                                               _p(out), _stream()), "scda_sigmoid_bce_rows_fwd_hip")
    return out, prob


def sigmoid_bce_rows_bwd(prob, t, w, scale, g):
    C, n = prob.shape
    dx = torch.empty_like(prob)
    _check(lib().scda_sigmoid_bce_rows_bwd_hip(_p(prob), _p(t), i32(t.numel() // n), _p(w), i32(C), i32(n), f32(scale), _p(g), _p(dx),
                                               _stream()), "scda_sigmoid_bce_rows_bwd_hip")
    return dx


def bias_grad_nchw(dy, out=None):
    """db[c] = sum over batch and pixels; with `out` given, accumulates into it"""
    _req(dy, "dy")
    B, C = dy.shape[0], dy.shape[1]
    HW = dy.numel() // (B * C)
    db = out if out is not None else torch.empty(C, dtype=torch.float32, device=dy.device)
    L = lib()
    L.scda_bias_grad_workspace_bytes.restype = ctypes.c_size_t
    ws = torch.empty(L.scda_bias_grad_workspace_bytes(i32(C)) // 4, dtype=torch.float32, device=dy.device)
    _check(L.scda_bias_grad_nchw_hip(_p(dy), _p(db), i32(B), i32(C), i32(HW), i32(0 if out is None else 1), _p(ws), _stream()),
           "scda_bias_grad_nchw_hip")
    return db


def colsum(dy, out=None):
    _req(dy, "dy")
    M, N = dy.shape
    db = out if out is not None else torch.empty(N, dtype=torch.float32, device=dy.device)
    _check(lib().scda_colsum_hip(_p(dy), _p(db), i32(M), i32(N), i32(0 if out is None else 1), _stream()), "scda_colsum_hip")
    return db


def softmax_ce_fwd(logits, targets, ignore_index=-100):
    _req(logits, "logits"); _req(targets, "targets", torch.int64)
    R, C = logits.shape
    probs = torch.empty_like(logits)
    out2 = torch.empty(2, dtype=torch.float32, device=logits.device)
    _check(lib().scda_softmax_ce_fwd_hip(_p(logits), _p(targets), i32(R), i32(C), i32(ignore_index), _p(probs), _p(out2), _stream()),
           "scda_softmax_ce_fwd_hip")
    return out2, probs


def softmax_ce_bwd(probs, targets, out2, g, ignore_index=-100):
    _req(probs, "probs"); _req(g, "g")
    R, C = probs.shape
    dx = torch.empty_like(probs)
    _check(lib().scda_softmax_ce_bwd_hip(_p(probs), _p(targets), i32(R), i32(C), i32(ignore_index), _p(out2), _p(g), _p(dx), _stream()),
           "scda_softmax_ce_bwd_hip")
    return dx


def row_softmax(x):
    _req(x, "x")
    R, C = x.shape
    y = torch.empty_like(x)
    _check(lib().scda_row_softmax_hip(_p(x), _p(y), i32(R), i32(C), _stream()), "scda_row_softmax_hip")
    return y


def accuracy(logits, targets, ignore_index=-1):
    _req(logits, "logits"); _req(targets, "targets", torch.int64)
    R, C = logits.shape
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    _check(lib().scda_accuracy_hip(_p(logits), _p(targets), i32(R), i32(C), i32(ignore_index), _p(out), _stream()), "scda_accuracy_hip")
    return out


def smooth_l1_fwd(pred, mask, target, sigma, scale):
    _req(pred, "pred"); _req(target, "target")
    if mask is not None:
        _req(mask, "mask")
    L = lib()
    L.scda_smooth_l1_workspace_bytes.restype = ctypes.c_size_t
    ws = torch.empty(L.scda_smooth_l1_workspace_bytes() // 4, dtype=torch.float32, device=pred.device)
    out = torch.empty(1, dtype=torch.float32, device=pred.device)
    _check(L.scda_smooth_l1_fwd_hip(_p(pred), _p(mask), _p(target), i64(pred.numel()), f32(sigma), f32(scale), _p(ws), _p(out), _stream()),
           "scda_smooth_l1_fwd_hip")
    return out


def smooth_l1_bwd(pred, mask, target, sigma, scale, g):
    dp = torch.empty_like(pred)
    _check(lib().scda_smooth_l1_bwd_hip(_p(pred), _p(mask), _p(target), i64(pred.numel()), f32(sigma), f32(scale), _p(g), _p(dp), _stream()),
           "scda_smooth_l1_bwd_hip")
    return dp


def instnorm_fwd(x, eps, act, slope):
    _req(x, "x")
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B * C, dtype=torch.float32, device=x.device)
    _check(lib().scda_instnorm_fwd_hip(_p(x), _p(y), _p(mean), _p(rstd), i32(B * C), i32(H * W), f32(eps), i32(act), f32(slope), _stream()),
           "scda_instnorm_fwd_hip")
    return y, mean, rstd


def instnorm_bwd(dy, x, mean, rstd, act, slope):
    _req(dy, "dy"); _req(x, "x")
    B, C, H, W = x.shape
    dx = torch.empty_like(x)
    _check(lib().scda_instnorm_bwd_hip(_p(dy), _p(x), _p(mean), _p(rstd), _p(dx), i32(B * C), i32(H * W), i32(act), f32(slope), _stream()),
           "scda_instnorm_bwd_hip")
    return dx


def instnorm_drop_add_fwd(x, residual, eps, p, seed):
    """residual + dropout_{p,seed}(instance_norm(x)) in one launch -> (y, mean, rstd)"""
    _req(x, "x"); _req(residual, "residual")
    if residual.shape != x.shape:
        raise ValueError("residual must have the shape of x")
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B * C, dtype=torch.float32, device=x.device)
    if torch.is_tensor(seed):     # a slot of a scda_amd.seeds.SeedArena: the kernel reads the seed from device memory
        _req(seed, "seed", torch.int64)
        _check(lib().scda_instnorm_drop_add_fwd_dev_hip(_p(x), _p(residual), _p(y), _p(mean), _p(rstd), i32(B * C), i32(H * W), f32(eps),
                                                        f32(p), _p(seed), f32(1.0 / (1.0 - p)), _stream()), "scda_instnorm_drop_add_fwd_dev_hip")
        return y, mean, rstd
    _check(lib().scda_instnorm_drop_add_fwd_hip(_p(x), _p(residual), _p(y), _p(mean), _p(rstd), i32(B * C), i32(H * W), f32(eps), f32(p),
                                                u64(seed & 0xFFFFFFFFFFFFFFFF), f32(1.0 / (1.0 - p)), _stream()), "scda_instnorm_drop_add_fwd_hip")
    return y, mean, rstd


def instnorm_drop_bwd(dy, x, mean, rstd, p, seed):
    _req(dy, "dy"); _req(x, "x")
    B, C, H, W = x.shape
    dx = torch.empty_like(x)
    if torch.is_tensor(seed):
        _check(lib().scda_instnorm_drop_bwd_dev_hip(_p(dy), _p(x), _p(mean), _p(rstd), _p(dx), i32(B * C), i32(H * W), f32(p), _p(seed),
                                                    f32(1.0 / (1.0 - p)), _stream()), "scda_instnorm_drop_bwd_dev_hip")
        return dx
    _check(lib().scda_instnorm_drop_bwd_hip(_p(dy), _p(x), _p(mean), _p(rstd), _p(dx), i32(B * C), i32(H * W), f32(p),
                                            u64(seed & 0xFFFFFFFFFFFFFFFF), f32(1.0 / (1.0 - p)), _stream()), "scda_instnorm_drop_bwd_dev_hip")
    return dx


def _bn_ws(B, C, HW, device):
    L = lib()
    L.scda_batchnorm_workspace_bytes.restype = ctypes.c_size_t
    n = L.scda_batchnorm_workspace_bytes(i32(B), i32(C), i32(HW))
    return torch.empty(n // 4, dtype=torch.float32, device=device) if n else None


def batchnorm_fwd(x, gamma, beta, run_mean, run_var, eps, momentum, act, slope):
    _req(x, "x"); _req(gamma, "gamma"); _req(beta, "beta")
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = _bn_ws(B, C, H * W, x.device)
    _check(lib().scda_batchnorm_fwd_hip(_p(x), _p(y), _p(gamma), _p(beta), _p(run_mean), _p(run_var), _p(mean), _p(rstd), i32(B), i32(C),
                                        i32(H * W), f32(eps), f32(momentum), i32(act), f32(slope), _p(ws), _stream()), "scda_batchnorm_fwd_hip")
    return y, mean, rstd


def batchnorm_bwd(dy, x, gamma, beta, mean, rstd, act, slope, need_dx=True, out=None):
    """out = (dgamma, dbeta) buffers to accumulate into, or None to allocate"""
    _req(dy, "dy"); _req(x, "x")
    B, C, H, W = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dg, db = out if out is not None else (torch.empty(C, dtype=torch.float32, device=x.device),
                                          torch.empty(C, dtype=torch.float32, device=x.device))
    _check(lib().scda_batchnorm_bwd_hip(_p(dy), _p(x), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), i32(B), i32(C),
                                        i32(H * W), i32(act), f32(slope), i32(0 if out is None else 1), _p(_bn_ws(B, C, H * W, x.device)),
                                        _stream()), "scda_batchnorm_bwd_hip")
    return dx, dg, db


def batchnorm_add_relu_ok(x):
    """is relu(bn(x) + residual) served as one kernel for this map?  (batch 1, plane of a multiple of 4 up to 40960 elements)"""
    B, C, H, W = x.shape
    return bool(lib().scda_batchnorm_add_relu_ok(i32(B), i32(H * W)))


def aligned16(*tensors):
    """the 16-byte alignment the plane kernels' float4 accesses need (a contiguous but OFFSET view -- a sliced residual, a dy that
    autograd hands over as a storage-offset view -- is contiguous and still not aligned)"""
    return all(t.data_ptr() % 16 == 0 for t in tensors)


def batchnorm_add_relu_fwd(x, residual, gamma, beta, run_mean, run_var, eps, momentum):
    _req(x, "x"); _req(residual, "residual"); _req(gamma, "gamma"); _req(beta, "beta")
    if residual.shape != x.shape:
        raise ValueError("batchnorm_add_relu_fwd: shape mismatch")
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    _check(lib().scda_batchnorm_add_relu_fwd_hip(_p(x), _p(residual), _p(y), _p(gamma), _p(beta), _p(run_mean), _p(run_var), _p(mean),
                                                 _p(rstd), i32(B), i32(C), i32(H * W), f32(eps), f32(momentum), _stream()),
           "scda_batchnorm_add_relu_fwd_hip")
    return y, mean, rstd


def batchnorm_add_relu_bwd(dy, x, y, gamma, beta, mean, rstd, need_dx=True, out=None):
    """-> (dx, d_residual, dgamma, dbeta); out = (dgamma, dbeta) buffers to accumulate into, or None to allocate"""
    _req(dy, "dy"); _req(x, "x"); _req(y, "y")
    B, C, H, W = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dres = torch.empty_like(x)
    dg, db = out if out is not None else (torch.empty(C, dtype=torch.float32, device=x.device),
                                          torch.empty(C, dtype=torch.float32, device=x.device))
    _check(lib().scda_batchnorm_add_relu_bwd_hip(_p(dy), _p(x), _p(y), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dx), _p(dres), _p(dg),
                                                 _p(db), i32(B), i32(C), i32(H * W), i32(0 if out is None else 1), _stream()),
           "scda_batchnorm_add_relu_bwd_hip")
    return dx, dres, dg, db


def batchnorm_eval(x, gamma, beta, run_mean, run_var, eps, act, slope, dy=None):
    """eval-mode batch norm (+act); with dy: the gradient w.r.t. x"""
    _req(x, "x"); _req(gamma, "gamma"); _req(beta, "beta"); _req(run_mean, "running_mean"); _req(run_var, "running_var")
    if dy is not None:
        _req(dy, "dy")
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    _check(lib().scda_batchnorm_eval_hip(_p(x), _p(dy), _p(out), _p(gamma), _p(beta), _p(run_mean), _p(run_var), i32(B), i32(C),
                                         i32(H * W), f32(eps), i32(act), f32(slope), _stream()), "scda_batchnorm_eval_hip")
    return out


def upsample2x_fwd(x):
    _req(x, "x")
    B, C, H, W = x.shape
    y = torch.empty(B, C, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
    _check(lib().scda_upsample2x_fwd_hip(_p(x), _p(y), i32(B * C), i32(H), i32(W), _stream()), "scda_upsample2x_fwd_hip")
    return y


def instnorm_up2_ok(x):
    """instance norm + the bilinear x2 behind it can run as one launch on this tensor (scda_instnorm_up2_supported + alignment)"""
    return (x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and aligned16(x)
            and bool(lib().scda_instnorm_up2_supported(i32(x.shape[2]), i32(x.shape[3]))))


def instnorm_up2_fwd(x, eps, act, slope):
    """upsample2x(act(instance_norm(x))) in one launch -> (y2 [B, C, 2H, 2W], mean, rstd)"""
    _req(x, "x")
    B, C, H, W = x.shape
    y2 = torch.empty(B, C, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
    mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B * C, dtype=torch.float32, device=x.device)
    _check(lib().scda_instnorm_up2_fwd_hip(_p(x), _p(y2), _p(mean), _p(rstd), i32(B * C), i32(H), i32(W), f32(eps), i32(act), f32(slope),
                                           _stream()), "scda_instnorm_up2_fwd_hip")
    return y2, mean, rstd


def instnorm_drop_add_up2_fwd(x, residual, eps, p, seed):
    """upsample2x(residual + dropout_{p,seed}(instance_norm(x))) in one launch -> (y2, mean, rstd)"""
    _req(x, "x"); _req(residual, "residual")
    if residual.shape != x.shape:
        raise ValueError("residual must have the shape of x")
    B, C, H, W = x.shape
    y2 = torch.empty(B, C, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
    mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B * C, dtype=torch.float32, device=x.device)
    if torch.is_tensor(seed):     # a slot of a scda_amd.seeds.SeedArena (see instnorm_drop_add_fwd)
        _req(seed, "seed", torch.int64)
        _check(lib().scda_instnorm_drop_add_up2_fwd_dev_hip(_p(x), _p(residual), _p(y2), _p(mean), _p(rstd), i32(B * C), i32(H), i32(W),
                                                            f32(eps), f32(p), _p(seed), f32(1.0 / (1.0 - p)), _stream()),
               "scda_instnorm_drop_add_up2_fwd_dev_hip")
        return y2, mean, rstd
    _check(lib().scda_instnorm_drop_add_up2_fwd_hip(_p(x), _p(residual), _p(y2), _p(mean), _p(rstd), i32(B * C), i32(H), i32(W), f32(eps),
                                                    f32(p), u64(seed & 0xFFFFFFFFFFFFFFFF), f32(1.0 / (1.0 - p)), _stream()),
           "scda_instnorm_drop_add_up2_fwd_hip")
    return y2, mean, rstd


def instnorm_up2_bwd(dy2, x, mean, rstd, act, slope):
    """gradient of instnorm_up2_fwd w.r.t. x: the bilinear gather and the norm's backward in one launch"""
    _req(dy2, "dy2"); _req(x, "x")
    B, C, H, W = x.shape
    dx = torch.empty_like(x)
    _check(lib().scda_instnorm_up2_bwd_hip(_p(dy2), _p(x), _p(mean), _p(rstd), _p(dx), i32(B * C), i32(H), i32(W), i32(act), f32(slope),
                                           _stream()), "scda_instnorm_up2_bwd_hip")
    return dx


def instnorm_drop_up2_bwd(dy2, x, mean, rstd, p, seed):
    """gradients of instnorm_drop_add_up2_fwd -> (dx, dresidual): dresidual = the gathered gradient of the small plane"""
    _req(dy2, "dy2"); _req(x, "x")
    B, C, H, W = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x)
    if torch.is_tensor(seed):
        _check(lib().scda_instnorm_drop_up2_bwd_dev_hip(_p(dy2), _p(x), _p(mean), _p(rstd), _p(dx), _p(dres), i32(B * C), i32(H), i32(W), f32(p),
                                                        _p(seed), f32(1.0 / (1.0 - p)), _stream()), "scda_instnorm_drop_up2_bwd_dev_hip")
        return dx, dres
    _check(lib().scda_instnorm_drop_up2_bwd_hip(_p(dy2), _p(x), _p(mean), _p(rstd), _p(dx), _p(dres), i32(B * C), i32(H), i32(W), f32(p),
                                                u64(seed & 0xFFFFFFFFFFFFFFFF), f32(1.0 / (1.0 - p)), _stream()), "scda_instnorm_drop_up2_bwd_hip")
    return dx, dres


def upsample2x_bwd(dy):
    _req(dy, "dy")
    B, C, OH, OW = dy.shape
    dx = torch.empty(B, C, OH // 2, OW // 2, dtype=torch.float32, device=dy.device)
    _check(lib().scda_upsample2x_bwd_hip(_p(dy), _p(dx), i32(B * C), i32(OH // 2), i32(OW // 2), _stream()), "scda_upsample2x_bwd_hip")
    return dx


def bce_fwd(p, t):
    _req(p, "p"); _req(t, "t")
    if p.numel() != t.numel():
        raise ValueError("bce: shape mismatch")
    out = torch.empty(1, dtype=torch.float32, device=p.device)
    _check(lib().scda_bce_fwd_hip(_p(p), _p(t), i32(p.numel()), _p(out), _stream()), "scda_bce_fwd_hip")
    return out


def bce_bwd(p, t, g):
    dp = torch.empty_like(p)
    _check(lib().scda_bce_bwd_hip(_p(p), _p(t), i32(p.numel()), _p(g), _p(dp), _stream()), "scda_bce_bwd_hip")
    return dp


def avg2x2s1_fwd(x):
    _req(x, "x")
    B, C, H1, W1 = x.shape
    y = torch.empty(B, C, H1 - 1, W1 - 1, dtype=torch.float32, device=x.device)
    _check(lib().scda_avg2x2s1_fwd_hip(_p(x), _p(y), i32(B * C), i32(H1 - 1), i32(W1 - 1), _stream()), "scda_avg2x2s1_fwd_hip")
    return y


def avg2x2s1_bwd(dy):
    _req(dy, "dy")
    B, C, H, W = dy.shape
    dx = torch.empty(B, C, H + 1, W + 1, dtype=torch.float32, device=dy.device)
    _check(lib().scda_avg2x2s1_bwd_hip(_p(dy), _p(dx), i32(B * C), i32(H), i32(W), _stream()), "scda_avg2x2s1_bwd_hip")
    return dx


def gap_fwd(x):
    _req(x, "x")
    B, C, H, W = x.shape
    y = torch.empty(B, C, dtype=torch.float32, device=x.device)
    _check(lib().scda_gap_fwd_hip(_p(x), _p(y), i32(B * C), i32(H * W), _stream()), "scda_gap_fwd_hip")
    return y


def gap_bwd(dy, x_shape):
    _req(dy, "dy")
    B, C, H, W = x_shape
    dx = torch.empty(B, C, H, W, dtype=torch.float32, device=dy.device)
    _check(lib().scda_gap_bwd_hip(_p(dy), _p(dx), i32(B * C), i32(H * W), _stream()), "scda_gap_bwd_hip")
    return dx


def row_mean(x):
    _req(x, "x")
    R, C = x.shape
    y = torch.empty(R, dtype=torch.float32, device=x.device)
    _check(lib().scda_row_mean_hip(_p(x), _p(y), i32(R), i32(C), _stream()), "scda_row_mean_hip")
    return y


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, max_blocks=0):
    _req(param, "param"); _req(grad, "grad"); _req(exp_avg, "exp_avg"); _req(exp_avg_sq, "exp_avg_sq")
    _check(lib().scda_adam_limited_hip(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), i64(param.numel()), f32(lr), f32(beta1),
                                       f32(beta2), f32(eps), f32(weight_decay), i32(step), i32(max_blocks), _stream()), "scda_adam_hip")


def last_plan():
    """(tile rows, tile cols, split-K count, direct-to-LDS?) of this thread's most recent conv / GEMM launch; the last entry is the
    integer 2 (truthy) when the dense GEMM ran as the exact-product bf16 x 9 kernel"""
    out = (ctypes.c_int * 4)()
    lib().scda_debug_last_plan(out)
    return out[0], out[1], out[2], (2 if out[3] == 2 else bool(out[3]))


def wino_last_persistent():
    """was this thread's most recent Winograd forward / data-gradient launch the persistent form (one workgroup per CU walking the tiles)?"""
    return bool(lib().scda_debug_wino_last_persistent())


def wino_last_order():
    """launch order of this thread's most recent Winograd launches: ((tile rows / 32, pixel-block-major?, gm, splits), (wgrad splits, wgrad order))"""
    out = (ctypes.c_int * 6)()
    lib().scda_debug_wino_last_order(out)
    return (out[0], bool(out[1]), out[2], out[3]), (out[4], out[5])


# ------------------------------------------------------------ profiler ------
def prof_kernel_names():
    L = lib()
    L.scda_prof_kernel_name.restype = ctypes.c_char_p
    return [L.scda_prof_kernel_name(i32(k)).decode() for k in range(L.scda_prof_num_kernels())]


def prof_enable(kernels):
    """kernels: False/None = off, True = all GEMM-class kernels, or an iterable of kernel names to time"""
    names = prof_kernel_names()
    if not kernels:
        mask = 0
    elif kernels is True:
        mask = (1 << len(names)) - 1
    else:
        mask = 0
        for k in kernels:
            mask |= 1 << names.index(k)
    lib().scda_prof_enable(ctypes.c_uint(mask))


def prof_collect():
    """after torch.cuda.synchronize(): {kernel name: (launches, total_ms, total_flops, total_algorithmic_bytes)} for the
    kernel classes that ran"""
    L = lib()
    L.scda_prof_kernel_name.restype = ctypes.c_char_p
    n = L.scda_prof_num_kernels()
    launches = (ctypes.c_longlong * n)()
    ms = (ctypes.c_double * n)()
    fl = (ctypes.c_double * n)()
    by = (ctypes.c_double * n)()
    _check(L.scda_prof_collect(launches, ms, fl, by), "scda_prof_collect")
    return {L.scda_prof_kernel_name(i32(k)).decode(): (int(launches[k]), float(ms[k]), float(fl[k]), float(by[k]))
            for k in range(n) if launches[k] > 0}


# ------------------------------------------------------------ data path -----
def image_resize_normalize(src, tables, out_h, out_w, normalize=True, mean=0.5, std=0.5, flip=False):
    """src uint8 [H, W, C] (device) -> float32 [C, out_h, out_w]: PIL resize + optional flip + ToTensor + Normalize on the device
    (scda_image_resize_normalize_hip; datasets/example_dataset.py:76-131).  `tables` = (bounds_h, kk_h, ksize_h, bounds_v, kk_v,
    ksize_v, row0, rows): device int32 tensors and ints from device_image.resize_tables."""
    _req(src, "src", torch.uint8)
    H, W, C = src.shape
    bh, kh, ksh, bv, kv, ksv, row0, rows = tables
    for t, name in ((bh, "bounds_h"), (kh, "kk_h"), (bv, "bounds_v"), (kv, "kk_v")):
        _req(t, name, torch.int32)
    L = lib()
    L.scda_image_resize_tmp_bytes.restype = ctypes.c_size_t
    nb = int(L.scda_image_resize_tmp_bytes(i32(rows), i32(out_w), i32(C)))
    tmp = torch.empty(nb, dtype=torch.uint8, device=src.device)
    out = torch.empty(C, out_h, out_w, dtype=torch.float32, device=src.device)
    _check(L.scda_image_resize_normalize_hip(_p(src), i32(H), i32(W), i32(C), _p(bh), _p(kh), i32(ksh), i32(out_w), _p(bv), _p(kv),
                                             i32(ksv), i32(out_h), i32(row0), i32(rows), _p(tmp), ctypes.c_size_t(nb),
                                             i32(1 if normalize else 0), f32(mean), f32(std), i32(1 if flip else 0), _p(out), _stream()),
           "scda_image_resize_normalize_hip")
    return out


# ------------------------------------------------------- host -> device ------
def upload(array_or_tensor, device, dtype=None):
    """numpy array / CPU tensor -> device tensor through a pinned staging buffer with a non-blocking copy.
    (`tensor.to(device)` from pageable memory is a synchronous, stream-ordered hipMemcpy: the host would sit behind
    every kernel already queued on the stream.)"""
    t = array_or_tensor if isinstance(array_or_tensor, torch.Tensor) else torch.from_numpy(array_or_tensor)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    pin = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    pin.copy_(t)
    return pin.to(device, non_blocking=True)
