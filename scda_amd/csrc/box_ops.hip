// box_ops.hip -- the detector's box logic on the device: what the reference does in numpy on the
// host between the RPN and the RCNN head (functions/anchor_target.py, functions/rpn_proposal.py), as HIP kernels that keep
// the tensors in HBM.  The host keeps exactly two jobs, because their results are part of the reference's observable
// behaviour: drawing random numbers from numpy's global generator (so it needs two COUNTS back from the device), and ranking
// RPN scores with numpy's own argpartition / argsort (whose tie order is numpy's).
//
// Exact arithmetic (compiled with -ffp-contract=off): IoU as extensions/_cython_bbox/cython_bbox.pyx:32-73 in fp32; box
// encode / decode as utils/bbox_helper.py:70-103 in float64 (numpy promotes to float64 there), rounded to fp32 where the
// reference stores into a float32 array.
#include <math.h>

#include "common.h"

namespace scda {

// IoU of cython_bbox.pyx:32-73 (no +1; 0 unless iw > 0 and ih > 0), fp32, one IEEE operation per source operator
__device__ __forceinline__ float bbox_iou(const float *b, const float *q) {
    const float box_area = (q[2] - q[0]) * (q[3] - q[1]);
    const float iw = fminf(b[2], q[2]) - fmaxf(b[0], q[0]);
    if (!(iw > 0)) return 0.f;
    const float ih = fminf(b[3], q[3]) - fmaxf(b[1], q[1]);
    if (!(ih > 0)) return 0.f;
    const float ua = (b[2] - b[0]) * (b[3] - b[1]) + box_area - iw * ih;
    return __fdiv_rn(iw * ih, ua);
}

// ---- anchor labelling: functions/anchor_target.py:38-64 -----------------------------------------------------------------
// pass 1: per anchor the best gt (first maximum, as ndarray.argmax) and its IoU; per gt the best IoU over all anchors
// (block-level max in LDS, then one atomicMax per block and gt on the IoU's bit pattern -- IoUs are >= 0, so unsigned order
// is float order)
__global__ __launch_bounds__(256) void anchor_match_kernel(const float *__restrict__ anchors, const int KA,
                                                           const float *__restrict__ gts, const int G, const int gt_stride,
                                                           float *__restrict__ best_iou, int *__restrict__ best_gt,
                                                           unsigned *__restrict__ gt_best_bits) {
    extern __shared__ unsigned sbest[];   // [G]
    for (int g = threadIdx.x; g < G; g += blockDim.x) sbest[g] = 0u;
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < KA) {
        const float *a = anchors + (size_t)k * 4;
        float bi = -1.f;
        int bg = 0;
        for (int g = 0; g < G; ++g) {
            const float v = bbox_iou(a, gts + (size_t)g * gt_stride);
            if (v > bi) { bi = v; bg = g; }
            if (v > 0.f) atomicMax(&sbest[g], __float_as_uint(v));
        }
        best_iou[k] = bi;
        best_gt[k] = bg;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += blockDim.x)
        if (sbest[g]) atomicMax(&gt_best_bits[g], sbest[g]);
}

// pass 2: labels.  -1 ignore / 0 background / 1 foreground, in the reference's order of assignments:
//   labels[best_iou < neg] = 0 ; labels[(anchor, gt) pairs whose IoU equals that gt's best, if that best >= 0.1] = 1 (and
//   best_gt <- that gt, the LAST such gt in np.where's row-major order) ; labels[best_iou > pos] = 1
__global__ __launch_bounds__(256) void anchor_label_kernel(const float *__restrict__ anchors, const int KA,
                                                           const float *__restrict__ gts, const int G, const int gt_stride,
                                                           const float *__restrict__ best_iou, int *__restrict__ best_gt,
                                                           const unsigned *__restrict__ gt_best_bits, const float neg_thresh,
                                                           const float pos_thresh, const float min_gt_best,
                                                           signed char *__restrict__ labels) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= KA) return;
    const float *a = anchors + (size_t)k * 4;
    const float bi = best_iou[k];
    int lab = bi < neg_thresh ? 0 : -1;
    int claim = -1;
    for (int g = 0; g < G; ++g) {
        const float gb = __uint_as_float(gt_best_bits[g]);
        if (gb < min_gt_best) continue;                    // per_gt_best[per_gt_best < 0.1] = -1: matches nothing
        if (bbox_iou(a, gts + (size_t)g * gt_stride) == gb) claim = g;
    }
    if (claim >= 0) { lab = 1; best_gt[k] = claim; }
    if (bi > pos_thresh) lab = 1;
    labels[k] = (signed char)lab;
}

// order-preserving compaction of the positives and the negatives (np.where order = ascending index): ONE workgroup of 1024
// threads, each owning a contiguous chunk; counts[0] = #pos, counts[1] = #neg
__global__ __launch_bounds__(1024) void anchor_compact_kernel(const signed char *__restrict__ labels, const int KA,
                                                              int *__restrict__ pos_list, int *__restrict__ neg_list,
                                                              int *__restrict__ counts) {
    __shared__ int spos[1024], sneg[1024];
    const int t = threadIdx.x, per = (KA + 1023) / 1024, lo = t * per, hi = min(KA, lo + per);
    int np_ = 0, nn = 0;
    for (int i = lo; i < hi; ++i) { np_ += labels[i] > 0; nn += labels[i] == 0; }
    spos[t] = np_; sneg[t] = nn;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {     // Hillis-Steele inclusive scan
        const int a = t >= off ? spos[t - off] : 0, b = t >= off ? sneg[t - off] : 0;
        __syncthreads();
        spos[t] += a; sneg[t] += b;
        __syncthreads();
    }
    int op = spos[t] - np_, on = sneg[t] - nn;
    for (int i = lo; i < hi; ++i) {
        if (labels[i] > 0) pos_list[op++] = i;
        else if (labels[i] == 0) neg_list[on++] = i;
    }
    if (t == 1023) { counts[0] = spos[1023]; counts[1] = sneg[1023]; }
}

// labels[list[drop[j]]] = -1  (the surplus the host's np.random.choice picked)
__global__ __launch_bounds__(256) void anchor_drop_kernel(const int *__restrict__ list, const int *__restrict__ drop,
                                                          const int n_drop, signed char *__restrict__ labels) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_drop; j += blockDim.x * gridDim.x) labels[list[drop[j]]] = -1;
}

// final maps in the layout the RPN loss consumes ([1, A, fh, fw] / [1, 4A, fh, fw], anchor k = (y*fw + x)*A + a):
// cls_targets int64, loc_targets = encode(anchor, matched gt) for the remaining positives, loc_masks.
// Encoding as utils/bbox_helper.py:70-86 with the reference's dtypes: the gt's centre / size in fp32 (a float32 array divided
// by a Python float stays float32), the anchor's in float64, the quotient / log in float64, stored as fp32.
__global__ __launch_bounds__(256) void anchor_finalize_kernel(const signed char *__restrict__ labels, const int *__restrict__ best_gt,
                                                              const double *__restrict__ anchors64, const float *__restrict__ gts,
                                                              const int gt_stride, const int A, const int fh, const int fw,
                                                              long long *__restrict__ cls_t, float *__restrict__ loc_t,
                                                              float *__restrict__ loc_m) {
    const int KA = A * fh * fw;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= KA) return;
    const int a = k % A, cell = k / A;
    const int lab = labels[k];
    cls_t[(size_t)a * fh * fw + cell] = lab;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    if (lab > 0) {
        const double *r = anchors64 + (size_t)k * 4;
        const float *g = gts + (size_t)best_gt[k] * gt_stride;
        const float gcx = (g[0] + g[2]) / 2.f, gcy = (g[1] + g[3]) / 2.f, gw = g[2] - g[0], gh = g[3] - g[1];
        const double rcx = (r[0] + r[2]) / 2., rcy = (r[1] + r[3]) / 2., rw = r[2] - r[0], rh = r[3] - r[1];
        t[0] = (float)(((double)gcx - rcx) / rw);
        t[1] = (float)(((double)gcy - rcy) / rh);
        t[2] = (float)log((double)gw / rw);
        t[3] = (float)log((double)gh / rh);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const size_t o = (size_t)(a * 4 + c) * fh * fw + cell;
        loc_t[o] = t[c];
        loc_m[o] = lab > 0 ? 1.f : 0.f;
    }
}

// ---- RPN proposals: functions/rpn_proposal.py:36-60 ------------------------------------------------------------------------
// props[j] = (clip(decode(anchor[order[j]], delta[order[j]])), score[order[j]]) for the host-ranked candidates, fp32 [n, 5];
// ok[j] = both sides >= roi_min_size.  delta / score are read straight out of the RPN's NCHW outputs
// (loc [4A, fh, fw], objectness [2A, fh, fw], anchor k = (y*fw + x)*A + a).  Decode in float64 as utils/bbox_helper.py:88-103
// (numpy promotes the fp32 deltas to the float64 anchors), clip as :105-110, the `+ 1` size test of rpn_proposal.py:57-58.
// exp_wh [n,2] = np.exp(delta_w), np.exp(delta_h) of the ranked candidates AS NUMPY COMPUTES THEM: on a float32 array np.exp
// returns float32 from numpy's own SIMD routine, which is not correctly rounded -- no device expf reproduces it bit for bit
// (a float64 exp here differed from the reference in the last fp32 bit of 7.6 % of the coordinates).  The host therefore
// evaluates these 2n exponentials with numpy (microseconds) and everything else happens here.
__global__ __launch_bounds__(256) void proposal_decode_kernel(const int *__restrict__ order, const float *__restrict__ exp_wh, const int n,
                                                              const double *__restrict__ anchors64,
                                                              const float *__restrict__ loc, const float *__restrict__ prob,
                                                              const int A, const int fh, const int fw, const double img_h,
                                                              const double img_w, const double min_size,
                                                              float *__restrict__ props32, unsigned char *__restrict__ ok) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int k = order[j], a = k % A, cell = k / A;
    const size_t plane = (size_t)fh * fw;
    const double *r = anchors64 + (size_t)k * 4;
    const double d0 = loc[(size_t)(a * 4 + 0) * plane + cell], d1 = loc[(size_t)(a * 4 + 1) * plane + cell];
    const double rcx = (r[0] + r[2]) / 2., rcy = (r[1] + r[3]) / 2., rw = r[2] - r[0], rh = r[3] - r[1];
    const double cx = d0 * rw + rcx, cy = d1 * rh + rcy, w = (double)exp_wh[2 * j] * rw, h = (double)exp_wh[2 * j + 1] * rh;
    double x1 = cx - w / 2., y1 = cy - h / 2., x2 = cx + w / 2., y2 = cy + h / 2.;
    x1 = fmin(fmax(x1, 0.), img_w - 1.); y1 = fmin(fmax(y1, 0.), img_h - 1.);
    x2 = fmin(fmax(x2, 0.), img_w - 1.); y2 = fmin(fmax(y2, 0.), img_h - 1.);
    float *o = props32 + (size_t)j * 5;     // the reference rounds to fp32 exactly once, on its way into the NMS / the result
    o[0] = (float)x1; o[1] = (float)y1; o[2] = (float)x2; o[3] = (float)y2;
    o[4] = prob[(size_t)(a * 2 + 1) * plane + cell];
    ok[j] = (x2 - x1 + 1. >= min_size) && (y2 - y1 + 1. >= min_size);   // tested on the float64 boxes, as the reference does
}

// out[i] = (image index, props[keep[i]]) for i < num_keep (device-resident count); rows beyond stay untouched
__global__ __launch_bounds__(256) void proposal_gather_kernel(const float *__restrict__ props32, const long long *__restrict__ keep,
                                                              const long long *__restrict__ num_keep, const float image_index,
                                                              const int max_rows, float *__restrict__ out6) {
    const int n = (int)min((long long)max_rows, num_keep[0]);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        const float *src = props32 + (size_t)keep[i] * 5;
        float *dst = out6 + (size_t)i * 6;
        dst[0] = image_index;
#pragma unroll
        for (int c = 0; c < 5; ++c) dst[1 + c] = src[c];
    }
}

// ---- RoI sampling for the RCNN head: functions/proposal_target.py:38-62 ----------------------------------------------------
// Candidate r < n_prop is proposal r's box (row stride prop_stride, box at columns 1..4 of a (b, x1, y1, x2, y2, score) row),
// candidate n_prop + g is ground-truth box g (append_gts).  Per candidate: clip to the image as utils/bbox_helper.py:105-110 does
// (np.clip on float32 columns), IoU against every gt on the CLIPPED box, first maximum (ndarray.argmax) -> best_gt / best_iou, and
// the class of the threshold tests of :55-59: 1 = foreground (best_iou > pos), 0 = background (lo <= best_iou < hi and not
// foreground), -1 = neither.  The ordered index lists np.where would return come from anchor_compact_kernel.
__global__ __launch_bounds__(256) void proposal_match_kernel(const float *__restrict__ props, const int n_prop, const int prop_stride,
                                                             const float *__restrict__ gts, const int G, const int gt_stride,
                                                             const float hi_x, const float hi_y, const float pos_thresh,
                                                             const float neg_hi, const float neg_lo, float *__restrict__ rois,
                                                             float *__restrict__ best_iou, int *__restrict__ best_gt,
                                                             signed char *__restrict__ labels) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_prop + G) return;
    const float *src = r < n_prop ? props + (size_t)r * prop_stride + 1 : gts + (size_t)(r - n_prop) * gt_stride;
    float box[4];
    box[0] = fminf(fmaxf(src[0], 0.f), hi_x); box[1] = fminf(fmaxf(src[1], 0.f), hi_y);
    box[2] = fminf(fmaxf(src[2], 0.f), hi_x); box[3] = fminf(fmaxf(src[3], 0.f), hi_y);
#pragma unroll
    for (int c = 0; c < 4; ++c) rois[(size_t)r * 4 + c] = box[c];
    float bi = -1.f;
    int bg = 0;
    for (int g = 0; g < G; ++g) {
        const float v = bbox_iou(box, gts + (size_t)g * gt_stride);
        if (v > bi) { bi = v; bg = g; }
    }
    best_iou[r] = bi;
    best_gt[r] = bg;
    labels[r] = (signed char)(bi > pos_thresh ? 1 : (bi < neg_hi && bi >= neg_lo) ? 0 : -1);
}

// The sampled RoIs of one image, after the host drew them: sel[i] = candidate index, gt_of[i] = its matched gt (>= 0: foreground,
// label = that gt's class; -1: background, label 0), enc [R,4] = the normalised regression target of the foreground rows as numpy
// computed it (:131-135; its float32 log is numpy's own routine).  -> rois [R,5] = (image, clipped box), labels int64 [R],
// loc_targets / loc_weights [R, 4*C] (the foreground row's target and ones in the four columns of its class, zeros elsewhere).
__global__ __launch_bounds__(256) void proposal_finalize_kernel(const float *__restrict__ cand_rois, const int *__restrict__ sel,
                                                                const int *__restrict__ gt_of, const float *__restrict__ enc,
                                                                const float *__restrict__ gts, const int gt_stride, const int R,
                                                                const int C, const float image_index, float *__restrict__ rois5,
                                                                long long *__restrict__ labels, float *__restrict__ loc_t,
                                                                float *__restrict__ loc_w) {
    const int cols = 4 * C;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < R * cols; idx += blockDim.x * gridDim.x) {
        const int i = idx / cols, c = idx - i * cols;
        const int g = gt_of[i];
        const int lab = g >= 0 ? (int)gts[(size_t)g * gt_stride + 4] : 0;
        const bool on = g >= 0 && (c >> 2) == lab;
        loc_t[idx] = on ? enc[(size_t)i * 4 + (c & 3)] : 0.f;
        loc_w[idx] = on ? 1.f : 0.f;
        if (c == 0) labels[i] = lab;
        if (c < 5) rois5[(size_t)i * 5 + c] = c == 0 ? image_index : cand_rois[(size_t)sel[i] * 4 + c - 1];
    }
}

}  // namespace scda

using namespace scda;

#define BOX_CHECK(cond, name) if (!(cond)) { set_error(name ": bad arguments"); return SCDA_EINVAL; }

// Anchor labelling, step 1 of 2 (functions/anchor_target.py:38-64): labels before sub-sampling, ordered index lists of the
// positives / negatives and their counts.  anchors [KA,4] fp32 (the float64 grid cast as the reference casts it for the IoU),
// gts [G, gt_stride >= 4] fp32.  Scratch and outputs are the caller's: best_iou [KA] f32, best_gt [KA] i32, gt_best [G] u32
// (zeroed here), labels [KA] i8, pos_list / neg_list [KA] i32, counts [2] i32.
SCDA_API int scda_anchor_label_hip(const float *anchors, int KA, const float *gts, int G, int gt_stride, float neg_thresh,
                                   float pos_thresh, float min_gt_best, float *best_iou, int *best_gt, unsigned *gt_best,
                                   signed char *labels, int *pos_list, int *neg_list, int *counts, void *stream) {
    BOX_CHECK(anchors && gts && best_iou && best_gt && gt_best && labels && pos_list && neg_list && counts && KA > 0 && G > 0 &&
              gt_stride >= 4, "scda_anchor_label_hip")
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(gt_best, 0, (size_t)G * sizeof(unsigned), st) != hipSuccess) return launch_status("hipMemsetAsync");
    const int blocks = cdiv(KA, 256);
    hipLaunchKernelGGL(anchor_match_kernel, dim3(blocks), dim3(256), (size_t)G * sizeof(unsigned), st, anchors, KA, gts, G, gt_stride,
                       best_iou, best_gt, gt_best);
    hipLaunchKernelGGL(anchor_label_kernel, dim3(blocks), dim3(256), 0, st, anchors, KA, gts, G, gt_stride, best_iou, best_gt, gt_best,
                       neg_thresh, pos_thresh, min_gt_best, labels);
    hipLaunchKernelGGL(anchor_compact_kernel, dim3(1), dim3(1024), 0, st, labels, KA, pos_list, neg_list, counts);
    return launch_status("anchor_label kernels");
}

// step 2 of 2 (:66-107): the host drew which surplus positives / negatives to drop (indices INTO pos_list / neg_list, as
// np.random.choice returns them); apply them and emit cls_targets int64 [A,fh,fw], loc_targets / loc_masks fp32 [4A,fh,fw]
SCDA_API int scda_anchor_finalize_hip(signed char *labels, const int *best_gt, const int *pos_list, const int *drop_pos, int n_drop_pos,
                                      const int *neg_list, const int *drop_neg, int n_drop_neg, const double *anchors64,
                                      const float *gts, int gt_stride, int A, int fh, int fw, long long *cls_targets,
                                      float *loc_targets, float *loc_masks, void *stream) {
    BOX_CHECK(labels && best_gt && pos_list && neg_list && anchors64 && gts && cls_targets && loc_targets && loc_masks && A > 0 &&
              fh > 0 && fw > 0 && n_drop_pos >= 0 && n_drop_neg >= 0 && (n_drop_pos == 0 || drop_pos) && (n_drop_neg == 0 || drop_neg),
              "scda_anchor_finalize_hip")
    hipStream_t st = as_stream(stream);
    if (n_drop_pos) hipLaunchKernelGGL(anchor_drop_kernel, dim3(ew_grid(n_drop_pos)), dim3(256), 0, st, pos_list, drop_pos, n_drop_pos, labels);
    if (n_drop_neg) hipLaunchKernelGGL(anchor_drop_kernel, dim3(ew_grid(n_drop_neg)), dim3(256), 0, st, neg_list, drop_neg, n_drop_neg, labels);
    hipLaunchKernelGGL(anchor_finalize_kernel, dim3(cdiv(A * fh * fw, 256)), dim3(256), 0, st, labels, best_gt, anchors64, gts, gt_stride,
                       A, fh, fw, cls_targets, loc_targets, loc_masks);
    return launch_status("anchor_finalize kernels");
}

// RPN proposals of one image (functions/rpn_proposal.py:36-60) for the candidates the host ranked: decode + clip + size test,
// then the caller runs scda_nms_valid_hip on props5 / ok and scda_proposal_gather_hip on the keep list.
SCDA_API int scda_proposal_decode_hip(const int *order, const float *exp_wh, int n, const double *anchors64, const float *loc,
                                      const float *prob, int A, int fh, int fw, double img_h, double img_w, double min_size,
                                      float *props5, unsigned char *ok, void *stream) {
    BOX_CHECK(n >= 0 && (n == 0 || (order && exp_wh && anchors64 && loc && prob && props5 && ok)) && A > 0 && fh > 0 && fw > 0,
              "scda_proposal_decode_hip")
    if (n == 0) return SCDA_OK;
    hipLaunchKernelGGL(proposal_decode_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), order, exp_wh, n, anchors64, loc, prob, A,
                       fh, fw, img_h, img_w, min_size, props5, ok);
    return launch_status("proposal_decode_kernel");
}

SCDA_API int scda_proposal_gather_hip(const float *props5, const long long *keep, const long long *num_keep, float image_index,
                                      int max_rows, float *out6, void *stream) {
    BOX_CHECK(props5 && keep && num_keep && out6 && max_rows > 0, "scda_proposal_gather_hip")
    hipLaunchKernelGGL(proposal_gather_kernel, dim3(ew_grid(max_rows)), dim3(256), 0, as_stream(stream), props5, keep, num_keep,
                       image_index, max_rows, out6);
    return launch_status("proposal_gather_kernel");
}

// RoI sampling, step 1 of 2 (functions/proposal_target.py:38-62): candidates = proposals (+ ground-truth boxes), clipped; their
// best gt / IoU; foreground / background classes; ordered index lists and counts.  props [n_prop, prop_stride >= 5] fp32 rows
// (b, x1, y1, x2, y2, ...), gts [G, gt_stride >= 5] fp32 (x1, y1, x2, y2, class).  Outputs (caller-owned): rois [n_prop+G, 4],
// best_iou / best_gt / labels [n_prop+G], pos_list / neg_list [n_prop+G] i32, counts [2] i32.
SCDA_API int scda_proposal_match_hip(const float *props, int n_prop, int prop_stride, const float *gts, int G, int gt_stride,
                                     float img_h, float img_w, float pos_thresh, float neg_hi, float neg_lo, float *rois,
                                     float *best_iou, int *best_gt, signed char *labels, int *pos_list, int *neg_list, int *counts,
                                     void *stream) {
    BOX_CHECK(gts && rois && best_iou && best_gt && labels && pos_list && neg_list && counts && n_prop >= 0 && (n_prop == 0 || props) &&
              G > 0 && prop_stride >= 5 && gt_stride >= 5, "scda_proposal_match_hip")
    hipStream_t st = as_stream(stream);
    const int n = n_prop + G;
    hipLaunchKernelGGL(proposal_match_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, props, n_prop, prop_stride, gts, G, gt_stride,
                       img_w - 1.f, img_h - 1.f, pos_thresh, neg_hi, neg_lo, rois, best_iou, best_gt, labels);
    hipLaunchKernelGGL(anchor_compact_kernel, dim3(1), dim3(1024), 0, st, labels, n, pos_list, neg_list, counts);
    return launch_status("proposal_match kernels");
}

// step 2 of 2 (:64-136): gather what the host sampled (see proposal_finalize_kernel)
SCDA_API int scda_proposal_finalize_hip(const float *cand_rois, const int *sel, const int *gt_of, const float *enc, const float *gts,
                                        int gt_stride, int R, int num_classes, float image_index, float *rois5, long long *labels,
                                        float *loc_targets, float *loc_weights, void *stream) {
    BOX_CHECK(cand_rois && sel && gt_of && enc && gts && rois5 && labels && loc_targets && loc_weights && R > 0 && num_classes > 1 &&
              gt_stride >= 5, "scda_proposal_finalize_hip")
    hipLaunchKernelGGL(proposal_finalize_kernel, dim3(ew_grid((long long)R * 4 * num_classes)), dim3(256), 0, as_stream(stream),
                       cand_rois, sel, gt_of, enc, gts, gt_stride, R, num_classes, image_index, rois5, labels, loc_targets, loc_weights);
    return launch_status("proposal_finalize_kernel");
}
