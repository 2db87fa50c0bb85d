/*
 * scda_oracle.c -- CPU restatement of the SCDA reference's native operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (scda_amd/) may
 * import, link or call this file.  It is used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it restates.  How each restatement is pinned:
 *   bbox_overlaps (IoU without +1): BIT-EXACT against the reference's own Cython
 *     (extensions/_cython_bbox/cython_bbox.pyx built unmodified by
 *     oracle/build_ref.py; vectors in tests/golden/bbox_overlaps.npz).
 *   NMS (mask + sweep, and the ">=" CPU variant): BIT-EXACT against keep lists
 *     produced by the reference's own extensions/_cython_bbox/cython_nms.pyx,
 *     compiled unmodified with the image's python3.9 / numpy 1.26 / Cython 0.29
 *     (oracle/build_ref.py; tests/golden/make_golden_nms.py -> nms_ref.npz:
 *     300 ... 12000 boxes, RPN-like / clustered / integer, thresholds .7/.5/.3,
 *     inputs tie-free at the threshold where ">=" and ">" part ways).
 *   RoIPool forward: BIT-EXACT against outputs of the reference's own pure-Python
 *     RoIPool, extensions/_roi_pooling/modules/roi_pool_py.py:7-47, imported
 *     UNMODIFIED (tests/golden/ref_harness.py adds the torch<=0.3 semantics it was
 *     written for: identity .cuda(), `max(x, dim)` keeping the reduced dimension,
 *     0.3-style row indexing).  tests/golden/make_golden_roipool.py ->
 *     roi_pool_ref.npz: [1,512,32,64] x 512 RoIs (the hot path's call) and two
 *     ragged shapes (two images, 3x5 pooling, RoIs outside the map, tied values).
 *     The inputs (tests/roipool_cases.py) stay off the two points where that file
 *     and roi_pooling_kernel.cu -- the canonical text -- are different operators:
 *     scaled corners on an exact .5 (np.round is half-to-even, CUDA round() half
 *     away from zero) and RoI extents of 29 / 57 / 58 cells (double vs float bin
 *     edges); both have tests of their own against the kernel's text.  The argmax
 *     and the backward are not part of roi_pool_py.py: argmax is checked through
 *     the contract the values imply (features[argmax] == value, first maximum of
 *     the bin in scan order, -1 <=> empty), the backward against the gather text
 *     of roi_pooling_kernel.cu:128-203 restated twice (C gather, numpy scatter).
 *   RoIAlign, focal loss, the "+1/clamped" IoU: PARITY UNPINNED against a build or a
 *     run of the reference.  Their only sources are .cu files (nvcc + the TH/THC
 *     headers torch 2.x no longer ships) and the reference holds no Python statement,
 *     test or vector of them.  What they ARE checked against: independent second
 *     statements written from the operator definitions (tests/np_restate.py,
 *     float64 autograd closed forms: tests/test_oracle_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off is part of the definition: the canonical arithmetic of the
 * reference kernels is "one IEEE fp32 operation per source operator".
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* NMS  (extensions/_nms/src/cuda/nms_kernel.cu:16-24 devIoU,          */
/*       :26-70 nms_kernel, extensions/_nms/src/nms_cuda.c:47-58 sweep) */
/* ------------------------------------------------------------------ */
static float orc_iou_plus1(const float *a, const float *b) {
    /* nms_kernel.cu:17-23, "+1" pixel convention, fp32 throughout */
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float width = fmaxf(right - left + 1, 0.f);
    float height = fmaxf(bottom - top + 1, 0.f);
    float interS = width * height;
    float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
    float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
    return interS / (Sa + Sb - interS);
}

/* boxes: [n,5] row-major (x1,y1,x2,y2,score), already sorted by score desc.
 * mask:  [n, ceil(n/64)] uint64, bit j of word (i, cb) set iff
 *        IoU(box i, box cb*64+j) > thresh and (cb*64+j) > i   (nms_kernel.cu:58-65) */
ORC_API void orc_nms_mask(const float *boxes, int n, float thresh, uint64_t *mask) {
    const int col_blocks = (n + 63) / 64;
    for (int i = 0; i < n; ++i) {
        const int rb = i / 64, ri = i % 64;
        for (int cb = 0; cb < col_blocks; ++cb) {
            const int col_size = (n - cb * 64) < 64 ? (n - cb * 64) : 64;
            uint64_t t = 0;
            int start = (rb == cb) ? ri + 1 : 0; /* nms_kernel.cu:58-60 */
            for (int j = start; j < col_size; ++j)
                if (orc_iou_plus1(boxes + (size_t)i * 5, boxes + (size_t)(cb * 64 + j) * 5) > thresh)
                    t |= 1ULL << j;
            mask[(size_t)i * col_blocks + cb] = t;
        }
    }
}

/* Greedy sweep of nms_cuda.c:47-58.  keep has room for n entries. Returns 1. */
ORC_API int orc_nms(const float *boxes, int n, float thresh, int64_t *keep, int64_t *num_out) {
    const int col_blocks = (n + 63) / 64;
    if (n == 0) { *num_out = 0; return 1; }
    uint64_t *mask = (uint64_t *)malloc((size_t)n * col_blocks * sizeof(uint64_t));
    uint64_t *remv = (uint64_t *)calloc(col_blocks, sizeof(uint64_t));
    orc_nms_mask(boxes, n, thresh, mask);
    int64_t nk = 0;
    for (int i = 0; i < n; ++i) {
        int nblock = i / 64, inblock = i % 64;
        if (!(remv[nblock] & (1ULL << inblock))) {
            keep[nk++] = i;
            const uint64_t *p = mask + (size_t)i * col_blocks;
            for (int j = nblock; j < col_blocks; ++j) remv[j] |= p[j];
        }
    }
    *num_out = nk;
    free(mask); free(remv);
    return 1;
}

/* CPU NMS of extensions/_nms/src/nms.c:4-68 (">=" threshold, explicit order/areas).
 * API symbol only; nobody calls it on the hot path (pth_nms.py:9-24 commented out). */
ORC_API int orc_cpu_nms(const float *boxes, int n, int dim, const int64_t *order, const float *areas,
                        float thresh, int64_t *keep, int64_t *num_out) {
    unsigned char *sup = (unsigned char *)calloc(n > 0 ? n : 1, 1);
    int64_t nk = 0;
    for (int _i = 0; _i < n; ++_i) {
        int i = (int)order[_i];
        if (sup[i]) continue;
        keep[nk++] = i;
        float ix1 = boxes[i * dim], iy1 = boxes[i * dim + 1], ix2 = boxes[i * dim + 2], iy2 = boxes[i * dim + 3];
        float iarea = areas[i];
        for (int _j = _i + 1; _j < n; ++_j) {
            int j = (int)order[_j];
            if (sup[j]) continue;
            float xx1 = fmaxf(ix1, boxes[j * dim]), yy1 = fmaxf(iy1, boxes[j * dim + 1]);
            float xx2 = fminf(ix2, boxes[j * dim + 2]), yy2 = fminf(iy2, boxes[j * dim + 3]);
            float w = fmaxf(0.0f, xx2 - xx1 + 1), h = fmaxf(0.0f, yy2 - yy1 + 1);
            float inter = w * h;
            float ovr = inter / (iarea + areas[j] - inter);
            if (ovr >= thresh) sup[j] = 1;
        }
    }
    *num_out = nk;
    free(sup);
    return 1;
}

/* ------------------------------------------------------------------ */
/* RoI max pooling (extensions/_roi_pooling/src/roi_pooling_kernel.cu) */
/* ------------------------------------------------------------------ */
/* Integer RoI rectangle on the feature map; roi_pooling_kernel.cu:45-49.
 * round() on a float is half-away-from-zero (roundf).                    */
static void orc_roi_rect(const float *roi, float scale, int *b, int *sw, int *sh, int *ew, int *eh) {
    *b = (int)roi[0];
    *sw = (int)roundf(roi[1] * scale);
    *sh = (int)roundf(roi[2] * scale);
    *ew = (int)roundf(roi[3] * scale);
    *eh = (int)roundf(roi[4] * scale);
}

/* features [B,C,H,W], rois [R,5] -> out [R,C,PH,PW], argmax int32 (flat index
 * into features, -1 when the bin is empty).  roi_pooling_kernel.cu:24-93.   */
ORC_API int orc_roi_pool_fwd(const float *feat, const float *rois, int R, int C, int H, int W, int PH, int PW,
                             float scale, float *out, int32_t *argmax) {
    for (int n = 0; n < R; ++n) {
        int b, sw, sh, ew, eh;
        orc_roi_rect(rois + (size_t)n * 5, scale, &b, &sw, &sh, &ew, &eh);
        int roi_w = (int)fmaxf((float)(ew - sw + 1), 1.f); /* :52-53 */
        int roi_h = (int)fmaxf((float)(eh - sh + 1), 1.f);
        float bin_h = (float)roi_h / (float)PH, bin_w = (float)roi_w / (float)PW;
        for (int c = 0; c < C; ++c) {
            const int base = (b * C + c) * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    int hstart = (int)floorf((float)ph * bin_h), wstart = (int)floorf((float)pw * bin_w);
                    int hend = (int)ceilf((float)(ph + 1) * bin_h), wend = (int)ceilf((float)(pw + 1) * bin_w);
                    hstart = (int)fminf(fmaxf((float)(hstart + sh), 0.f), (float)H); /* :63-66 */
                    hend = (int)fminf(fmaxf((float)(hend + sh), 0.f), (float)H);
                    wstart = (int)fminf(fmaxf((float)(wstart + sw), 0.f), (float)W);
                    wend = (int)fminf(fmaxf((float)(wend + sw), 0.f), (float)W);
                    int empty = (hend <= hstart) || (wend <= wstart);
                    float maxval = empty ? 0.f : -FLT_MAX;
                    int maxidx = -1;
                    for (int h = hstart; h < hend; ++h)
                        for (int w = wstart; w < wend; ++w) {
                            int idx = base + h * W + w;
                            if (feat[idx] > maxval) { maxval = feat[idx]; maxidx = idx; }
                        }
                    size_t o = (((size_t)n * C + c) * PH + ph) * PW + pw;
                    out[o] = maxval;
                    if (argmax) argmax[o] = maxidx;
                }
        }
    }
    return 1;
}

/* Gather-form backward, summation order roi^, ph^, pw^ (roi_pooling_kernel.cu:128-203). */
ORC_API int orc_roi_pool_bwd(const float *top_diff, const int32_t *argmax, const float *rois, int R, int B, int C,
                             int H, int W, int PH, int PW, float scale, float *bottom_diff) {
    int *rect = (int *)malloc((size_t)(R > 0 ? R : 1) * 5 * sizeof(int));
    for (int r = 0; r < R; ++r)
        orc_roi_rect(rois + (size_t)r * 5, scale, rect + r * 5, rect + r * 5 + 1, rect + r * 5 + 2, rect + r * 5 + 3,
                     rect + r * 5 + 4);
    for (int n = 0; n < B; ++n)
        for (int c = 0; c < C; ++c)
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    const int index = ((n * C + c) * H + h) * W + w;
                    float g = 0.f;
                    for (int r = 0; r < R; ++r) {
                        const int *q = rect + r * 5;
                        if (q[0] != n) continue;
                        int sw = q[1], sh = q[2], ew = q[3], eh = q[4];
                        if (!(w >= sw && w <= ew && h >= sh && h <= eh)) continue;
                        int roi_w = (int)fmaxf((float)(ew - sw + 1), 1.f);
                        int roi_h = (int)fmaxf((float)(eh - sh + 1), 1.f);
                        float bin_h = (float)roi_h / (float)PH, bin_w = (float)roi_w / (float)PW;
                        int phs = (int)floorf((float)(h - sh) / bin_h), phe = (int)ceilf((float)(h - sh + 1) / bin_h);
                        int pws = (int)floorf((float)(w - sw) / bin_w), pwe = (int)ceilf((float)(w - sw + 1) / bin_w);
                        phs = (int)fminf(fmaxf((float)phs, 0.f), (float)PH);
                        phe = (int)fminf(fmaxf((float)phe, 0.f), (float)PH);
                        pws = (int)fminf(fmaxf((float)pws, 0.f), (float)PW);
                        pwe = (int)fminf(fmaxf((float)pwe, 0.f), (float)PW);
                        const size_t off = (size_t)r * C * PH * PW;
                        for (int ph = phs; ph < phe; ++ph)
                            for (int pw = pws; pw < pwe; ++pw) {
                                size_t o = off + ((size_t)c * PH + ph) * PW + pw;
                                if (argmax[o] == index) g += top_diff[o];
                            }
                    }
                    bottom_diff[index] = g;
                }
    free(rect);
    return 1;
}

/* Scatter form of the same backward: walking the outputs in (roi, c, ph, pw) order and adding top_diff into
 * bottom_diff[argmax] visits, for every input element, its contributors in (roi, ph, pw) order -- the gather
 * kernel's order -- so the fp32 sums are bit-identical (asserted in tests/test_oracle_golden.py) at 1/1000 of the
 * cost.  Used by the model-level oracle / CPU baseline.                                                        */
ORC_API int orc_roi_pool_bwd_scatter(const float *top_diff, const int32_t *argmax, int R, int B, int C, int H, int W,
                                     int PH, int PW, float *bottom_diff) {
    memset(bottom_diff, 0, (size_t)B * C * H * W * sizeof(float));
    const size_t total = (size_t)R * C * PH * PW;
    for (size_t i = 0; i < total; ++i)
        if (argmax[i] >= 0) bottom_diff[argmax[i]] += top_diff[i];
    return 1;
}

/* ------------------------------------------------------------------ */
/* RoIAlign, old single-sample variant (extensions/_roi_align/src/roi_align_kernel.cu) */
/* The CUDA text mixes float and double literals (1., 0.); the double ones  */
/* are kept as double here so that rounding happens where nvcc rounds.      */
/* ------------------------------------------------------------------ */
typedef struct { int ok; int upleft; float hr, wr; } orc_ra_pt;

static orc_ra_pt orc_ra_point(const float *roi, float scale, int C, int H, int W, int AH, int AW, int c, int ph, int pw) {
    /* roi_align_kernel.cu:32-53 */
    orc_ra_pt r; r.ok = 0; r.upleft = 0; r.hr = r.wr = 0.f;
    float roi_batch_ind = roi[0];
    float sw = roi[1] * scale, sh = roi[2] * scale, ew = roi[3] * scale, eh = roi[4] * scale;
    float roi_w = fmaxf((float)((double)(ew - sw) + 1.), 0.f);
    float roi_h = fmaxf((float)((double)(eh - sh) + 1.), 0.f);
    float bin_h = (float)((double)roi_h / ((double)AH - 1.));
    float bin_w = (float)((double)roi_w / ((double)AW - 1.));
    float h = (float)ph * bin_h + sh;
    float w = (float)pw * bin_w + sw;
    int hstart = (int)fminf(floorf(h), (float)(H - 2));
    int wstart = (int)fminf(floorf(w), (float)(W - 2));
    int img_start = (int)(roi_batch_ind * (float)(C * H * W));
    if (h < 0 || h >= H || w < 0 || w >= W) return r;
    r.ok = 1;
    r.hr = h - (float)hstart;
    r.wr = w - (float)wstart;
    r.upleft = img_start + (c * H + hstart) * W + wstart;
    return r;
}

ORC_API int orc_roi_align_fwd(const float *feat, const float *rois, int R, int C, int H, int W, int AH, int AW,
                              float scale, float *out) {
    for (int n = 0; n < R; ++n)
        for (int c = 0; c < C; ++c)
            for (int ph = 0; ph < AH; ++ph)
                for (int pw = 0; pw < AW; ++pw) {
                    size_t o = (((size_t)n * C + c) * AH + ph) * AW + pw;
                    orc_ra_pt p = orc_ra_point(rois + (size_t)n * 5, scale, C, H, W, AH, AW, c, ph, pw);
                    if (!p.ok) { out[o] = 0.f; continue; }
                    /* :62-65, evaluated in double because of the "1." literals */
                    double hr = p.hr, wr = p.wr;
                    double v = (double)feat[p.upleft] * (1. - hr) * (1. - wr) + (double)feat[p.upleft + 1] * (1. - hr) * wr +
                               (double)feat[p.upleft + W] * hr * (1. - wr) + (double)feat[p.upleft + W + 1] * hr * wr;
                    out[o] = (float)v;
                }
    return 1;
}

/* Scatter-add backward (atomicAdd in the reference, :94-143): order of the fp32
 * sums is unspecified there; this restatement adds in output-index order.
 * Parity tolerance 1e-5 relative.                          */
ORC_API int orc_roi_align_bwd(const float *top_diff, const float *rois, int R, int C, int H, int W, int AH, int AW,
                              float scale, float *bottom_diff /* pre-zeroed by caller */) {
    for (int n = 0; n < R; ++n)
        for (int c = 0; c < C; ++c)
            for (int ph = 0; ph < AH; ++ph)
                for (int pw = 0; pw < AW; ++pw) {
                    size_t o = (((size_t)n * C + c) * AH + ph) * AW + pw;
                    orc_ra_pt p = orc_ra_point(rois + (size_t)n * 5, scale, C, H, W, AH, AW, c, ph, pw);
                    if (!p.ok) continue;
                    double hr = p.hr, wr = p.wr, d = top_diff[o];
                    bottom_diff[p.upleft] += (float)(d * (1. - hr) * (1 - wr));
                    bottom_diff[p.upleft + 1] += (float)(d * (1. - hr) * wr);
                    bottom_diff[p.upleft + W] += (float)(d * hr * (1 - wr));
                    bottom_diff[p.upleft + W + 1] += (float)(d * hr * wr);
                }
    return 1;
}

/* ------------------------------------------------------------------ */
/* Focal loss (extensions/_focal_loss/src/cuda/focal_loss_*_kernel.cu) */
/* ------------------------------------------------------------------ */
/* sigmoid forward: focal_loss_sigmoid_kernel.cu:12-47.  N = rows*num_classes. */
ORC_API int orc_focal_sigmoid_fwd(int N, const float *logits, const int32_t *targets, float weight_pos, float gamma,
                                  float alpha, int num_classes, float *losses) {
    for (int i = 0; i < N; ++i) {
        int d = i % num_classes, t = targets[i / num_classes];
        float c1 = (t == (d + 1));
        float c2 = ((t != -1) & (t != (d + 1)));
        float Np = (float)fmax((double)weight_pos, 1.0);
        float zn = (float)((1.0 - (double)alpha) / (double)Np);
        float zp = alpha / Np;
        float x = logits[i];
        float p = (float)(1. / (1. + (double)expf(-x)));
        float term1 = (float)((double)powf((float)(1. - (double)p), gamma) * (double)logf(fmaxf(p, FLT_MIN)));
        float ge = (x >= 0);
        float term2 = (float)((double)powf(p, gamma) *
                              (-1. * (double)x * (double)ge -
                               (double)logf((float)(1. + (double)expf((float)((double)x - 2. * (double)x * (double)ge))))));
        float l = 0.0f;
        l += -c1 * term1 * zp;
        l += -c2 * term2 * zn;
        losses[i] = l;
    }
    return 1;
}

/* sigmoid gradient: focal_loss_sigmoid_kernel.cu:49-81 */
ORC_API int orc_focal_sigmoid_bwd(int N, const float *logits, const int32_t *targets, float *dX, float weight_pos,
                                  float gamma, float alpha, int num_classes) {
    for (int i = 0; i < N; ++i) {
        int d = i % num_classes, t = targets[i / num_classes];
        float Np = (float)fmax((double)weight_pos, 1.0);
        float zn = (float)((1.0 - (double)alpha) / (double)Np);
        float zp = alpha / Np;
        float c1 = (t == (d + 1));
        float c2 = ((t != -1) & (t != (d + 1)));
        float x = logits[i];
        float p = (float)(1. / (1. + (double)expf(-x)));
        float term1 = (float)((double)powf((float)(1. - (double)p), gamma) *
                              (1. - (double)p - (double)(p * gamma * logf(fmaxf(p, FLT_MIN)))));
        float ge = (x >= 0);
        double lg = -1. * (double)x * (double)ge -
                    (double)logf((float)(1. + (double)expf((float)((double)x - 2. * (double)x * (double)ge))));
        float term2 = (float)((double)powf(p, gamma) * (lg * (1. - (double)p) * (double)gamma - (double)p));
        float g = 0.0f;
        g += -c1 * zp * term1;
        g += -c2 * zn * term2;
        dX[i] = g;
    }
    return 1;
}

/* softmax forward: SpatialSoftmaxKernel :12-34 + SoftmaxFocalLossKernel :36-57 */
ORC_API int orc_focal_softmax_fwd(int N, const float *logits, const int32_t *targets, float weight_pos, float gamma,
                                  float alpha, int num_classes, float *losses, float *priors) {
    int rows = N / num_classes;
    for (int r = 0; r < rows; ++r) {
        int base = r * num_classes;
        float mx = -FLT_MAX;
        for (int c = 0; c < num_classes; ++c) mx = fmaxf(mx, logits[base + c]);
        float es = 0.0f;
        for (int c = 0; c < num_classes; ++c) { float e = expf(logits[base + c] - mx); priors[base + c] = e; es += e; }
        for (int c = 0; c < num_classes; ++c) priors[base + c] /= es;
        int label = targets[r];
        float Np = (float)fmax((double)weight_pos, 1.0);
        float z = (label == 0) * (1 - alpha) / Np + (label >= 1) * alpha / Np;
        float l = 0.0f;
        if (label >= 0) {
            float pl = priors[base + label];
            /* log(float) in CUDA C++ is the float overload; only "1.0 - p" is double */
            l = -(powf((float)(1.0 - (double)pl), gamma) * logf(fmaxf(pl, FLT_MIN))) * z;
        }
        losses[r] = l;
    }
    return 1;
}

/* softmax backward: GradientWeightKernel :59-81 + GradientKernel :84-100 */
ORC_API int orc_focal_softmax_bwd(int N, const float *logits, const int32_t *targets, float *dX, float weight_pos,
                                  float gamma, float alpha, int num_classes, const float *priors, float *buff) {
    (void)logits;
    int rows = N / num_classes;
    for (int r = 0; r < rows; ++r) {
        int base = r * num_classes, label = targets[r];
        float Np = (float)fmax((double)weight_pos, 1.0);
        float z = (label == 0) * (1 - alpha) / Np + (label >= 1) * alpha / Np;
        float b = 0.0f;
        if (label >= 0) {
            float onemp = (float)(1. - (double)priors[base + label]);
            float p = priors[base + label];
            b = (-powf(onemp, gamma) + gamma * powf(onemp, gamma - 1) * p * logf(fmaxf(p, FLT_MIN))) * z;
        }
        buff[r] = b;
    }
    for (int i = 0; i < N; ++i) {
        int ind = i / num_classes, cls = i % num_classes, label = targets[ind];
        float c1 = (float)((label >= 0) * 1.0), c2 = (float)((label == cls) * 1.0);
        dX[i] = c1 * buff[ind] * (c2 - priors[i]);
    }
    return 1;
}

/* ------------------------------------------------------------------ */
/* Box overlap matrices                                                */
/* ------------------------------------------------------------------ */
/* extensions/_bbox_helper/src/cuda/iou_overlap_kernel.cu:33-65: no +1, union clamped >= 1 */
ORC_API int orc_iou_overlaps(const float *b1, const float *b2, int size_bbox, int n1, int n2, float *out) {
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n2; ++j) {
            const float *p = b1 + (size_t)i * size_bbox, *q = b2 + (size_t)j * size_bbox;
            float a1 = (p[2] - p[0]) * (p[3] - p[1]), a2 = (q[2] - q[0]) * (q[3] - q[1]);
            float left = fmaxf(p[0], q[0]), right = fminf(p[2], q[2]);
            float top = fmaxf(p[1], q[1]), bottom = fminf(p[3], q[3]);
            float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
            float interS = width * height;
            float unionS = fmaxf(a1 + a2 - interS, 1.0f);
            out[(size_t)i * n2 + j] = interS / unionS;
        }
    return 1;
}

/* extensions/_cython_bbox/cython_bbox.pyx:32-73: no +1, zero unless iw>0 && ih>0,
 * no union clamp.  Every variable is a C float (DTYPE_t); `ua = float(expr)` widens
 * to double and narrows back on assignment, so the arithmetic is plain fp32.
 * Pinned bit-exactly against the reference .pyx built unmodified
 * (tests/golden/make_golden.py -> tests/golden/bbox_overlaps.npz).                  */
ORC_API int orc_bbox_overlaps(const float *boxes, int N, const float *query, int K, float *out) {
    memset(out, 0, (size_t)N * K * sizeof(float));
    for (int k = 0; k < K; ++k) {
        const float *q = query + (size_t)k * 4;
        float box_area = (q[2] - q[0]) * (q[3] - q[1]);
        for (int n = 0; n < N; ++n) {
            const float *b = boxes + (size_t)n * 4;
            float iw = fminf(b[2], q[2]) - fmaxf(b[0], q[0]);
            if (iw > 0) {
                float ih = fminf(b[3], q[3]) - fmaxf(b[1], q[1]);
                if (ih > 0) {
                    float ua = (b[2] - b[0]) * (b[3] - b[1]) + box_area - iw * ih;
                    out[(size_t)n * K + k] = iw * ih / ua;
                }
            }
        }
    }
    return 1;
}
