/*
 * scda_ops.h -- C ABI of libscda_ops.so, the MI355X (gfx950) implementation of
 * the SCDA Faster-R-CNN hot path.
 *
 * This is the drop-in boundary: every entry point replaces one function that
 * the reference exported through torch.utils.ffi (cffi) or Cython; the
 * reference declaration it replaces is cited as  <file>:<line>  relative to the
 * reference tree.  INTEGRATION.md shows the ctypes stub that binds each one.
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is a DEVICE pointer
 *     (hipMalloc'd / torch CUDA tensor .data_ptr()) unless the name ends in _host
 *   - the caller owns and allocates every output and workspace buffer
 *     (reference convention: Python allocates keep/num_out/output/argmax/...,
 *     extensions/_roi_pooling/functions/roi_pool.py:19-22)
 *   - `stream` is a hipStream_t passed as void* (NULL = legacy default stream);
 *     all work is enqueued asynchronously on it, nothing synchronises
 *   - return value: 0 = ok, <0 = error (SCDA_E*), never exit(): the reference's
 *     "1 = ok / 0 = bad shape / exit(-1) on launch failure"
 *     (roi_pooling_cuda.c:20-23, roi_pooling_kernel.cu:117-122) becomes a status
 *   - all tensors are contiguous, row-major, fp32 unless stated
 */
#ifndef SCDA_OPS_H
#define SCDA_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCDA_OK 0
#define SCDA_EINVAL (-1)  /* bad shape / null pointer / unsupported size */
#define SCDA_ELAUNCH (-2) /* hipGetLastError() != hipSuccess after a launch */
#define SCDA_ENODEV (-3)  /* no HIP device */

/* library / device probes (no compute) */
int scda_version(void);
int scda_device_count(void);
const char *scda_last_error(void);

/* launch profiler for the GEMM-class kernels: for every kernel class k whose bit is set in kernel_mask a hipEvent
 * pair is recorded on the launch stream around each launch of that kernel; after a device synchronisation
 * scda_prof_collect() fills, per class k in [0, scda_prof_num_kernels()), the number of launches, their summed
 * duration (ms), algorithmic FLOPs and (bytes may be NULL) algorithmic HBM bytes = operands read once + result written
 * once.  (Event records are queue markers: keep the mask narrow inside timed regions.) */
void scda_prof_enable(unsigned kernel_mask);
int scda_prof_num_kernels(void);
const char *scda_prof_kernel_name(int k);
int scda_prof_collect(long long *launches, double *ms, double *flops, double *bytes);
/* test aid: {tile rows, tile cols, split-K count, 1 = direct-to-LDS kernel family} of the calling thread's most recent conv /
 * GEMM launch (the planner's choice, or the SCDA_PLAN_FORCE="bm,bn,splits" override when that is legal for the shape) */
void scda_debug_last_plan(int *out4);
/* test aid: launch order of the calling thread's most recent Winograd launches -- forward / data gradient {tile rows / 32,
 * 1 = contiguous pixel-block runs per XCD, gm (XCDs split gm x 8/gm over m-tile groups x runs; 1 = none), split-K count}, weight
 * gradient {K-splits, 0 = dealt as they come / 1 = whole splits per XCD / 2 = one split + one m-tile group per XCD} */
void scda_debug_wino_last_order(int *out6);
/* ... and whether the calling thread's most recent forward / data-gradient launch was the PERSISTENT form (one workgroup per CU walking
 * the tiles: launches of more 64-row tiles than CUs; SCDA_WINO_PERSIST=0 turns it off): 1 / 0 */
int scda_debug_wino_last_persistent(void);

/* ---------------------------------------------------------------- NMS ---- */
/* replaces  int gpu_nms(THLongTensor* keep, THLongTensor* num_out, THCudaTensor* boxes, float thresh)
 *           extensions/_nms/src/nms_cuda.h:1, nms_cuda.c:17-67, cuda/nms_kernel.cu:26-83
 * boxes   [n,5] (x1,y1,x2,y2,score) sorted by score descending
 * mask_ws  workspace, scda_nms_workspace_bytes(n) bytes
 * keep    int64 [n]   indices of kept boxes, ascending            (device)
 * num_out int64 [1]   number of valid entries in keep             (device)
 * max_keep  <=0: full sweep (reference semantics); >0: stop after max_keep
 *           kept boxes (identical to truncating the full result, which is what
 *           both call sites do: functions/rpn_proposal.py:65-66)
 * Unlike the reference nothing is copied to the host: the greedy sweep
 * (nms_cuda.c:47-58) also runs on the device.                                */
size_t scda_nms_workspace_bytes(int n);
int scda_nms_hip(const float *boxes, int n, float thresh, void *mask_ws, int64_t *keep, int64_t *num_out,
                 int max_keep, void *stream);
/* the same with per-box validity flags (uint8 [n], may be NULL): boxes flagged 0 are treated as absent -- never kept, never
 * suppressing -- and `keep` indexes the ORIGINAL list: the min-size filter of functions/rpn_proposal.py:57-59 without compaction */
int scda_nms_valid_hip(const float *boxes, const unsigned char *valid, int n, float thresh, void *mask_ws, int64_t *keep,
                       int64_t *num_out, int max_keep, void *stream);
/* the pairwise suppression bit-mask alone (upper triangle of col-blocks only):
 * mask uint64 [n, ceil(n/64)]                                                */
int scda_nms_mask_hip(const float *boxes, int n, float thresh, uint64_t *mask, void *stream);
/* S independent score-sorted lists in one mask launch + one sweep launch (one workgroup per list) -- the per-(class, image) calls of
 * functions/predict_bbox.py:29-55 batched.  seg: DEVICE int64 [S][3] = {first row of the list in boxes / keep, its length n, first
 * word of its mask in mask_ws (a list owns n * ceil(n / 64) words)}; max_n = the longest list.  keep [all rows]: each list's kept
 * indices, LOCAL to the list, from its first row on; num_out [S]: the counts. */
int scda_nms_segments_hip(const float *boxes, const long long *seg, int S, int max_n, float thresh, void *mask_ws, int64_t *keep,
                          int64_t *num_out, void *stream);

/* ------------------------------------------------------------ RoIPool ---- */
/* replaces  int roi_pooling_forward_cuda(int ph,int pw,float scale, THCudaTensor* features,
 *               THCudaTensor* rois, THCudaTensor* output, THCudaIntTensor* argmax)
 *           extensions/_roi_pooling/src/roi_pooling_cuda.h:1-2, roi_pooling_kernel.cu:24-125
 * features [B,C,H,W], rois [R,5] (batch,x1,y1,x2,y2), out [R,C,PH,PW],
 * argmax int32 [R,C,PH,PW] (flat index into features, -1 = empty bin; may be NULL) */
int scda_roi_pool_fwd_hip(const float *features, const float *rois, int R, int B, int C, int H, int W, int PH, int PW,
                          float spatial_scale, float *out, int32_t *argmax, void *stream);
/* replaces  int roi_pooling_backward_cuda(..., top_grad, rois, bottom_grad, argmax)
 *           roi_pooling_cuda.h:4-5, roi_pooling_kernel.cu:128-234
 * bottom_grad [B,C,H,W] is fully overwritten (need not be zeroed).
 * Summation order per input element: roi ascending, ph ascending, pw ascending
 * -- the reference's order, so the result is bit-identical.                    */
int scda_roi_pool_bwd_hip(const float *top_grad, const int32_t *argmax, const float *rois, int R, int B, int C, int H,
                          int W, int PH, int PW, float spatial_scale, float *bottom_grad, void *stream);

/* ----------------------------------------------------------- RoIAlign ---- */
/* replaces  roi_align_forward_cuda / roi_align_backward_cuda
 *           extensions/_roi_align/src/roi_align_cuda.h:1-5, roi_align_kernel.cu:15-162
 * out [R,C,AH,AW]; bottom_grad [B,C,H,W] must be zeroed by the caller
 * (reference convention, functions/roi_align.py:40-41; the kernel accumulates) */
int scda_roi_align_fwd_hip(const float *features, const float *rois, int R, int B, int C, int H, int W, int AH, int AW,
                           float spatial_scale, float *out, void *stream);
int scda_roi_align_bwd_hip(const float *top_grad, const float *rois, int R, int B, int C, int H, int W, int AH,
                           int AW, float spatial_scale, float *bottom_grad, void *stream);
/* the same operator with the pooled maps stored CHANNEL-MAJOR: out / top_grad are [C,R,AH,AW].  No reference counterpart: it is
 * the layout the ResNet-C4 RoI head (models/mask_rcnn/resnet.py:140-146, layer4 on R x 7 x 7 maps) runs in here -- viewed as
 * [1, C, R*AH', AW'] every 1x1 convolution and every batch-norm of the head sees one long contiguous row per channel instead of
 * R pieces of 49 floats (see the row_period argument of the conv entry points for the 3x3 convolutions). */
int scda_roi_align_cmajor_fwd_hip(const float *features, const float *rois, int R, int B, int C, int H, int W, int AH, int AW,
                                  float spatial_scale, float *out, void *stream);
int scda_roi_align_cmajor_bwd_hip(const float *top_grad, const float *rois, int R, int B, int C, int H, int W, int AH,
                                  int AW, float spatial_scale, float *bottom_grad, void *stream);

/* --------------------------------------------------------- focal loss ---- */
/* replaces the four functions of extensions/_focal_loss/src/focal_loss_cuda.h:2-43
 * N = rows*num_classes; logits [rows,num_classes]; targets int32 [rows]
 * (-1 ignore, 0 background, 1..C foreground)                                  */
int scda_focal_sigmoid_fwd_hip(int N, const float *logits, const int32_t *targets, float weight_pos, float gamma,
                               float alpha, int num_classes, float *losses /*[N]*/, void *stream);
int scda_focal_sigmoid_bwd_hip(int N, const float *logits, const int32_t *targets, float *dX /*[N]*/,
                               float weight_pos, float gamma, float alpha, int num_classes, void *stream);
int scda_focal_softmax_fwd_hip(int N, const float *logits, const int32_t *targets, float weight_pos, float gamma,
                               float alpha, int num_classes, float *losses /*[rows]*/, float *priors /*[N]*/,
                               void *stream);
int scda_focal_softmax_bwd_hip(int N, const float *logits, const int32_t *targets, float *dX /*[N]*/,
                               float weight_pos, float gamma, float alpha, int num_classes,
                               const float *priors /*[N]*/, float *buff /*[rows]*/, void *stream);

/* -------------------------------------------------------- box overlaps ---- */
/* replaces  int gpu_iou_overlaps(THCudaTensor* b1, THCudaTensor* b2, THCudaTensor* out)
 *           extensions/_bbox_helper/src/bbox_helper_cuda.h:1, cuda/iou_overlap_kernel.cu:33-100
 * (no +1, union clamped to >= 1)                                              */
int scda_iou_overlaps_hip(const float *b1, const float *b2, int size_bbox, int n1, int n2, float *out, void *stream);
/* replaces  cython_bbox.bbox_overlaps(boxes f32[N,4], query f32[K,4]) -> f32[N,K]
 *           extensions/_cython_bbox/cython_bbox.pyx:32-73   (no +1, 0 unless iw>0 and ih>0) */
int scda_bbox_overlaps_hip(const float *boxes, int N, const float *query, int K, float *out, void *stream);

/* ------------------------------------------ box logic on the device ---- */
/* What the reference computes in numpy on the host between the RPN and the RCNN head.  The host keeps
 * the two jobs whose results are observable behaviour of the reference: drawing from numpy's global RNG (it gets two counts
 * back) and ranking scores with numpy's argpartition / argsort.
 *
 * functions/anchor_target.py:38-64 -- labels of one image before sub-sampling (-1 ignore, 0 bg, 1 fg), ascending index lists
 * of the positives / negatives, counts = {#pos, #neg}.  anchors [KA,4] fp32, gts [G,gt_stride>=4] fp32; all buffers are the
 * caller's (best_iou f32 [KA], best_gt i32 [KA], gt_best u32 [G], labels i8 [KA], pos_list / neg_list i32 [KA], counts i32 [2]) */
int scda_anchor_label_hip(const float *anchors, int KA, const float *gts, int G, int gt_stride, float neg_thresh, float pos_thresh,
                          float min_gt_best, float *best_iou, int *best_gt, unsigned *gt_best, signed char *labels, int *pos_list,
                          int *neg_list, int *counts, void *stream);
/* :66-107 -- drop the surplus the host drew (drop_* index INTO pos_list / neg_list, as np.random.choice returns them) and emit
 * cls_targets int64 [A,fh,fw], loc_targets / loc_masks fp32 [4A,fh,fw]; anchors64 = the float64 anchor grid [KA,4] */
int scda_anchor_finalize_hip(signed char *labels, const int *best_gt, const int *pos_list, const int *drop_pos, int n_drop_pos,
                             const int *neg_list, const int *drop_neg, int n_drop_neg, const double *anchors64, const float *gts,
                             int gt_stride, int A, int fh, int fw, long long *cls_targets, float *loc_targets, float *loc_masks,
                             void *stream);
/* functions/rpn_proposal.py:36-60 for the n candidates the host ranked (order i32 [n], anchor indices): decode + clip in float64,
 * props5 fp32 [n,5] = (x1,y1,x2,y2,score), ok u8 [n] = the roi_min_size test.  loc [4A,fh,fw] / prob [2A,fh,fw] are the RPN's
 * NCHW outputs of that image; exp_wh f32 [n,2] = np.exp of the candidates' (dw, dh) evaluated by numpy on the host (numpy's
 * float32 exp is not correctly rounded: only numpy reproduces it).  Then scda_nms_valid_hip(props5, ok, ...) and
 * scda_proposal_gather_hip. */
int scda_proposal_decode_hip(const int *order, const float *exp_wh, int n, const double *anchors64, const float *loc, const float *prob, int A, int fh,
                             int fw, double img_h, double img_w, double min_size, float *props5, unsigned char *ok, void *stream);
/* out6 [max_rows,6] rows i < min(max_rows, *num_keep) = (image_index, props5[keep[i]]) */
int scda_proposal_gather_hip(const float *props5, const long long *keep, const long long *num_keep, float image_index, int max_rows,
                             float *out6, void *stream);
/* RoI sampling for the RCNN head with the candidates resident on the device (functions/proposal_target.py:38-62, one image).
 * Step 1: candidates = the n_prop proposals (rows (b, x1, y1, x2, y2, ...), stride prop_stride) followed by the G ground-truth
 * boxes (rows (x1, y1, x2, y2, class), stride gt_stride), clipped to the image (utils/bbox_helper.py:105-110) -> rois [n_prop+G,4];
 * per candidate the first best gt and its IoU (cython_bbox.pyx:32-73 arithmetic); labels 1 (IoU > pos_thresh) / 0 (neg_lo <= IoU <
 * neg_hi, not foreground) / -1; pos_list / neg_list = the ascending index lists np.where returns, counts [2] their lengths.  The
 * host then orders the negatives as the reference's Python-set arithmetic does, draws np.random.choice, and evaluates the <= 128
 * foreground rows' regression targets with numpy (its float32 log is numpy's own routine). */
int scda_proposal_match_hip(const float *props, int n_prop, int prop_stride, const float *gts, int G, int gt_stride, float img_h,
                            float img_w, float pos_thresh, float neg_hi, float neg_lo, float *rois, float *best_iou, int *best_gt,
                            signed char *labels, int *pos_list, int *neg_list, int *counts, void *stream);
/* Step 2 (:64-136): sel i32 [R] = sampled candidate indices, gt_of i32 [R] = matched gt (-1: background), enc f32 [R,4] = the
 * foreground rows' normalised targets -> rois5 [R,5] = (image_index, box), labels int64 [R], loc_targets / loc_weights [R, 4*C]. */
int scda_proposal_finalize_hip(const float *cand_rois, const int *sel, const int *gt_of, const float *enc, const float *gts,
                               int gt_stride, int R, int num_classes, float image_index, float *rois5, long long *labels,
                               float *loc_targets, float *loc_weights, void *stream);

/* ------------------------------------------------- convolution / GEMM ---- */
/* The reference reaches these through torch.nn (cuDNN / cuBLAS): nn.Conv2d in
 * models/faster_rcnn/vgg_adver_expansion_cluster.py:101-114 (VGG body),
 * models/head.py:13-18 (RPN), models/faster_rcnn/common_net.py:59-80,251-293 (GAN blocks);
 * nn.Linear in vgg_adver_expansion_cluster.py:46-60 (FC6/FC7/heads).
 * Here they are fp32-MFMA implicit-GEMM kernels; tensors are NCHW fp32, weights
 * [Cout,Cin,KH,KW] exactly as the reference's state_dict stores them.
 * Supported (KH,KW,stride): (3,3,1) (3,3,2) (1,1,1); any padding.
 * act: 0 none, 1 ReLU, 2 LeakyReLU(slope) fused into the epilogue.
 * ws: workspace of scda_conv2d_workspace_bytes(...) bytes (split-K slabs).      */
size_t scda_conv2d_workspace_bytes(int batch, int Cin, int IH, int IW, int Cout, int KH, int KW, int S, int P);
/* GEMM-ready weight: [Cout,Cin,KH,KW] -> the A operand of the forward (for_dgrad = 0, M = Cout, reduced channels
 * C = Cin) or data-gradient (for_dgrad = 1, M = Cin, C = Cout) implicit GEMM.  The layout is private to the library:
 *   C % 16 == 0 : [K][mpad] (M contiguous, padded with zero columns to the 64/128-row tile), K ordered
 *                 channel-block major so that a 16-deep K-slab is 16 consecutive channels at one filter tap
 *   otherwise   : [M][K], K tap-major
 * `out` holds scda_conv2d_packed_elems(...) floats.  Re-pack whenever the weight changes. */
size_t scda_conv2d_packed_elems(int Cout, int Cin, int KH, int KW, int for_dgrad);
int scda_conv2d_pack_weight_hip(const float *w, float *out, int Cout, int Cin, int KH, int KW, int for_dgrad,
                                void *stream);
/* The same packing for MANY weights in one launch (all conv layers of an optimiser group, right after its Adam step), one
 * workgroup per tile, both sides coalesced through LDS: desc = n rows of 7 int64 {source offset in floats from `base`, destination
 * offset in floats from `out`, Cout, Cin, KH*KW, for_dgrad, first tile id}, destinations ascending and back to back, tile ids
 * consecutive: a row owns scda_conv2d_pack_tiles(...) of them; n_tiles = their sum.  for_dgrad 2 / 3 in a row (and in
 * scda_conv2d_packed_elems / scda_conv2d_pack_tiles): the Winograd kernel's transformed filters for the forward / the data gradient
 * (3x3 only; see scda_conv2d_wino_pack_hip below). */
long long scda_conv2d_pack_tiles(int Cout, int Cin, int KH, int KW, int for_dgrad);
int scda_conv2d_pack_weights_batched_hip(const float *base, float *out, const long long *desc, int n, long long n_tiles,
                                         void *stream);
/* wp = pack(w, 0) */
/* row_period (all conv entry points): 0 = plain image.  > 0: the image [batch, C, IH, IW] is a vertical STACK of independent
 * maps of `row_period` rows each (IH % row_period == 0; stride 1, 2*P == K-1) and filter taps must not reach from one map into
 * the next -- what a batch of R maps [R, C, period, IW] computes, on the channel-major layout [1, C, R*period, IW].  A call
 * that cannot honour it fails with SCDA_EINVAL. */
int scda_conv2d_fwd_hip(const float *x, const float *wp, const float *bias /*[Cout] or NULL*/, float *y, int batch,
                        int Cin, int IH, int IW, int Cout, int KH, int KW, int S, int P, int row_period, int act, float slope,
                        void *ws, size_t ws_bytes, void *stream);
/* dx [batch,Cin,IH,IW] = conv-transpose of dy [batch,Cout,OH,OW] (fully overwritten); wt = pack(w, 1) */
int scda_conv2d_dgrad_hip(const float *dy, const float *wt, float *dx, int batch, int Cin, int IH, int IW, int Cout,
                          int KH, int KW, int S, int P, int row_period, void *ws, size_t ws_bytes, void *stream);
/* ... with the gradient of the activation that PRODUCED this conv's input folded into the epilogue (replaces one elementwise
 * pass of the reference's autograd: ReLU / LeakyReLU backward of models/faster_rcnn/vgg_adver_expansion_cluster.py:108-111,
 * common_net.py:251-262): dx = dgrad(dy) * (act_src > 0 ? 1 : act_slope); act_src = the conv's input x, or NULL */
int scda_conv2d_dgrad_act_hip(const float *dy, const float *wt, float *dx, int batch, int Cin, int IH, int IW, int Cout,
                              int KH, int KW, int S, int P, int row_period, const float *act_src, float act_slope, void *ws,
                              size_t ws_bytes, void *stream);
/* the same for a conv with <= 4 input channels (image-side layers), direct form, HBM-bound on dy; w = the UNPACKED weight */
int scda_conv2d_dgrad_small_cin_hip(const float *dy, const float *w, float *dx, int batch, int Cin, int IH, int IW, int Cout,
                                    int KH, int KW, int S, int P, void *stream);
/* dw [Cout,Cin,KH,KW] (+)= sum over batch and pixels; deterministic split-K (no atomics) */
int scda_conv2d_wgrad_hip(const float *dy, const float *x, float *dw, int batch, int Cin, int IH, int IW, int Cout,
                          int KH, int KW, int S, int P, int row_period, int accumulate, void *ws, size_t ws_bytes, void *stream);
/* the same plus the bias gradient db[Cout] (+)= sum over batch and pixels of dy, fused: the row sums ride along on the
 * operand fragments of the weight-gradient GEMM and are finished by its split-K reduce (no extra launches, no second read
 * of dy).  Only when scda_conv2d_wgrad_bias_fusable(...) != 0 (OH*OW % 16 == 0, 16-byte aligned dy); otherwise call
 * scda_conv2d_wgrad_hip + scda_bias_grad_nchw_hip. */
int scda_conv2d_wgrad_bias_fusable(int batch, int Cout, int OH, int OW, const float *dy);
int scda_conv2d_wgrad_bias_hip(const float *dy, const float *x, float *dw, float *db, int batch, int Cin, int IH, int IW,
                               int Cout, int KH, int KW, int S, int P, int row_period, int accumulate, int db_accumulate, void *ws,
                               size_t ws_bytes, void *stream);

/* Winograd F(2x2, 3x3) form of the stride-1, pad-1 3x3 convolution on the fp32 MFMA (csrc/conv_wino.hip): 2.25x fewer matrix
 * operations than the implicit GEMM above, fp32 throughout (results differ from the direct form by rounding only: the transforms
 * use the exact constants 0, +-1, +-1/2).  What cuDNN picks for the same nn.Conv2d layers of the reference
 * (vgg_adver_expansion_cluster.py:101-114, head.py:13, common_net.py:59-80).
 *   scda_conv2d_wino_supported: C % 8 == 0, H and W even (8 x 32-pixel blocks, partial on the right / bottom edge), per-image
 *       tensors below 2 GB
 *   u = scda_conv2d_wino_pack_hip(w [Cout,Cin,3,3], for_dgrad): the transformed filters G g G^T in the kernel's MFMA fragment
 *       order, scda_conv2d_wino_packed_elems floats; for_dgrad = 1: the data gradient's filters (rows = Cin, rotated by 180 degrees)
 *   scda_conv2d_wino_hip: y [batch,M,H,W] = act(conv3x3(x [batch,C,H,W]) + bias), optionally * act'(mask_src) as
 *       scda_conv2d_dgrad_act_hip does; forward: (C, M) = (Cin, Cout), u = pack(w, 0); data gradient: x = dy, (C, M) = (Cout, Cin),
 *       u = pack(w, 1).  for_dgrad only labels the launch for scda_prof_*.  ws: split-K slabs (scda_conv2d_workspace_bytes). */
int scda_conv2d_wino_supported(int batch, int C, int H, int W, int M);
size_t scda_conv2d_wino_packed_elems(int Cout, int Cin, int for_dgrad);
int scda_conv2d_wino_pack_hip(const float *w, float *out, int Cout, int Cin, int for_dgrad, void *stream);
int scda_conv2d_wino_hip(const float *x, const float *u, const float *bias, float *y, int batch, int C, int H, int W, int M, int act,
                         float slope, const float *mask_src, float mask_slope, int for_dgrad, void *ws, size_t ws_bytes, void *stream);

/* conv3x3 + bias + activation + nn.MaxPool2d(2, 2) in ONE launch (a Winograd tile is a pooling window): pool_y [batch,M,H/2,W/2] and
 * pool_idx (uint8 winner 0..3, scda_maxpool2x2_fwd_hip's convention: the backward is scda_maxpool2x2_bwd[_relu]_hip as usual); the
 * full-resolution map is never written.  The pools of vgg_adver_expansion_cluster.py:101-114 behind conv1_2 / 2_2 / 3_3 / 4_3. */
/* the same on x [1, C, maps * 7, 7] read as a vertical STACK of `maps` independent 7 x 7 maps (row period 7, see scda_conv2d_fwd_hip's
 * row_period: the channel-major RoI head of models/mask_rcnn/resnet.py:131-148): four maps per pixel block, as 8 x 8 each with row /
 * column 7 discarded; y [1, M, maps * 7, 7].  scda_conv2d_wino_stacked_supported: C % 8 == 0, tensors below 2 GB. */
int scda_conv2d_wino_stacked_supported(int maps, int C, int M);
int scda_conv2d_wino_stacked_hip(const float *x, const float *u, const float *bias, float *y, int maps, int C, int M, int act, float slope,
                                 const float *mask_src, float mask_slope, int for_dgrad, void *ws, size_t ws_bytes, void *stream);
int scda_conv2d_wino_pool_hip(const float *x, const float *u, const float *bias, float *pool_y, unsigned char *pool_idx, int batch, int C,
                              int H, int W, int M, int act, float slope, void *stream);
/* ... and the weight gradient in the same (transposed) algorithm: dw [Cout,Cin,3,3] (+)= G^T [ sum over 2x2 tiles (A dy A^T) .*
 * (B^T x B) ] G, db [Cout] (+)= sum of dy (fused, may be NULL); deterministic split-K like scda_conv2d_wgrad_hip.
 * scda_conv2d_wino_wgrad_supported: >= 32 channels on both sides, H and W even (K-slabs of 2 x 16 pixels, partial at the right edge). */
int scda_conv2d_wino_wgrad_supported(int batch, int Cin, int H, int W, int Cout);
int scda_conv2d_wino_wgrad_hip(const float *dy, const float *x, float *dw, float *db, int batch, int Cin, int H, int W, int Cout,
                               int accumulate, int db_accumulate, void *ws, size_t ws_bytes, void *stream);
/* ... on dy [1, Cout, maps * 7, 7] / x [1, Cin, maps * 7, 7] read as stacks of `maps` independent 7 x 7 maps (row period 7, see
 * scda_conv2d_wino_stacked_hip): a K-slab is one tile row of a pair of maps */
int scda_conv2d_wino_wgrad_stacked_supported(int maps, int Cin, int Cout);
int scda_conv2d_wino_wgrad_stacked_hip(const float *dy, const float *x, float *dw, float *db, int maps, int Cin, int Cout, int accumulate,
                                       int db_accumulate, void *ws, size_t ws_bytes, void *stream);

/* C[M,N] (row stride ldc) (+)= op(A) op(B) (+ bias) -> act
 * trans_a = 0: A is [M,K] row-major (lda);  1: A is stored [K,M]
 * trans_b = 0: B is [N,K] row-major (ldb);  1: B is stored [K,N]
 * nn.Linear forward  y = x W^T + b : A=x, B=W, trans_a=0, trans_b=0, bias_on_n=1        */
size_t scda_gemm_workspace_bytes(int M, int N, int K);
int scda_gemm_hip(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc,
                  int trans_a, int trans_b, const float *bias, int bias_on_n, int act, float slope, int accumulate,
                  void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------ layer kernels ---- */
/* HBM-bound pieces the reference reaches through torch.nn / torch.nn.functional.
 * `grad_scalar` arguments are DEVICE pointers to the upstream scalar gradient so
 * that no loss value ever has to visit the host.                                */
/* nn.MaxPool2d(2,2): vgg_adver_expansion_cluster.py:106.  idx uint8 = winner 0..3 */
int scda_maxpool2x2_fwd_hip(const float *x, float *y, uint8_t *idx, int planes, int H, int W, void *stream);
int scda_maxpool2x2_bwd_hip(const float *dy, const uint8_t *idx, float *dx, int planes, int H, int W, void *stream);
/* nn.MaxPool2d(3, stride 2, padding 1), forward (ResNet stem, models/mask_rcnn/resnet.py:120; frozen there, so no backward);
 * y [planes, (H-1)/2+1, (W-1)/2+1] */
int scda_maxpool3x3s2_fwd_hip(const float *x, float *y, int planes, int H, int W, void *stream);
/* y = relu(a + b): residual join of a ResNet block (resnet.py:104-105) */
int scda_add_relu_hip(const float *a, const float *b, float *y, long long n, void *stream);
/* max-pool backward + backward of the ReLU in front of the pool (a window's winner is > 0 iff the pooled value y_pooled is) */
int scda_maxpool2x2_bwd_relu_hip(const float *dy, const uint8_t *idx, const float *y_pooled, float *dx, int planes, int H, int W,
                                 void *stream);
/* mode: 0 ReLU, 1 LeakyReLU(slope), 2 tanh, 3 sigmoid; backward takes the forward OUTPUT y */
int scda_act_fwd_hip(const float *x, float *y, long long n, int mode, float slope, void *stream);
int scda_act_bwd_hip(const float *dy, const float *y, float *dx, long long n, int mode, float slope, void *stream);
/* y = alpha*a + beta*b (b may be NULL) */
int scda_axpby_hip(const float *a, const float *b, float *y, long long n, float alpha, float beta, void *stream);
/* nn.Dropout(p): mask[i] = keep ? 1 : 0 from a counter-based generator; y = mask ? x*scale : 0 */
int scda_dropout_mask_hip(uint8_t *mask, long long n, float p, uint64_t seed, void *stream);
int scda_dropout_apply_hip(const float *x, const uint8_t *mask, float *y, long long n, float scale, void *stream);
/* nn.Dropout with the keep decision recomputed from (seed, index) -- forward and backward, no mask tensor; relu_src (backward,
 * may be NULL): the dropout's input when it is a ReLU output, whose gradient is then applied in the same pass */
int scda_dropout_seeded_hip(const float *x, float *y, long long n, float p, uint64_t seed, float scale, const float *relu_src,
                            void *stream);
/* out1 (+)= scale * sum_c w[c] * mean_i BCE(sigmoid(x[c][i]), t[c or 0][i]): the per-cluster adversarial loss terms of
 * tools/faster_rcnn_train_val.py:584-600,675-687,723-732 in one launch (x [C,n] logits; t [t_rows,n], t_rows 1 or C; w [C] or NULL);
 * prob_out [C,n] (may be NULL) receives sigmoid(x) for the backward, which returns d out1 / d x * grad_scalar */
int scda_sigmoid_bce_rows_fwd_hip(const float *x, const float *t, int t_rows, const float *w, int C, int n, float scale,
                                  int accumulate, float *prob_out, float *out1, void *stream);
int scda_sigmoid_bce_rows_bwd_hip(const float *prob, const float *t, int t_rows, const float *w, int C, int n, float scale,
                                  const float *grad_scalar, float *dx, void *stream);
size_t scda_bias_grad_workspace_bytes(int C);
int scda_bias_grad_nchw_hip(const float *dy, float *db, int B, int C, int HW, int accumulate, float *ws, void *stream);
int scda_colsum_hip(const float *dy, float *db, int M, int N, int accumulate, void *stream);
/* F.cross_entropy(ignore_index), mean over valid rows (faster_rcnn_adver_expansion_reweight_cluster.py:49,63)
 * out2[0] = loss, out2[1] = number of valid rows; probs [R,C] is kept for the backward */
int scda_softmax_ce_fwd_hip(const float *logits, const int64_t *targets, int R, int C, int ignore_index, float *probs,
                            float *out2, void *stream);
int scda_softmax_ce_bwd_hip(const float *probs, const int64_t *targets, int R, int C, int ignore_index,
                            const float *fwd_out2, const float *grad_scalar, float *dlogits, void *stream);
int scda_row_softmax_hip(const float *x, float *y, int R, int C, void *stream);
/* top-1 accuracy in percent over rows whose target != ignore_index (...reweight_cluster.py:249-267) */
int scda_accuracy_hip(const float *logits, const int64_t *targets, int R, int C, int ignore_index, float *out1,
                      void *stream);
/* smooth_l1_loss_with_sigma(pred*mask, target, sigma) * scale  (...reweight_cluster.py:238-246; mask may be NULL) */
size_t scda_smooth_l1_workspace_bytes(void);
int scda_smooth_l1_fwd_hip(const float *pred, const float *mask, const float *target, long long n, float sigma,
                           float scale, float *partial_ws, float *out1, void *stream);
int scda_smooth_l1_bwd_hip(const float *pred, const float *mask, const float *target, long long n, float sigma,
                           float scale, const float *grad_scalar, float *dpred, void *stream);
/* nn.InstanceNorm2d(affine=False) with optional fused activation (act 0/1/2 as for conv) : common_net.py:69-72,288-290 */
int scda_instnorm_fwd_hip(const float *x, float *y, float *mean, float *rstd, int planes, int HW, float eps, int act,
                          float slope, void *stream);
int scda_instnorm_bwd_hip(const float *dy, const float *x, const float *mean, const float *rstd, float *dx, int planes,
                          int HW, int act, float slope, void *stream);
/* The tail of a residual block, x + Dropout(InstanceNorm(h)) (INSResBlock, common_net.py:59-80), as ONE launch each way: the norm of
 * `x`, nn.Dropout(p) with the keep decision of element i recomputed from (seed, i) exactly as scda_dropout_seeded_hip makes it, plus
 * `residual` -- and the matching gradient w.r.t. x (the residual's gradient is dy itself).  Bit-identical to the three launches. */
int scda_instnorm_drop_add_fwd_hip(const float *x, const float *residual, float *y, float *mean, float *rstd, int planes, int HW,
                                   float eps, float p, uint64_t seed, float scale /* 1 / (1 - p) */, void *stream);
int scda_instnorm_drop_bwd_hip(const float *dy, const float *x, const float *mean, const float *rstd, float *dx, int planes, int HW,
                               float p, uint64_t seed, float scale, void *stream);
/* ... with the seed read from DEVICE memory (one uint64): a launch recorded in a hipGraph draws fresh keep decisions on every replay */
int scda_instnorm_drop_add_fwd_dev_hip(const float *x, const float *residual, float *y, float *mean, float *rstd, int planes, int HW,
                                       float eps, float p, const uint64_t *seed_dev, float scale, void *stream);
int scda_instnorm_drop_bwd_dev_hip(const float *dy, const float *x, const float *mean, const float *rstd, float *dx, int planes, int HW,
                                   float p, const uint64_t *seed_dev, float scale, void *stream);
/* nn.BatchNorm2d, training mode, with optional fused activation : common_net.py:214-223 */
size_t scda_batchnorm_workspace_bytes(int B, int C, int HW);   /* 0: the one-workgroup-per-channel form needs none (ws may be NULL) */
int scda_batchnorm_fwd_hip(const float *x, float *y, const float *gamma, const float *beta, float *running_mean,
                           float *running_var, float *save_mean, float *save_rstd, int B, int C, int HW, float eps,
                           float momentum, int act, float slope, float *ws, void *stream);
int scda_batchnorm_bwd_hip(const float *dy, const float *x, const float *gamma, const float *beta, const float *save_mean,
                           const float *save_rstd, float *dx, float *dgamma, float *dbeta, int B, int C, int HW, int act,
                           float slope, int accumulate, float *ws, void *stream);
/* bn3 -> "out += residual" -> ReLU of a bottleneck (models/mask_rcnn/resnet.py:95-104) inside the batch norm's pass:
 * y = relu(bn_train(x) + residual), statistics and running-stat update as scda_batchnorm_fwd_hip; the backward gates dy by y > 0,
 * writes the gated gradient (what the residual branch receives) to d_residual and differentiates the batch norm on it.  Served for the
 * shapes scda_batchnorm_add_relu_ok() accepts (batch 1, HW % 4 == 0, HW <= 40960, 16-byte aligned tensors); SCDA_EINVAL otherwise. */
int scda_batchnorm_add_relu_ok(int B, int HW);
int scda_batchnorm_add_relu_fwd_hip(const float *x, const float *residual, float *y, const float *gamma, const float *beta,
                                    float *running_mean, float *running_var, float *save_mean, float *save_rstd, int B, int C,
                                    int HW, float eps, float momentum, void *stream);
int scda_batchnorm_add_relu_bwd_hip(const float *dy, const float *x, const float *y, const float *gamma, const float *beta,
                                    const float *save_mean, const float *save_rstd, float *dx_or_null, float *d_residual,
                                    float *dgamma, float *dbeta, int B, int C, int HW, int accumulate, void *stream);
/* nn.BatchNorm2d in eval mode (running statistics): out = act((x - mean) * rsqrt(var + eps) * gamma + beta); with dy given,
 * out = the gradient w.r.t. x instead (dy * act'(y) * gamma * rsqrt(var + eps); statistics and affine parameters are constants) */
int scda_batchnorm_eval_hip(const float *x, const float *dy_or_null, float *out, const float *gamma, const float *beta,
                            const float *running_mean, const float *running_var, int B, int C, int HW, float eps, int act,
                            float slope, void *stream);
/* Interpolate(scale_factor=2, 'bilinear', align_corners=True) : common_net.py:160-170 */
int scda_upsample2x_fwd_hip(const float *x, float *y, int planes, int IH, int IW, void *stream);
int scda_upsample2x_bwd_hip(const float *dy, float *dx, int planes, int IH, int IW, void *stream);
/* Instance norm (+ fused activation, or + the residual block's dropout-and-add tail) and the Interpolate behind it as ONE launch
 * (common_net.py:59-80 / :288-289 feeding :279-293 -- in the decoders nothing but the Interpolate reads those norms' outputs):
 * y2 [planes, 2 IH, 2 IW] = upsample2x(instance_norm...(x)), bit-identical to the two launches; mean / rstd as scda_instnorm_fwd_hip
 * (the backward is scda_upsample2x_bwd_hip followed by scda_instnorm_bwd_hip / scda_instnorm_drop_bwd_hip: it needs x, not the small
 * plane).  Planes of 4096 or 16384 elements, IW % 32 == 0, 16-byte aligned tensors (scda_instnorm_up2_supported; SCDA_EINVAL otherwise). */
int scda_instnorm_up2_supported(int IH, int IW);
int scda_instnorm_up2_fwd_hip(const float *x, float *y2, float *mean, float *rstd, int planes, int IH, int IW, float eps, int act,
                              float slope, void *stream);
int scda_instnorm_drop_add_up2_fwd_hip(const float *x, const float *residual, float *y2, float *mean, float *rstd, int planes, int IH,
                                       int IW, float eps, float p, uint64_t seed, float scale /* 1 / (1 - p) */, void *stream);
int scda_instnorm_drop_add_up2_fwd_dev_hip(const float *x, const float *residual, float *y2, float *mean, float *rstd, int planes,
                                           int IH, int IW, float eps, float p, const uint64_t *seed_dev, float scale, void *stream);
/* ... and their backward in one launch: the bilinear gather of dy2 [planes, 2 IH, 2 IW] feeds the norm's gradient in registers;
 * dresidual [planes, IH, IW] = the gathered gradient itself (the residual input's gradient of the tail form).  Bit-identical to
 * scda_upsample2x_bwd_hip followed by scda_instnorm_bwd_hip / scda_instnorm_drop_bwd_hip. */
int scda_instnorm_up2_bwd_hip(const float *dy2, const float *x, const float *mean, const float *rstd, float *dx, int planes, int IH,
                              int IW, int act, float slope, void *stream);
int scda_instnorm_drop_up2_bwd_hip(const float *dy2, const float *x, const float *mean, const float *rstd, float *dx, float *dresidual,
                                   int planes, int IH, int IW, float p, uint64_t seed, float scale, void *stream);
int scda_instnorm_drop_up2_bwd_dev_hip(const float *dy2, const float *x, const float *mean, const float *rstd, float *dx,
                                       float *dresidual, int planes, int IH, int IW, float p, const uint64_t *seed_dev, float scale,
                                       void *stream);
/* F.binary_cross_entropy(p, t), mean : tools/faster_rcnn_train_val.py:584-600,627-628,675-687,723-732 */
int scda_bce_fwd_hip(const float *p, const float *t, int n, float *out1, void *stream);
int scda_bce_bwd_hip(const float *p, const float *t, int n, const float *grad_scalar, float *dp, void *stream);
/* F.avg_pool2d(kernel_size=2, stride=1) of [planes, H+1, W+1] -> [planes, H, W] and its gradient: the pooling half of RoIAlignAvg
 * (extensions/_roi_align/modules/roi_align.py:18-30) */
int scda_avg2x2s1_fwd_hip(const float *x, float *y, int planes, int H, int W, void *stream);
int scda_avg2x2s1_bwd_hip(const float *dy, float *dx, int planes, int H, int W, void *stream);
/* nn.AvgPool2d(full extent) and torch.mean(x, 1) */
int scda_gap_fwd_hip(const float *x, float *y, int planes, int HW, void *stream);
int scda_gap_bwd_hip(const float *dy, float *dx, int planes, int HW, void *stream);
int scda_row_mean_hip(const float *x, float *y, int R, int C, void *stream);
/* torch.optim.Adam step on one flat bucket (tools/faster_rcnn_train_val.py:305-316); step counts from 1 */
int scda_adam_hip(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, void *stream);
/* the same step with the cap of the (grid-stride) launch given explicitly; 0 = the library's choice (512 workgroups) */
int scda_adam_limited_hip(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int step, int max_blocks, void *stream);

/* ---- data path: one image from 8-bit interleaved pixels to the network's input tensor ------------------------------
 * datasets/example_dataset.py:76-100,106-131 and datasets/target_dataset.py:23-71: PIL `img.resize((new_w, new_h))`, optional
 * FLIP_LEFT_RIGHT, ToTensor (/ 255) and Normalize ((x - mean) / std).  Pillow's resize is a two-pass separable convolution in
 * fixed point (22 fractional bits) with an 8-bit intermediate image; the caller builds its per-output-coordinate tables
 * (scda_amd/device_image.py: bounds = [xmin, n] pairs, kk = n weights each, `ksize` entries per coordinate) and the two launches
 * reproduce it bit for bit.  src [H, W, C] uint8 (C = 1 or 3: modes L and RGB; PIL pre-multiplies alpha modes before resizing, those are not supported), tmp >= scda_image_resize_tmp_bytes(rows, out_w, C) bytes holds the
 * horizontally resized rows [row0, row0 + rows) -- the rows the vertical tables reach --, out [C, out_h, out_w] float.
 * normalize = 0 stops after ToTensor.  All pointers are device pointers. */
size_t scda_image_resize_tmp_bytes(int rows, int out_w, int C);
int scda_image_resize_normalize_hip(const unsigned char *src, int H, int W, int C, const int *bounds_h, const int *kk_h,
                                    int ksize_h, int out_w, const int *bounds_v, const int *kk_v, int ksize_v, int out_h,
                                    int row0, int rows, unsigned char *tmp, size_t tmp_bytes, int normalize, float mean,
                                    float stdv, int flip, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SCDA_OPS_H */
