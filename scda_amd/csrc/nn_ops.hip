// nn_ops.hip -- the HBM-bound layer kernels of the SCDA step for gfx950:
// max-pool, activation gradients, bias gradients, dropout, soft-max cross-entropy,
// smooth-L1, instance / batch norm, bilinear x2 up-sampling, tanh / sigmoid, BCE,
// global average pool and the fused Adam update.  All reductions are
// deterministic (fixed-shape trees, no float atomics).
#include <float.h>

#include "common.h"

namespace scda {

// ------------------------------------------------------------ reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum over the workgroup; result valid in every thread. blockDim.x multiple of 64, <= 1024
__device__ __forceinline__ float block_sum(float v, float *red /* >= 16 floats of LDS */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();  // protect red[] against a previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];  // fixed order -> deterministic
    return t;
}

// ------------------------------------------------------------- max pool -----
// 2x2 stride 2 (models/faster_rcnn/vgg_adver_expansion_cluster.py:106). idx: 0..3 = winner (dy*2+dx)
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                           uint8_t *__restrict__ idx, const long long total,
                                                           const int H, const int W, const int OH, const int OW) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)blockDim.x * gridDim.x) {
        const int ox = (int)(i % OW);
        const long long r = i / OW;
        const int oy = (int)(r % OH);
        const long long pc = r / OH;
        const float *p = x + (pc * H + 2 * oy) * W + 2 * ox;
        float2 a, b;
        if (W & 1) {   // odd width: rows are not 8-byte aligned (floor mode: the last column / row is simply never read)
            a = make_float2(p[0], p[1]);
            b = make_float2(p[W], p[W + 1]);
        } else {
            a = *reinterpret_cast<const float2 *>(p);
            b = *reinterpret_cast<const float2 *>(p + W);
        }
        float m = a.x; int k = 0;
        if (a.y > m || a.y != a.y) { m = a.y; k = 1; }
        if (b.x > m || b.x != b.x) { m = b.x; k = 2; }
        if (b.y > m || b.y != b.y) { m = b.y; k = 3; }
        y[i] = m;
        idx[i] = (uint8_t)k;
    }
}

// relu_y (may be null): the pool's OUTPUT; when given, the gradient of a ReLU that produced the pool's input is applied too --
// the winner of a window is > 0 exactly when the pooled value is (fused "max-pool backward + ReLU backward")
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float *__restrict__ dy, const uint8_t *__restrict__ idx,
                                                           float *__restrict__ dx, const long long total, const int H,
                                                           const int W, const int OH, const int OW,
                                                           const float *__restrict__ relu_y) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)blockDim.x * gridDim.x) {
        const int ox = (int)(i % OW);
        const long long r = i / OW;
        const int oy = (int)(r % OH);
        const long long pc = r / OH;
        float g = dy[i];
        if (relu_y && !(relu_y[i] > 0.f)) g = 0.f;
        const int k = idx[i];
        float *p = dx + (pc * H + 2 * oy) * W + 2 * ox;
        if (W & 1) {
            p[0] = k == 0 ? g : 0.f; p[1] = k == 1 ? g : 0.f;
            p[W] = k == 2 ? g : 0.f; p[W + 1] = k == 3 ? g : 0.f;
            if (ox == OW - 1) { p[2] = 0.f; p[W + 2] = 0.f; }                  // the column floor mode dropped
        } else {
            *reinterpret_cast<float2 *>(p) = make_float2(k == 0 ? g : 0.f, k == 1 ? g : 0.f);
            *reinterpret_cast<float2 *>(p + W) = make_float2(k == 2 ? g : 0.f, k == 3 ? g : 0.f);
        }
        if ((H & 1) && oy == OH - 1) {                                         // the row floor mode dropped
            p[2 * W] = 0.f; p[2 * W + 1] = 0.f;
            if ((W & 1) && ox == OW - 1) p[2 * W + 2] = 0.f;
        }
    }
}

// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) -- the ResNet stem pool (models/mask_rcnn/resnet.py:120); forward only: the
// stem and layer1 are frozen in the reference (:230-238), nothing is differentiated through it.  -inf padding semantics.
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, const long long total,
                                                             const int H, const int W, const int OH, const int OW) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)blockDim.x * gridDim.x) {
        const int ox = (int)(i % OW);
        const long long r = i / OW;
        const int oy = (int)(r % OH);
        const long long pc = r / OH;
        const float *p = x + pc * H * W;
        float m = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = 2 * oy - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = 2 * ox - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const float v = p[(size_t)iy * W + ix];
                if (v > m || v != v) m = v;
            }
        }
        y[i] = m;
    }
}

// y = relu(a + b): the residual join of a bottleneck (resnet.py:104-105) in one pass; backward = ReLU mask on y for both inputs
__global__ __launch_bounds__(256) void add_relu_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ y,
                                                       const long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)blockDim.x * gridDim.x) {
        const float v = a[i] + b[i];
        y[i] = v > 0.f ? v : 0.f;
    }
}

// ----------------------------------------------------------- elementwise ----
// mode 0: relu'(y) ; 1: leaky'(y, slope) ; 2: tanh'(y) = 1-y^2 ; 3: sigmoid'(y) = y(1-y)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                      float *__restrict__ dx, const long long n, const int mode,
                                                      const float slope) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)blockDim.x * gridDim.x) {
        const float g = dy[i], v = y[i];
        float d;
        if (mode == 0) d = v > 0.f ? g : 0.f;
        else if (mode == 1) d = v > 0.f ? g : g * slope;
        else if (mode == 2) d = g * (1.f - v * v);
        else d = g * v * (1.f - v);
        dx[i] = d;
    }
}

// mode 0 relu, 1 leaky, 2 tanh, 3 sigmoid (forward)
__global__ __launch_bounds__(256) void act_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                      const long long n, const int mode, const float slope) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)blockDim.x * gridDim.x) {
        const float v = x[i];
        float o;
        if (mode == 0) o = v > 0.f ? v : 0.f;
        else if (mode == 1) o = v > 0.f ? v : v * slope;
        else if (mode == 2) o = tanhf(v);
        else o = 1.f / (1.f + expf(-v));
        y[i] = o;
    }
}

// y = alpha*a + beta*b   (b may be null)
__global__ __launch_bounds__(256) void axpby_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                    float *__restrict__ y, const long long n, const float alpha,
                                                    const float beta) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)blockDim.x * gridDim.x)
        y[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

// counter-based dropout mask: keep with probability (1-p).  splitmix64 of (seed, index).
__device__ __forceinline__ uint32_t mix_hash(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t *__restrict__ mask, const long long n, const float p,
                                                           const uint64_t seed) {
    const uint32_t thr = (uint32_t)((double)p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)p * 4294967296.0);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)blockDim.x * gridDim.x)
        mask[i] = mix_hash(seed, (uint64_t)i) >= thr ? 1 : 0;
}

__global__ __launch_bounds__(256) void dropout_apply_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask,
                                                            float *__restrict__ y, const long long n,
                                                            const float scale) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)blockDim.x * gridDim.x)
        y[i] = mask[i] ? x[i] * scale : 0.f;
}

// nn.Dropout without a mask tensor: keep(i) is recomputed from (seed, i) in the forward AND the backward pass (the same
// generator as dropout_mask_kernel, so a seeded run and a mask-replay run agree).  relu_src (backward only, may be null): the
// dropout's INPUT when that is a ReLU output -- the ReLU's gradient is applied in the same pass.
__global__ __launch_bounds__(256) void dropout_seeded_kernel(const float *__restrict__ x, float *__restrict__ y, const long long n,
                                                             const float p, const uint64_t seed, const float scale,
                                                             const float *__restrict__ relu_src) {
    const uint32_t thr = (uint32_t)((double)p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)p * 4294967296.0);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)blockDim.x * gridDim.x) {
        bool keep = mix_hash(seed, (uint64_t)i) >= thr;
        if (relu_src) keep = keep && relu_src[i] > 0.f;
        y[i] = keep ? x[i] * scale : 0.f;
    }
}

// sum_c w[c] * mean_i BCE(sigmoid(x[c][i]), t[c or 0][i]) * scale -- the per-cluster adversarial terms of
// tools/faster_rcnn_train_val.py:584-600,675-687,723-732 (F.binary_cross_entropy on torch.sigmoid outputs, log clamp -100,
// one mean per cluster row, rows weighted by the patch discriminator) as ONE launch; prob_out keeps sigmoid(x) for the backward.
// accumulate: out[0] += (several groups of one loss).  Single workgroup: C x n is a few thousand elements.
__global__ __launch_bounds__(1024) void sigmoid_bce_rows_fwd_kernel(const float *__restrict__ x, const float *__restrict__ t,
                                                                    const int t_rows, const float *__restrict__ w,
                                                                    const int C, const int n, const float scale,
                                                                    const int accumulate, float *__restrict__ prob_out,
                                                                    float *__restrict__ out) {
    __shared__ float red[16];
    float total = 0.f;
    for (int c = 0; c < C; ++c) {
        const float *tr = t + (t_rows == 1 ? 0 : (size_t)c * n);
        float s = 0.f;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float pi = 1.f / (1.f + expf(-x[(size_t)c * n + i])), ti = tr[i];
            if (prob_out) prob_out[(size_t)c * n + i] = pi;
            const float lp = fmaxf(logf(pi), -100.f), l1p = fmaxf(logf(1.f - pi), -100.f);
            s += -(ti * lp + (1.f - ti) * l1p);
        }
        s = block_sum(s, red);
        total += (w ? w[c] : 1.f) * (s / (float)n);
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + total * scale;
}

// d/dx of the above: chain of BCE'(p) = (p - t) / max(p (1 - p), 1e-12) (torch's clamp) and sigmoid'(p) = p (1 - p)
__global__ __launch_bounds__(256) void sigmoid_bce_rows_bwd_kernel(const float *__restrict__ prob, const float *__restrict__ t,
                                                                   const int t_rows, const float *__restrict__ w,
                                                                   const int C, const int n, const float scale,
                                                                   const float *__restrict__ g, float *__restrict__ dx) {
    const long long total = (long long)C * n;
    for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < total; j += (long long)blockDim.x * gridDim.x) {
        const int c = (int)(j / n), i = (int)(j - (long long)c * n);
        const float pi = prob[j], ti = t[t_rows == 1 ? i : j];
        const float gs = g[0] * scale * (w ? w[c] : 1.f) / (float)n;
        dx[j] = (pi - ti) / fmaxf((1.f - pi) * pi, 1e-12f) * gs * (pi * (1.f - pi));
    }
}

// ------------------------------------------------------- bias gradients -----
// db[c] (+)= sum_{b,p} dy[b][c][p].  Stage 1: grid (C, nsplit) -- each workgroup reduces one contiguous slice of a
// channel's pixels (float4 loads), partial[c][s]; stage 2: one wave per channel adds the slices in fixed order.
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const float *__restrict__ dy, float *__restrict__ partial,
                                                                const int B, const int C, const int HW,
                                                                const int slice) {
    __shared__ float red[16];
    const int c = blockIdx.x, s0 = blockIdx.y * slice;
    const int s1 = min(HW, s0 + slice);
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float *p = dy + ((size_t)b * C + c) * HW;
        if (((HW | s0) & 3) == 0) {
            const float4 *p4 = reinterpret_cast<const float4 *>(p + s0);
            const int n4 = (s1 - s0) >> 2;
            for (int i = threadIdx.x; i < n4; i += blockDim.x) { const float4 v = p4[i]; s += (v.x + v.y) + (v.z + v.w); }
            for (int i = s0 + (n4 << 2) + threadIdx.x; i < s1; i += blockDim.x) s += p[i];
        } else {
            for (int i = s0 + threadIdx.x; i < s1; i += blockDim.x) s += p[i];
        }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[(size_t)c * gridDim.y + blockIdx.y] = s;
}

__global__ __launch_bounds__(256) void bias_grad_finish_kernel(const float *__restrict__ partial, float *__restrict__ db,
                                                               const int C, const int nsplit, const int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int i = 0; i < nsplit; ++i) s += partial[(size_t)c * nsplit + i];
    db[c] = accumulate ? db[c] + s : s;
}

// db[n] (+)= sum_m dy[m][n]    (linear bias).  Workgroup = 32 columns x 8 row groups: thread (g, c) sums rows g, g+8, ...
// of its column (128-byte coalesced row segments), the 8 partial sums are added in fixed order.  (One thread per column
// put 4096 columns on 16 workgroups: 80 us for a 8 MB read.)
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ dy, float *__restrict__ db, const int M,
                                                     const int N, const int accumulate) {
    __shared__ float part[8][33];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    float s = 0.f;
    if (n < N)
        for (int m = g; m < M; m += 8) s += dy[(size_t)m * N + n];
    part[g][c] = s;
    __syncthreads();
    if (g == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += part[i][c];
        db[n] = accumulate ? db[n] + t : t;
    }
}

// ------------------------------------------- soft-max cross entropy (mean) --
// F.cross_entropy(logits[R,C], targets[R], ignore_index) -- faster_rcnn_adver_expansion_reweight_cluster.py:49,63
// single workgroup: R <= ~1e5 rows, C small.  out[0] = mean loss, out[1] = #valid rows, probs[R,C] saved.
__global__ __launch_bounds__(1024) void softmax_ce_fwd_kernel(const float *__restrict__ x, const int64_t *__restrict__ t,
                                                              const int R, const int C, const int ignore,
                                                              float *__restrict__ probs, float *__restrict__ out) {
    __shared__ float red[16];
    float ls = 0.f, cnt = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const float *row = x + (size_t)r * C;
        float mx = -FLT_MAX;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, row[c]);
        float es = 0.f;
        for (int c = 0; c < C; ++c) es += expf(row[c] - mx);
        const float lse = logf(es);
        for (int c = 0; c < C; ++c) probs[(size_t)r * C + c] = expf(row[c] - mx - lse);
        const int tt = (int)t[r];
        if (tt != ignore) { ls += -(row[tt] - mx - lse); cnt += 1.f; }
    }
    ls = block_sum(ls, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) { out[0] = ls / cnt; out[1] = cnt; }
}

// dlogits = (probs - onehot) * g / count for valid rows, 0 otherwise.  g: device scalar (upstream grad)
__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(const float *__restrict__ probs, const int64_t *__restrict__ t,
                                                             const int R, const int C, const int ignore,
                                                             const float *__restrict__ fwd_out, const float *__restrict__ g,
                                                             float *__restrict__ dx) {
    const float scale = g[0] / fwd_out[1];
    const long long n = (long long)R * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)blockDim.x * gridDim.x) {
        const int r = (int)(i / C), c = (int)(i - (long long)r * C);
        const int tt = (int)t[r];
        dx[i] = (tt == ignore) ? 0.f : (probs[i] - (c == tt ? 1.f : 0.f)) * scale;
    }
}

// row soft-max (RPN objectness, C = 2), not differentiated (used through .data in the reference)
__global__ __launch_bounds__(256) void row_softmax_kernel(const float *__restrict__ x, float *__restrict__ y, const int R,
                                                          const int C) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += blockDim.x * gridDim.x) {
        const float *row = x + (size_t)r * C;
        float mx = -FLT_MAX;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, row[c]);
        float es = 0.f;
        for (int c = 0; c < C; ++c) es += expf(row[c] - mx);
        for (int c = 0; c < C; ++c) y[(size_t)r * C + c] = expf(row[c] - mx) / es;
    }
}

// top-1 accuracy in percent over rows with target != ignore (reweight_cluster.py:249-267)
__global__ __launch_bounds__(1024) void accuracy_kernel(const float *__restrict__ x, const int64_t *__restrict__ t,
                                                        const int R, const int C, const int ignore,
                                                        float *__restrict__ out) {
    __shared__ float red[16];
    float ok = 0.f, cnt = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const int tt = (int)t[r];
        if (tt == ignore) continue;
        const float *row = x + (size_t)r * C;
        int best = 0; float bv = row[0];
        for (int c = 1; c < C; ++c) if (row[c] > bv) { bv = row[c]; best = c; }
        ok += (best == tt) ? 1.f : 0.f;
        cnt += 1.f;
    }
    ok = block_sum(ok, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) out[0] = ok * (100.0f / cnt);
}

// ---------------------------------------------------------- smooth L1 -------
// smooth_l1_loss_with_sigma(pred*mask, target, sigma) (reweight_cluster.py:238-246): SUM over elements.
// two-stage deterministic reduction: per-block partials then a 1-block finish.
__global__ __launch_bounds__(256) void smooth_l1_fwd_kernel(const float *__restrict__ pred, const float *__restrict__ mask,
                                                            const float *__restrict__ target, const long long n,
                                                            const float sigma2, float *__restrict__ partial) {
    __shared__ float red[16];
    float s = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)blockDim.x * gridDim.x) {
        const float d = pred[i] * (mask ? mask[i] : 1.f) - target[i];
        const float a = fabsf(d);
        s += (a < 1.f / sigma2) ? d * d * sigma2 * 0.5f : a - 0.5f / sigma2;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void finish_sum_kernel(const float *__restrict__ partial, const int np, const float scale,
                                                         float *__restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += blockDim.x) s += partial[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s * scale;
}

__global__ __launch_bounds__(256) void smooth_l1_bwd_kernel(const float *__restrict__ pred, const float *__restrict__ mask,
                                                            const float *__restrict__ target, const long long n,
                                                            const float sigma2, const float scale,
                                                            const float *__restrict__ g, float *__restrict__ dpred) {
    const float gs = g[0] * scale;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)blockDim.x * gridDim.x) {
        const float m = mask ? mask[i] : 1.f;
        const float d = pred[i] * m - target[i];
        const float a = fabsf(d);
        const float dl = (a < 1.f / sigma2) ? d * sigma2 : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        dpred[i] = dl * m * gs;
    }
}

// ------------------------------------------------------ instance norm -------
// nn.InstanceNorm2d(affine=False, eps) (+ optional fused ReLU / LeakyReLU), one 1024-thread workgroup per (b,c) plane.
// common_net.py:69-72 (INSResBlock), :288-289 (LeakyReLUConvTranspose2d_2)
// HBM-bound: the plane is read ONCE with float4 loads and kept in registers across the mean / centred-variance /
// normalise passes when it fits (VPT float4 per thread: 64x64, 128x128 and 256x256 planes = VPT 1, 4, 16); other sizes
// take the generic path, which re-reads the plane (L2) with scalar loads.  Two-pass statistics (mean, then centred sum of
// squares), fixed reduction order.
__device__ __forceinline__ float inorm_act(float o, const int act, const float slope) {
    if (act == 1) return o > 0.f ? o : 0.f;
    if (act == 2) return o > 0.f ? o : o * slope;
    return o;
}

// Tail of a residual block fused into the norm (INSResBlock: x + Dropout(IN(conv(..))), common_net.py:59-80): with `residual` set the
// kernel writes residual + dropout(normalised value) -- the keep decision of element i from (seed, flat index i) exactly as
// dropout_seeded_kernel makes it -- and the backward applies the same mask to dy before the norm's gradient.  Same arithmetic,
// same bits as the three separate launches (norm, dropout, add); two launches and two passes over the map fewer each way.
struct DropTail {
    const float *residual;   // null: plain instance norm
    unsigned thr;            // keep(i) = mix_hash(seed, i) >= thr
    unsigned long long seed;
    float scale;             // 1 / (1 - p)
    int on;                  // backward: mask dy
    const unsigned long long *seed_ptr;   // non-null: the seed is read from device memory (a launch recorded in a hipGraph gets a new
                                          // seed on every replay without being re-recorded)
};

template <int VPT>
__global__ __launch_bounds__(1024) void instnorm_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                            float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                            const int HW, const float eps, const int act,
                                                            const float slope, const DropTail dt) {
    __shared__ float red[16];
    const size_t base = (size_t)blockIdx.x * HW;
    float mean, rstd;
    const unsigned long long dseed = dt.seed_ptr ? *dt.seed_ptr : dt.seed;
    auto tail = [&](float o, const size_t i) -> float {      // i: flat index into the tensor
        if (!dt.residual) return o;
        const float d = mix_hash(dseed, (uint64_t)i) >= dt.thr ? o * dt.scale : 0.f;
        return d + dt.residual[i];
    };
    if (VPT > 0) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x + base);
        float4 v[VPT > 0 ? VPT : 1];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            v[j] = x4[j * 1024 + threadIdx.x];
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        mean = block_sum(s, red) / (float)HW;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
        rstd = 1.f / sqrtf(block_sum(q, red) / (float)HW + eps);
        float4 *y4 = reinterpret_cast<float4 *>(y + base);
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            float4 o;
            const size_t i0 = base + 4 * (size_t)(j * 1024 + threadIdx.x);
            o.x = tail(inorm_act((v[j].x - mean) * rstd, act, slope), i0);
            o.y = tail(inorm_act((v[j].y - mean) * rstd, act, slope), i0 + 1);
            o.z = tail(inorm_act((v[j].z - mean) * rstd, act, slope), i0 + 2);
            o.w = tail(inorm_act((v[j].w - mean) * rstd, act, slope), i0 + 3);
            y4[j * 1024 + threadIdx.x] = o;
        }
    } else {
        float s = 0.f;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) s += x[base + i];
        mean = block_sum(s, red) / (float)HW;
        float q = 0.f;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) { const float d = x[base + i] - mean; q += d * d; }
        rstd = 1.f / sqrtf(block_sum(q, red) / (float)HW + eps);
        for (int i = threadIdx.x; i < HW; i += blockDim.x) y[base + i] = tail(inorm_act((x[base + i] - mean) * rstd, act, slope), base + i);
    }
    if (threadIdx.x == 0) { mean_out[blockIdx.x] = mean; rstd_out[blockIdx.x] = rstd; }
}

// dx = rstd * (g - mean(g) - xh * mean(g * xh)),  g = dy gated by the fused activation, xh = (x - mean) * rstd.
// VPT > 0: g (and, with CACHE_X, xh) stay in registers between the reduction and the output pass (VPT float4 each);
// the 256x256 planes keep g only (64 VGPRs) and re-read x for the output pass.
// The register-resident form (VPT float4 per thread) as a body two kernels share: instnorm_bwd_kernel reads the incoming gradient
// from memory, instnorm_up2_bwd_kernel (behind the bilinear kernels) hands over the values it has just gathered.  `gload(j)` = the
// thread's j-th float4 of dy (slot j * 1024 + threadIdx.x of the plane).
template <int VPT, bool CACHE_X, class GLoad>
__device__ __forceinline__ void instnorm_bwd_plane(const GLoad &gload, const float *__restrict__ x, const float mean, const float rstd,
                                                   float *__restrict__ dx, const int HW, const int act, const float slope,
                                                   const DropTail dt, const unsigned long long dseed, const size_t base, float *red) {
    auto drop = [&](float g, const size_t i) -> float {      // the fused dropout's gradient (see DropTail)
        if (!dt.on) return g;
        return mix_hash(dseed, (uint64_t)i) >= dt.thr ? g * dt.scale : 0.f;
    };
    auto gate = [&](float g, float xh) -> float {
        if (act == 1) return xh > 0.f ? g : 0.f;
        if (act == 2) return xh > 0.f ? g : g * slope;
        return g;
    };
    const float4 *x4 = reinterpret_cast<const float4 *>(x + base);
    float4 xc[CACHE_X ? VPT : 1], g[VPT];
    auto norm4 = [&](const float4 xv) -> float4 {
        float4 h;
        h.x = (xv.x - mean) * rstd; h.y = (xv.y - mean) * rstd; h.z = (xv.z - mean) * rstd; h.w = (xv.w - mean) * rstd;
        return h;
    };
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const float4 xh = norm4(x4[j * 1024 + threadIdx.x]);
        float4 gv = gload(j);
        const size_t i0 = base + 4 * (size_t)(j * 1024 + threadIdx.x);
        gv.x = drop(gv.x, i0); gv.y = drop(gv.y, i0 + 1); gv.z = drop(gv.z, i0 + 2); gv.w = drop(gv.w, i0 + 3);
        if (CACHE_X) xc[j] = xh;
        g[j].x = gate(gv.x, xh.x); g[j].y = gate(gv.y, xh.y); g[j].z = gate(gv.z, xh.z); g[j].w = gate(gv.w, xh.w);
        s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
        s2 += (g[j].x * xh.x + g[j].y * xh.y) + (g[j].z * xh.z + g[j].w * xh.w);
    }
    const float m1 = block_sum(s1, red) / (float)HW;
    const float m2 = block_sum(s2, red) / (float)HW;
    float4 *d4 = reinterpret_cast<float4 *>(dx + base);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const float4 xh = CACHE_X ? xc[j] : norm4(x4[j * 1024 + threadIdx.x]);
        float4 o;
        o.x = rstd * (g[j].x - m1 - xh.x * m2); o.y = rstd * (g[j].y - m1 - xh.y * m2);
        o.z = rstd * (g[j].z - m1 - xh.z * m2); o.w = rstd * (g[j].w - m1 - xh.w * m2);
        d4[j * 1024 + threadIdx.x] = o;
    }
}

template <int VPT, bool CACHE_X>
__global__ __launch_bounds__(1024) void instnorm_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                            const float *__restrict__ mean_in,
                                                            const float *__restrict__ rstd_in, float *__restrict__ dx,
                                                            const int HW, const int act, const float slope, const DropTail dt) {
    __shared__ float red[16];
    const size_t base = (size_t)blockIdx.x * HW;
    const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];
    const unsigned long long dseed = dt.seed_ptr ? *dt.seed_ptr : dt.seed;
    if constexpr (VPT > 0) {
        const float4 *g4 = reinterpret_cast<const float4 *>(dy + base);
        instnorm_bwd_plane<VPT, CACHE_X>([&](const int j) { return g4[j * 1024 + threadIdx.x]; }, x, mean, rstd, dx, HW, act, slope, dt, dseed, base, red);
    } else {
        auto drop = [&](float g, const size_t i) -> float {
            if (!dt.on) return g;
            return mix_hash(dseed, (uint64_t)i) >= dt.thr ? g * dt.scale : 0.f;
        };
        auto gate = [&](float g, float xh) -> float {
            if (act == 1) return xh > 0.f ? g : 0.f;
            if (act == 2) return xh > 0.f ? g : g * slope;
            return g;
        };
        float s1 = 0.f, s2 = 0.f;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const float xh = (x[base + i] - mean) * rstd;
            const float g = gate(drop(dy[base + i], base + i), xh);
            s1 += g;
            s2 += g * xh;
        }
        const float m1 = block_sum(s1, red) / (float)HW;
        const float m2 = block_sum(s2, red) / (float)HW;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const float xh = (x[base + i] - mean) * rstd;
            dx[base + i] = rstd * (gate(drop(dy[base + i], base + i), xh) - m1 - xh * m2);
        }
    }
}

// --------------------------------------------------------- batch norm -------
// nn.BatchNorm2d in training mode (+ fused LeakyReLU): common_net.py:214-223 (ResDis_cluster).
// one workgroup per channel, statistics over (B, HW).
__global__ __launch_bounds__(256) void batchnorm_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            float *__restrict__ run_mean, float *__restrict__ run_var,
                                                            float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                            const int B, const int C, const int HW, const float eps,
                                                            const float momentum, const int act, const float slope) {
    __shared__ float red[16];
    const int c = blockIdx.x;
    const float n = (float)B * (float)HW;
    // the (image, pixel) pairs of this channel as ONE index space: RoI-sized maps (HW = 49, B = 512) keep all threads busy
    const int BHW = B * HW;
    auto at = [&](const int j) -> size_t { const int b = j / HW; return ((size_t)b * C + c) * HW + (j - b * HW); };
    float s = 0.f;
    for (int j = threadIdx.x; j < BHW; j += blockDim.x) s += x[at(j)];
    const float mean = block_sum(s, red) / n;
    float v = 0.f;
    for (int j = threadIdx.x; j < BHW; j += blockDim.x) { const float d = x[at(j)] - mean; v += d * d; }
    const float var = block_sum(v, red) / n;
    const float rstd = 1.f / sqrtf(var + eps);
    const float ga = gamma[c], be = beta[c];
    for (int j = threadIdx.x; j < BHW; j += blockDim.x) {
        const size_t o_ = at(j);
        float o = (x[o_] - mean) * rstd * ga + be;
        if (act == 2) o = o > 0.f ? o : o * slope;
        else if (act == 1) o = o > 0.f ? o : 0.f;
        y[o_] = o;
    }
    if (threadIdx.x == 0) {
        mean_out[c] = mean; rstd_out[c] = rstd;
        if (run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (n / (n - 1.f));
        }
    }
}

__global__ __launch_bounds__(256) void batchnorm_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                            float *__restrict__ dx, float *__restrict__ dgamma,
                                                            float *__restrict__ dbeta, const int B, const int C,
                                                            const int HW, const int act, const float slope,
                                                            const int accumulate) {
    __shared__ float red[16];
    const int c = blockIdx.x;
    const float n = (float)B * (float)HW;
    const float mean = mean_in[c], rstd = rstd_in[c], ga = gamma[c], be = beta[c];
    const int BHW = B * HW;
    auto at = [&](const int j) -> size_t { const int b = j / HW; return ((size_t)b * C + c) * HW + (j - b * HW); };
    float s1 = 0.f, s2 = 0.f;
    for (int j = threadIdx.x; j < BHW; j += blockDim.x) {
        const size_t o_ = at(j);
        const float xh = (x[o_] - mean) * rstd;
        const float pre = xh * ga + be;
        float g = dy[o_];
        if (act == 2) g = pre > 0.f ? g : g * slope;
        else if (act == 1) g = pre > 0.f ? g : 0.f;
        s1 += g; s2 += g * xh;
    }
    const float sum1 = block_sum(s1, red), sum2 = block_sum(s2, red);
    const float m1 = sum1 / n, m2 = sum2 / n;
    if (dx) {
        for (int j = threadIdx.x; j < BHW; j += blockDim.x) {
            const size_t o_ = at(j);
            const float xh = (x[o_] - mean) * rstd;
            const float pre = xh * ga + be;
            float g = dy[o_];
            if (act == 2) g = pre > 0.f ? g : g * slope;
            else if (act == 1) g = pre > 0.f ? g : 0.f;
            dx[o_] = ga * rstd * (g - m1 - xh * m2);
        }
    }
    if (threadIdx.x == 0) {
        dgamma[c] = accumulate ? dgamma[c] + sum2 : sum2;
        dbeta[c] = accumulate ? dbeta[c] + sum1 : sum1;
    }
}

// ---- batch 1 (every layer of a batch-1 detector, and the channel-major RoI head [1, C, R*7, 7]): a channel is ONE contiguous
// plane.  One 1024-thread workgroup per channel reads the plane ONCE with float4 loads and keeps it in registers (VPT float4 per
// thread) across the mean / centred-variance / normalise passes, like the instance norm above: 8 B per element forward (the
// three-pass forms move 16), 12 B backward (20).  Same arithmetic and reduction structure as batchnorm_fwd_kernel.
template <int VPT>
__global__ __launch_bounds__(1024) void bn_plane_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            float *__restrict__ run_mean, float *__restrict__ run_var,
                                                            float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                            const int HW, const float eps, const float momentum,
                                                            const int act, const float slope,
                                                            const float *__restrict__ res = nullptr) {
    // res (may be null): the residual join of a ResNet block in the same pass, y = relu(bn(x) + res) -- the same roundings as the
    // normalise kernel followed by add_relu_kernel, without writing and re-reading the normalised map
    __shared__ float red[16];
    const int c = blockIdx.x, n4 = HW >> 2;
    const float n = (float)HW;
    const float4 *x4 = reinterpret_cast<const float4 *>(x + (size_t)c * HW);
    float4 v[VPT];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = j * 1024 + threadIdx.x;
        v[j] = i < n4 ? x4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    const float mean = block_sum(s, red) / n;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        if (j * 1024 + (int)threadIdx.x < n4) {
            const float a = v[j].x - mean, b = v[j].y - mean, cc = v[j].z - mean, d = v[j].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float var = block_sum(q, red) / n;
    const float rstd = 1.f / sqrtf(var + eps);
    const float ga = gamma[c], be = beta[c];
    float4 *y4 = reinterpret_cast<float4 *>(y + (size_t)c * HW);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = j * 1024 + threadIdx.x;
        if (i < n4) {
            float4 o;
            o.x = inorm_act((v[j].x - mean) * rstd * ga + be, act, slope);
            o.y = inorm_act((v[j].y - mean) * rstd * ga + be, act, slope);
            o.z = inorm_act((v[j].z - mean) * rstd * ga + be, act, slope);
            o.w = inorm_act((v[j].w - mean) * rstd * ga + be, act, slope);
            if (res) {
                const float4 r = reinterpret_cast<const float4 *>(res + (size_t)c * HW)[i];
                o.x = fmaxf(o.x + r.x, 0.f); o.y = fmaxf(o.y + r.y, 0.f); o.z = fmaxf(o.z + r.z, 0.f); o.w = fmaxf(o.w + r.w, 0.f);
            }
            y4[i] = o;
        }
    }
    if (threadIdx.x == 0) {
        mean_out[c] = mean; rstd_out[c] = rstd;
        if (run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (n / (n - 1.f));
        }
    }
}

template <int VPT>
__global__ __launch_bounds__(1024) void bn_plane_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                            float *__restrict__ dx, float *__restrict__ dgamma,
                                                            float *__restrict__ dbeta, const int HW, const int act,
                                                            const float slope, const int accumulate,
                                                            const float *__restrict__ join_y = nullptr,
                                                            float *__restrict__ join_g = nullptr) {
    // join_y / join_g (may be null): backward of y = relu(bn(x) + res) in the same pass -- the incoming gradient is gated by
    // join_y > 0 first (act_bwd_kernel's test), the gated gradient is what the residual branch receives (written to join_g) and what
    // the batch norm differentiates
    __shared__ float red[16];
    const int c = blockIdx.x, n4 = HW >> 2;
    const float n = (float)HW;
    const float mean = mean_in[c], rstd = rstd_in[c], ga = gamma[c], be = beta[c];
    const float4 *x4 = reinterpret_cast<const float4 *>(x + (size_t)c * HW), *g4 = reinterpret_cast<const float4 *>(dy + (size_t)c * HW);
    auto gate = [&](float g, float xh) -> float {
        const float pre = xh * ga + be;
        if (act == 2) return pre > 0.f ? g : g * slope;
        if (act == 1) return pre > 0.f ? g : 0.f;
        return g;
    };
    float4 xh[VPT], g[VPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = j * 1024 + threadIdx.x;
        const bool ok = i < n4;
        const float4 xv = ok ? x4[i] : make_float4(mean, mean, mean, mean);
        float4 gv = ok ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (join_y && ok) {
            const float4 yv = reinterpret_cast<const float4 *>(join_y + (size_t)c * HW)[i];
            gv.x = yv.x > 0.f ? gv.x : 0.f; gv.y = yv.y > 0.f ? gv.y : 0.f; gv.z = yv.z > 0.f ? gv.z : 0.f; gv.w = yv.w > 0.f ? gv.w : 0.f;
            reinterpret_cast<float4 *>(join_g + (size_t)c * HW)[i] = gv;
        }
        xh[j].x = (xv.x - mean) * rstd; xh[j].y = (xv.y - mean) * rstd; xh[j].z = (xv.z - mean) * rstd; xh[j].w = (xv.w - mean) * rstd;
        g[j].x = gate(gv.x, xh[j].x); g[j].y = gate(gv.y, xh[j].y); g[j].z = gate(gv.z, xh[j].z); g[j].w = gate(gv.w, xh[j].w);
        s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
        s2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
    }
    const float sum1 = block_sum(s1, red), sum2 = block_sum(s2, red);
    const float m1 = sum1 / n, m2 = sum2 / n;
    if (dx) {
        float4 *d4 = reinterpret_cast<float4 *>(dx + (size_t)c * HW);
        const float k = ga * rstd;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int i = j * 1024 + threadIdx.x;
            if (i < n4) {
                float4 o;
                o.x = k * (g[j].x - m1 - xh[j].x * m2); o.y = k * (g[j].y - m1 - xh[j].y * m2);
                o.z = k * (g[j].z - m1 - xh[j].z * m2); o.w = k * (g[j].w - m1 - xh[j].w * m2);
                d4[i] = o;
            }
        }
    }
    if (threadIdx.x == 0) {
        dgamma[c] = accumulate ? dgamma[c] + sum2 : sum2;
        dbeta[c] = accumulate ? dbeta[c] + sum1 : sum1;
    }
}

// VPT of the plane kernels for a batch-1 layer, or 0 when it does not apply (batch > 1, plane not a multiple of 4 floats or larger
// than 16 float4 per thread, unaligned pointers)
static int bn_plane_vpt(int B, int HW, const void *a, const void *b, const void *c, int max_vpt = 16) {
    if (B != 1 || (HW & 3) || HW > max_vpt * 4096 || getenv("SCDA_BN_NO_PLANE")) return 0;
    if ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) return 0;
    const int need = (HW / 4 + 1023) / 1024;
    return need <= 2 ? 2 : need <= 5 ? 5 : need <= 7 ? 7 : need <= 10 ? 10 : 16;
}

// ---- the same, for maps large enough that one workgroup per channel leaves the chip idle (ResNet layer2/3 at 800 x 1344: 64-256
// channels x 16800-67200 pixels): S slices per channel, statistics through per-slice partials (fixed summation order).
//   fwd: bn_slice_sum -> bn_slice_sq (centred, as the one-kernel form) -> bn_slice_apply     bwd: bn_slice_grad_sums -> bn_slice_dx
__device__ __forceinline__ size_t bn_at(const int j, const int c, const int C, const int HW) {
    const int b = j / HW;
    return ((size_t)b * C + c) * HW + (j - b * HW);
}

__global__ __launch_bounds__(256) void bn_slice_sum_kernel(const float *__restrict__ x, float *__restrict__ part, const int C,
                                                           const int HW, const int BHW, const int S, const int per) {
    __shared__ float red[16];
    const int c = blockIdx.x / S, sl = blockIdx.x % S;
    const int lo = sl * per, hi = min(BHW, lo + per);
    float s = 0.f;
    for (int j = lo + threadIdx.x; j < hi; j += blockDim.x) s += x[bn_at(j, c, C, HW)];
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[(size_t)c * S + sl] = s;
}

__device__ __forceinline__ float bn_sum_parts(const float *__restrict__ part, const int c, const int S) {
    float t = 0.f;
    for (int i = 0; i < S; ++i) t += part[(size_t)c * S + i];
    return t;
}

__global__ __launch_bounds__(256) void bn_slice_sq_kernel(const float *__restrict__ x, const float *__restrict__ part_sum,
                                                          float *__restrict__ part_sq, const int C, const int HW, const int BHW,
                                                          const int S, const int per) {
    __shared__ float red[16];
    const int c = blockIdx.x / S, sl = blockIdx.x % S;
    const float mean = bn_sum_parts(part_sum, c, S) / (float)BHW;
    const int lo = sl * per, hi = min(BHW, lo + per);
    float v = 0.f;
    for (int j = lo + threadIdx.x; j < hi; j += blockDim.x) { const float d = x[bn_at(j, c, C, HW)] - mean; v += d * d; }
    v = block_sum(v, red);
    if (threadIdx.x == 0) part_sq[(size_t)c * S + sl] = v;
}

__global__ __launch_bounds__(256) void bn_slice_apply_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float *__restrict__ run_mean, float *__restrict__ run_var,
                                                             float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                             const float *__restrict__ part_sum, const float *__restrict__ part_sq,
                                                             const int C, const int HW, const int BHW, const int S, const int per,
                                                             const float eps, const float momentum, const int act, const float slope) {
    const int c = blockIdx.x / S, sl = blockIdx.x % S;
    const float n = (float)BHW;
    const float mean = bn_sum_parts(part_sum, c, S) / n, var = bn_sum_parts(part_sq, c, S) / n;
    const float rstd = 1.f / sqrtf(var + eps), ga = gamma[c], be = beta[c];
    const int lo = sl * per, hi = min(BHW, lo + per);
    for (int j = lo + threadIdx.x; j < hi; j += blockDim.x) {
        const size_t o_ = bn_at(j, c, C, HW);
        float o = (x[o_] - mean) * rstd * ga + be;
        if (act == 2) o = o > 0.f ? o : o * slope;
        else if (act == 1) o = o > 0.f ? o : 0.f;
        y[o_] = o;
    }
    if (sl == 0 && threadIdx.x == 0) {
        mean_out[c] = mean; rstd_out[c] = rstd;
        if (run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (n / (n - 1.f));
        }
    }
}

__global__ __launch_bounds__(256) void bn_slice_grad_sums_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                                 float *__restrict__ part, const int C, const int HW, const int BHW,
                                                                 const int S, const int per, const int act, const float slope) {
    __shared__ float red[16];
    const int c = blockIdx.x / S, sl = blockIdx.x % S;
    const float mean = mean_in[c], rstd = rstd_in[c], ga = gamma[c], be = beta[c];
    const int lo = sl * per, hi = min(BHW, lo + per);
    float s1 = 0.f, s2 = 0.f;
    for (int j = lo + threadIdx.x; j < hi; j += blockDim.x) {
        const size_t o_ = bn_at(j, c, C, HW);
        const float xh = (x[o_] - mean) * rstd, pre = xh * ga + be;
        float g = dy[o_];
        if (act == 2) g = pre > 0.f ? g : g * slope;
        else if (act == 1) g = pre > 0.f ? g : 0.f;
        s1 += g; s2 += g * xh;
    }
    s1 = block_sum(s1, red); s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { part[(size_t)c * S + sl] = s1; part[(size_t)(C + c) * S + sl] = s2; }
}

__global__ __launch_bounds__(256) void bn_slice_dx_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                          const float *__restrict__ part, float *__restrict__ dx,
                                                          float *__restrict__ dgamma, float *__restrict__ dbeta, const int C,
                                                          const int HW, const int BHW, const int S, const int per, const int act,
                                                          const float slope, const int accumulate) {
    const int c = blockIdx.x / S, sl = blockIdx.x % S;
    const float n = (float)BHW;
    const float mean = mean_in[c], rstd = rstd_in[c], ga = gamma[c], be = beta[c];
    const float sum1 = bn_sum_parts(part, c, S), sum2 = bn_sum_parts(part, C + c, S);
    const float m1 = sum1 / n, m2 = sum2 / n;
    if (dx) {
        const int lo = sl * per, hi = min(BHW, lo + per);
        for (int j = lo + threadIdx.x; j < hi; j += blockDim.x) {
            const size_t o_ = bn_at(j, c, C, HW);
            const float xh = (x[o_] - mean) * rstd, pre = xh * ga + be;
            float g = dy[o_];
            if (act == 2) g = pre > 0.f ? g : g * slope;
            else if (act == 1) g = pre > 0.f ? g : 0.f;
            dx[o_] = ga * rstd * (g - m1 - xh * m2);
        }
    }
    if (sl == 0 && threadIdx.x == 0) {
        dgamma[c] = accumulate ? dgamma[c] + sum2 : sum2;
        dbeta[c] = accumulate ? dbeta[c] + sum1 : sum1;
    }
}

// slices per channel: enough workgroups for ~4 per CU, at least 4096 elements each
static int bn_slices(int B, int C, int HW) {
    const long long bhw = (long long)B * HW;
    long long s = 1024 / (C > 0 ? C : 1);
    if (s * 4096 > bhw) s = bhw / 4096;
    if (s > 64) s = 64;
    return s < 2 ? 1 : (int)s;
}

// nn.BatchNorm2d in EVAL mode (running statistics; validation of the vgg16_bn detector, dis_patch.eval()): a per-channel
// affine map, HBM-bound (8 B / element).  One float4 per thread where the plane size allows.  The backward (w.r.t. x only:
// statistics and affine parameters are constants in eval mode) is dx = dy * act'(y) * gamma * rstd.
__global__ __launch_bounds__(256) void batchnorm_eval_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             const float *__restrict__ mean, const float *__restrict__ var,
                                                             const long long total, const int C, const int HW,
                                                             const float eps, const int act, const float slope,
                                                             const float *__restrict__ dy_bwd) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)blockDim.x * gridDim.x) {
        const int c = (int)((i / HW) % C);
        const float rstd = 1.f / sqrtf(var[c] + eps);
        float o = (x[i] - mean[c]) * rstd * gamma[c] + beta[c];
        if (dy_bwd) {   // backward: y is recomputed from x to get the activation mask
            float g = dy_bwd[i];
            if (act == 2) g = o > 0.f ? g : g * slope;
            else if (act == 1) g = o > 0.f ? g : 0.f;
            y[i] = g * gamma[c] * rstd;
        } else {
            if (act == 2) o = o > 0.f ? o : o * slope;
            else if (act == 1) o = o > 0.f ? o : 0.f;
            y[i] = o;
        }
    }
}

// ------------------------------------------- bilinear x2, align_corners -----
// Interpolate(scale_factor=2, mode='bilinear', align_corners=True): common_net.py:160-170
// One workgroup = ROWS consecutive output rows of one plane (OH % ROWS == 0), a thread = four consecutive output pixels per step
// (16-byte stores).  (One workgroup per output ROW with a pixel per thread was 65536 half-empty workgroups for the decoders'
// [4, 64, 128, 128] -> 256 x 256 stage: 39-48 us for 84 MB, workgroup dispatch, not memory, set its pace.)
// The two forward kernels that blend (upsample2_fwd_kernel, instnorm_up2_fwd_kernel) must agree bit for bit, and the compiler
// contracts `fy - y0` and the blend differently from one kernel body to the next -- so the roundings are spelled out (they are the
// ones upsample2_fwd_kernel was compiled to before: the fraction as ONE fma of scale and index, clamped; each row pair as
// fma(l0, a, l1 * b); the two rows as fma(ly0, top, ly1 * bottom)).
__device__ __forceinline__ void up2_taps(const float s, const int o, const int In, int &i0, int &i1, float &l0, float &l1) {
    i0 = (int)__fmul_rn(s, (float)o);
    i1 = i0 + (i0 < In - 1 ? 1 : 0);
    l1 = fminf(fmaxf(__fmaf_rn(s, (float)o, -(float)i0), 0.f), 1.f);
    l0 = __fsub_rn(1.f, l1);
}
__device__ __forceinline__ float up2_blend(const float ly0, const float ly1, const float lx0, const float lx1, const float a, const float b,
                                           const float c, const float d) {
    const float top = __fmaf_rn(lx0, a, __fmul_rn(lx1, b)), bot = __fmaf_rn(lx0, c, __fmul_rn(lx1, d));
    return __fmaf_rn(ly0, top, __fmul_rn(ly1, bot));
}

// the weight with which output row / column o enters input row / column i (0: it does not), and the gather's accumulation step --
// spelled out for the same reason: upsample2_bwd_kernel, upsample2_bwd_rows_kernel and instnorm_up2_bwd_kernel agree bit for bit
__device__ __forceinline__ float up2_weight(const float s, const int o, const int i, const int In) {
    int i0, i1;
    float l0, l1;
    up2_taps(s, o, In, i0, i1, l0, l1);
    return __fadd_rn(i0 == i ? l0 : 0.f, i1 == i ? l1 : 0.f);
}

template <int ROWS>
__global__ __launch_bounds__(256) void upsample2_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                            const int IH, const int IW, const int OH, const int OW,
                                                            const float sh, const float sw) {
    const int row0 = blockIdx.x * ROWS, pc = row0 / OH, oy0 = row0 - pc * OH;       // workgroup-uniform
    const int q_per_row = (OW + 3) >> 2;
    const float *plane = x + (size_t)pc * IH * IW;
    for (int i = threadIdx.x; i < ROWS * q_per_row; i += 256) {
        const int r = i / q_per_row, q = i - r * q_per_row;
        const int oy = oy0 + r;
        int y0, y1;
        float ly0, ly1;
        up2_taps(sh, oy, IH, y0, y1, ly0, ly1);
        const float *p0 = plane + (size_t)y0 * IW, *p1 = plane + (size_t)y1 * IW;
        float *out = y + ((size_t)row0 + r) * OW;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int x0, x1;
            float lx0, lx1;
            up2_taps(sw, min(q * 4 + k, OW - 1), IW, x0, x1, lx0, lx1);
            v[k] = up2_blend(ly0, ly1, lx0, lx1, p0[x0], p0[x1], p1[x0], p1[x1]);
        }
        if (q * 4 + 3 < OW && (OW & 3) == 0) {
            *reinterpret_cast<float4 *>(out + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (q * 4 + k < OW) out[q * 4 + k] = v[k];
        }
    }
}

// Instance norm (+ fused activation, or + the residual block's dropout-and-add tail) AND the bilinear x2 behind it in one launch
// (round 6; common_net.py:59-80 -> :279-293: in both decoders every Interpolate reads the output of an instance norm and nothing
// else does): the norm's workgroup holds the whole plane, so the normalised plane goes to LDS instead of HBM and the same 1024
// threads write the 2H x 2W map from there -- a lane owns output columns (its four taps' columns and weights are computed once), a
// wave a set of output rows; stores are 256-byte runs.  The small plane is never written (its backward needs x, mean, rstd only)
// and never re-read; values are those of instnorm_fwd_kernel followed by upsample2_fwd_kernel, bit for bit (up2_taps / up2_blend).
template <int VPT>
__global__ __launch_bounds__(1024) void instnorm_up2_fwd_kernel(const float *__restrict__ x, float *__restrict__ y2,
                                                                float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                                const int IH, const int IW, const float eps, const int act,
                                                                const float slope, const DropTail dt, const float sh, const float sw) {
    constexpr int HW = VPT * 4096;
    __shared__ float red[16];
    __shared__ __attribute__((aligned(16))) float plane[HW];
    const size_t base = (size_t)blockIdx.x * HW;
    const unsigned long long dseed = dt.seed_ptr ? *dt.seed_ptr : dt.seed;
    auto tail = [&](float o, const size_t i) -> float {      // as in instnorm_fwd_kernel
        if (!dt.residual) return o;
        const float d = mix_hash(dseed, (uint64_t)i) >= dt.thr ? o * dt.scale : 0.f;
        return d + dt.residual[i];
    };
    {
        const float4 *x4 = reinterpret_cast<const float4 *>(x + base);
        float4 v[VPT];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            v[j] = x4[j * 1024 + threadIdx.x];
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        const float mean = block_sum(s, red) / (float)HW;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
        const float rstd = 1.f / sqrtf(block_sum(q, red) / (float)HW + eps);
        float4 *p4 = reinterpret_cast<float4 *>(plane);
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            float4 o;
            const size_t i0 = base + 4 * (size_t)(j * 1024 + threadIdx.x);
            o.x = tail(inorm_act((v[j].x - mean) * rstd, act, slope), i0);
            o.y = tail(inorm_act((v[j].y - mean) * rstd, act, slope), i0 + 1);
            o.z = tail(inorm_act((v[j].z - mean) * rstd, act, slope), i0 + 2);
            o.w = tail(inorm_act((v[j].w - mean) * rstd, act, slope), i0 + 3);
            p4[j * 1024 + threadIdx.x] = o;
        }
        if (threadIdx.x == 0) { mean_out[blockIdx.x] = mean; rstd_out[blockIdx.x] = rstd; }
    }
    __syncthreads();
    const int OH = 2 * IH, OW = 2 * IW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *out = y2 + (size_t)blockIdx.x * 4 * HW;
    for (int c0 = 0; c0 < OW; c0 += 256) {        // (one round for the decoders' maps: OW = 128 / 256)
        int x0[4], x1[4];
        float lx0[4], lx1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) up2_taps(sw, min(c0 + k * 64 + lane, OW - 1), IW, x0[k], x1[k], lx0[k], lx1[k]);
        for (int oy = wave; oy < OH; oy += 16) {
            int y0, y1;
            float ly0, ly1;
            up2_taps(sh, oy, IH, y0, y1, ly0, ly1);
            const float *p0 = plane + y0 * IW, *p1 = plane + y1 * IW;
            float *orow = out + (size_t)oy * OW + c0 + lane;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c0 + k * 64 + lane < OW) orow[k * 64] = up2_blend(ly0, ly1, lx0[k], lx1[k], p0[x0[k]], p0[x1[k]], p1[x0[k]], p1[x1[k]]);
        }
    }
}

// gather-form backward: input pixel (iy,ix) collects from the output rows/cols whose
// (y0|y1) equals iy -- deterministic, no atomics.
__device__ __forceinline__ void up2_candidates(int i, int In, int On, float s, int &lo, int &hi) {
    // outputs o with floor(s*o) in {i-1, i}:  o in [ceil((i-1)/s), floor((i+1)/s)] (clamped), checked exactly by the caller
    lo = (int)floorf((float)(i - 1) / s) - 1;
    hi = (int)ceilf((float)(i + 1) / s) + 1;
    if (lo < 0) lo = 0;
    if (hi > On - 1) hi = On - 1;
    (void)In;
}

// Separable gather: a workgroup owns one input row.  Pass A adds the (4-5) output rows that touch it, weighted, into an LDS
// row of OW floats (coalesced reads of dy); pass B gives every input column its (4-5) weighted entries of that row.
// 10 coalesced loads per result instead of 25-56 predicated ones; deterministic (fixed order, no atomics).
constexpr int kUpMaxOW = 4096;

template <int ROWS>
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx,
                                                            const int IH, const int IW, const int OH, const int OW,
                                                            const float sh, const float sw) {
    __shared__ float rowbuf[kUpMaxOW];
    // ROWS consecutive input rows of one plane per workgroup, one after the other (IH % ROWS == 0): a workgroup per input row was
    // 32768 workgroups of 256 threads for 128 results each
    for (int rr = 0; rr < ROWS; ++rr) {
    const int row = blockIdx.x * ROWS + rr, iy = row % IH, pc = row / IH;       // workgroup-uniform
    int ylo, yhi;
    up2_candidates(iy, IH, OH, sh, ylo, yhi);
    const float *g = dy + (size_t)pc * OH * OW;
    if (rr) __syncthreads();
    for (int ox = threadIdx.x; ox < OW; ox += blockDim.x) {
        float acc = 0.f;
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float wy = up2_weight(sh, oy, iy, IH);
            if (wy != 0.f) acc = __fmaf_rn(wy, g[(size_t)oy * OW + ox], acc);
        }
        rowbuf[ox] = acc;
    }
    __syncthreads();
    for (int ix = threadIdx.x; ix < IW; ix += blockDim.x) {
        int xlo, xhi;
        up2_candidates(ix, IW, OW, sw, xlo, xhi);
        float acc = 0.f;
        for (int ox = xlo; ox <= xhi; ++ox) {
            const float wx = up2_weight(sw, ox, ix, IW);
            if (wx != 0.f) acc = __fmaf_rn(wx, rowbuf[ox], acc);
        }
        dx[(size_t)row * IW + ix] = acc;
    }
    }
}

// The same gather with ROWS input rows of a plane in flight at once (IH % ROWS == 0, OW % 4 == 0, OW <= kUpRowsOW): pass A fills an
// LDS block [ROWS][OW] -- a thread owns four consecutive columns of one row and reads the 4 - 5 output rows that touch it with 16-byte
// loads -- ONE barrier, pass B gives every (row, column) of the ROWS x IW results its 4 - 5 weighted entries.  The row-at-a-time kernel
// above is two barriers and two half-empty passes (OW and IW of 256 threads busy) per input row, four rows one after the other per
// workgroup: 52 - 70 us for the decoders' [4, 128, 128, 128] / [4, 64, 256, 256] gradients (0.8 - 1.2 TB/s), a latency chain.  Per
// element the same weights added in the same order (output rows, then output columns, ascending): bit-identical results.
constexpr int kUpRowsOW = 1024;

template <int ROWS>
__global__ __launch_bounds__(256) void upsample2_bwd_rows_kernel(const float *__restrict__ dy, float *__restrict__ dx,
                                                                 const int IH, const int IW, const int OH, const int OW,
                                                                 const float sh, const float sw) {
    __shared__ __attribute__((aligned(16))) float rowbuf[ROWS * kUpRowsOW];
    const int row0 = blockIdx.x * ROWS, pc = row0 / IH, iy0 = row0 - pc * IH;       // workgroup-uniform: the ROWS rows share a plane
    const float *g = dy + (size_t)pc * OH * OW;
    const int q_per_row = OW >> 2;
    for (int i = threadIdx.x; i < ROWS * q_per_row; i += 256) {
        const int rr = i / q_per_row, q = i - rr * q_per_row;
        const int iy = iy0 + rr;
        int ylo, yhi;
        up2_candidates(iy, IH, OH, sh, ylo, yhi);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float wy = up2_weight(sh, oy, iy, IH);
            if (wy != 0.f) {
                const float4 v = *reinterpret_cast<const float4 *>(g + (size_t)oy * OW + q * 4);
                acc.x = __fmaf_rn(wy, v.x, acc.x); acc.y = __fmaf_rn(wy, v.y, acc.y); acc.z = __fmaf_rn(wy, v.z, acc.z); acc.w = __fmaf_rn(wy, v.w, acc.w);
            }
        }
        *reinterpret_cast<float4 *>(rowbuf + rr * OW + q * 4) = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ROWS * IW; i += 256) {
        const int rr = i / IW, ix = i - rr * IW;
        int xlo, xhi;
        up2_candidates(ix, IW, OW, sw, xlo, xhi);
        const float *rb = rowbuf + rr * OW;
        float acc = 0.f;
        for (int ox = xlo; ox <= xhi; ++ox) {
            const float wx = up2_weight(sw, ox, ix, IW);
            if (wx != 0.f) acc = __fmaf_rn(wx, rb[ox], acc);
        }
        dx[(size_t)(row0 + rr) * IW + ix] = acc;
    }
}

// The backward of instnorm_up2_fwd_kernel in one launch: the bilinear gather of the 2H x 2W gradient (pass A: output rows into an
// LDS block [IH][2 IW], as upsample2_bwd_rows_kernel for the whole plane; pass B: every thread gathers the four consecutive columns
// of each of ITS float4 slots) hands the small plane's gradient to the norm's backward in registers (instnorm_bwd_plane); with
// `dsmall` it is also written out (the residual input's gradient of the dropout-and-add tail).  Same values as the two launches.
template <int VPT>
__global__ __launch_bounds__(1024) void instnorm_up2_bwd_kernel(const float *__restrict__ dy2, const float *__restrict__ x,
                                                                const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                                float *__restrict__ dx, float *__restrict__ dsmall, const int IH,
                                                                const int IW, const int act, const float slope, const DropTail dt,
                                                                const float sh, const float sw) {
    constexpr int HW = VPT * 4096;
    __shared__ float red[16];
    __shared__ __attribute__((aligned(16))) float inter[2 * HW];        // [IH][OW]
    const int OH = 2 * IH, OW = 2 * IW, q_per_row = OW >> 2;
    const size_t base = (size_t)blockIdx.x * HW;
    const float *g = dy2 + base * 4;
    for (int i = threadIdx.x; i < IH * q_per_row; i += 1024) {
        const int iy = i / q_per_row, q = i - iy * q_per_row;
        int ylo, yhi;
        up2_candidates(iy, IH, OH, sh, ylo, yhi);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float wy = up2_weight(sh, oy, iy, IH);
            if (wy != 0.f) {
                const float4 v = *reinterpret_cast<const float4 *>(g + (size_t)oy * OW + q * 4);
                acc.x = __fmaf_rn(wy, v.x, acc.x); acc.y = __fmaf_rn(wy, v.y, acc.y); acc.z = __fmaf_rn(wy, v.z, acc.z); acc.w = __fmaf_rn(wy, v.w, acc.w);
            }
        }
        *reinterpret_cast<float4 *>(inter + iy * OW + q * 4) = acc;
    }
    __syncthreads();
    float4 gs[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int idx = 4 * (j * 1024 + (int)threadIdx.x), iy = idx / IW, ix0 = idx - iy * IW;
        const float *rb = inter + iy * OW;
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int xlo, xhi;
            up2_candidates(ix0 + k, IW, OW, sw, xlo, xhi);
            float acc = 0.f;
            for (int ox = xlo; ox <= xhi; ++ox) {
                const float wx = up2_weight(sw, ox, ix0 + k, IW);
                if (wx != 0.f) acc = __fmaf_rn(wx, rb[ox], acc);
            }
            r[k] = acc;
        }
        gs[j] = make_float4(r[0], r[1], r[2], r[3]);
        if (dsmall) reinterpret_cast<float4 *>(dsmall + base)[j * 1024 + threadIdx.x] = gs[j];
    }
    const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];
    const unsigned long long dseed = dt.seed_ptr ? *dt.seed_ptr : dt.seed;
    instnorm_bwd_plane<VPT, true>([&](const int j) { return gs[j]; }, x, mean, rstd, dx, HW, act, slope, dt, dseed, base, red);
}

// ------------------------------------------------------------- BCE ----------
// F.binary_cross_entropy(p, t) mean with the log clamp at -100 (torch semantics);
// faster_rcnn_train_val.py:584-600,627-628,675-687,723-732.  single workgroup (n <= ~1e5).
__global__ __launch_bounds__(1024) void bce_fwd_kernel(const float *__restrict__ p, const float *__restrict__ t,
                                                       const int n, float *__restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float pi = p[i], ti = t[i];
        const float lp = fmaxf(logf(pi), -100.f), l1p = fmaxf(logf(1.f - pi), -100.f);
        s += -(ti * lp + (1.f - ti) * l1p);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s / (float)n;
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float *__restrict__ p, const float *__restrict__ t, const int n,
                                                      const float *__restrict__ g, float *__restrict__ dp) {
    const float gs = g[0] / (float)n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        const float pi = p[i];
        dp[i] = (pi - t[i]) / fmaxf((1.f - pi) * pi, 1e-12f) * gs;
    }
}

// ------------------------------------------------- global average pool ------
// nn.AvgPool2d(full extent) : common_net.py:239.  y[bc] = mean_p x[bc][p]
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, const int HW) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) s += x[(size_t)blockIdx.x * HW + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) y[blockIdx.x] = s / (float)HW;
}

// small planes (7 x 7 RoI maps x 2048 channels x 512 RoIs = 1 M planes): a workgroup takes 256 consecutive planes, streams their
// (contiguous) elements through LDS with coalesced loads, then one thread sums one plane (stride HW in LDS: conflict-free for odd HW)
__global__ __launch_bounds__(256) void gap_fwd_small_kernel(const float *__restrict__ x, float *__restrict__ y, const long long planes,
                                                            const int HW) {
    extern __shared__ float tile[];   // [256 * HW]
    for (long long p0 = (long long)blockIdx.x * 256; p0 < planes; p0 += (long long)gridDim.x * 256) {
        const int np = (int)min((long long)256, planes - p0), n = np * HW;
        const float *src = x + p0 * HW;
        for (int i = threadIdx.x; i < n; i += 256) tile[i] = src[i];
        __syncthreads();
        if ((int)threadIdx.x < np) {
            float s = 0.f;
            for (int i = 0; i < HW; ++i) s += tile[threadIdx.x * HW + i];
            y[p0 + threadIdx.x] = s / (float)HW;
        }
        __syncthreads();
    }
}

// 2x2 stride-1 average over an (H+1) x (W+1) map -> H x W: the pooling half of RoIAlignAvg
// (extensions/_roi_align/modules/roi_align.py:18-30: align to (h+1, w+1), then F.avg_pool2d(kernel_size=2, stride=1))
__global__ __launch_bounds__(256) void avg2x2s1_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, const long long total,
                                                           const int H, const int W) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)blockDim.x * gridDim.x) {
        const int w = (int)(i % W);
        const long long r = i / W;
        const int h = (int)(r % H);
        const float *p = x + ((r / H) * (H + 1) + h) * (W + 1) + w;
        y[i] = (p[0] + p[1] + p[W + 1] + p[W + 2]) * 0.25f;
    }
}

__global__ __launch_bounds__(256) void avg2x2s1_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, const long long total,
                                                           const int H, const int W) {   // total counts the (H+1) x (W+1) inputs
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)blockDim.x * gridDim.x) {
        const int w = (int)(i % (W + 1));
        const long long r = i / (W + 1);
        const int h = (int)(r % (H + 1));
        const float *p = dy + (r / (H + 1)) * H * W;
        float s = 0.f;
        if (h < H && w < W) s += p[h * W + w];
        if (h < H && w > 0) s += p[h * W + w - 1];
        if (h > 0 && w < W) s += p[(h - 1) * W + w];
        if (h > 0 && w > 0) s += p[(h - 1) * W + w - 1];
        dx[i] = s * 0.25f;
    }
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx,
                                                      const long long total, const int HW) {
    const float inv = 1.f / (float)HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)blockDim.x * gridDim.x)
        dx[i] = dy[i / HW] * inv;
}

// mean over the last dim: y[r] = mean_c x[r][c]   (x_target_patch_pro_mean, faster_rcnn_train_val.py:591)
__global__ __launch_bounds__(256) void row_mean_kernel(const float *__restrict__ x, float *__restrict__ y, const int C) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < C; i += blockDim.x) s += x[(size_t)blockIdx.x * C + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) y[blockIdx.x] = s / (float)C;
}

// ------------------------------------------------------------- Adam ---------
// torch.optim.Adam(betas, eps=1e-8, weight_decay) with L2-coupled decay (faster_rcnn_train_val.py:305-316):
//   g += wd*p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// one pass over a flat parameter bucket: 4 reads + 3 writes of 4 B per parameter.
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, const long long n, const float lr,
                                                   const float b1, const float b2, const float eps, const float wd,
                                                   const float bc1, const float bc2_sqrt) {
    const float step = lr / bc1;
    const long long n4 = n / 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)blockDim.x * gridDim.x) {
        float4 pp = reinterpret_cast<float4 *>(p)[i];
        const float4 gg = reinterpret_cast<const float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
#define ADAM1(P, G, M, V)                                   \
    {                                                       \
        const float gr = G + wd * P;                        \
        M = b1 * M + (1.f - b1) * gr;                       \
        V = b2 * V + (1.f - b2) * gr * gr;                  \
        P = P - step * (M / (sqrtf(V) / bc2_sqrt + eps));   \
    }
        ADAM1(pp.x, gg.x, mm.x, vv.x) ADAM1(pp.y, gg.y, mm.y, vv.y) ADAM1(pp.z, gg.z, mm.z, vv.z) ADAM1(pp.w, gg.w, mm.w, vv.w)
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
    }
    for (long long i = n4 * 4 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)blockDim.x * gridDim.x) {
        float P = p[i], M = m[i], V = v[i];
        const float G = g[i];
        ADAM1(P, G, M, V)
        p[i] = P; m[i] = M; v[i] = V;
    }
#undef ADAM1
}

}  // namespace scda

using namespace scda;

#define NN_CHECK(cond, name)                                      \
    if (!(cond)) { set_error(name ": bad arguments"); return SCDA_EINVAL; }

SCDA_API int scda_maxpool2x2_fwd_hip(const float *x, float *y, uint8_t *idx, int planes, int H, int W, void *stream) {
    NN_CHECK(x && y && idx && planes > 0 && H >= 2 && W >= 2, "scda_maxpool2x2_fwd_hip")
    const int OH = H / 2, OW = W / 2;
    const long long total = (long long)planes * OH * OW;
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_grid(total) * 4), dim3(256), 0, as_stream(stream), x, y, idx, total, H, W, OH, OW);
    return launch_status("maxpool2_fwd_kernel");
}

SCDA_API int scda_maxpool2x2_bwd_hip(const float *dy, const uint8_t *idx, float *dx, int planes, int H, int W,
                                     void *stream) {
    NN_CHECK(dy && dx && idx && planes > 0 && H >= 2 && W >= 2, "scda_maxpool2x2_bwd_hip")
    const int OH = H / 2, OW = W / 2;
    const long long total = (long long)planes * OH * OW;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid(total) * 4), dim3(256), 0, as_stream(stream), dy, idx, dx, total, H, W, OH, OW,
                       (const float *)nullptr);
    return launch_status("maxpool2_bwd_kernel");
}

SCDA_API int scda_maxpool2x2_bwd_relu_hip(const float *dy, const uint8_t *idx, const float *y_pooled, float *dx, int planes, int H,
                                          int W, void *stream) {
    NN_CHECK(dy && dx && idx && y_pooled && planes > 0 && H >= 2 && W >= 2, "scda_maxpool2x2_bwd_relu_hip")
    const int OH = H / 2, OW = W / 2;
    const long long total = (long long)planes * OH * OW;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid(total) * 4), dim3(256), 0, as_stream(stream), dy, idx, dx, total, H, W, OH, OW,
                       y_pooled);
    return launch_status("maxpool2_bwd_kernel");
}

SCDA_API int scda_maxpool3x3s2_fwd_hip(const float *x, float *y, int planes, int H, int W, void *stream) {
    NN_CHECK(x && y && planes > 0 && H >= 1 && W >= 1, "scda_maxpool3x3s2_fwd_hip")
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)planes * OH * OW;
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(ew_grid(total) * 4), dim3(256), 0, as_stream(stream), x, y, total, H, W, OH, OW);
    return launch_status("maxpool3s2_fwd_kernel");
}

SCDA_API int scda_add_relu_hip(const float *a, const float *b, float *y, long long n, void *stream) {
    NN_CHECK(a && b && y && n >= 0, "scda_add_relu_hip")
    if (n == 0) return SCDA_OK;
    hipLaunchKernelGGL(add_relu_kernel, dim3(ew_grid(n) * 4), dim3(256), 0, as_stream(stream), a, b, y, n);
    return launch_status("add_relu_kernel");
}

SCDA_API int scda_act_fwd_hip(const float *x, float *y, long long n, int mode, float slope, void *stream) {
    NN_CHECK(x && y && n >= 0 && mode >= 0 && mode <= 3, "scda_act_fwd_hip")
    if (n == 0) return SCDA_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n) * 4), dim3(256), 0, as_stream(stream), x, y, n, mode, slope);
    return launch_status("act_fwd_kernel");
}

SCDA_API int scda_act_bwd_hip(const float *dy, const float *y, float *dx, long long n, int mode, float slope,
                              void *stream) {
    NN_CHECK(dy && y && dx && n >= 0 && mode >= 0 && mode <= 3, "scda_act_bwd_hip")
    if (n == 0) return SCDA_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n) * 4), dim3(256), 0, as_stream(stream), dy, y, dx, n, mode, slope);
    return launch_status("act_bwd_kernel");
}

SCDA_API int scda_axpby_hip(const float *a, const float *b, float *y, long long n, float alpha, float beta, void *stream) {
    NN_CHECK(a && y && n >= 0, "scda_axpby_hip")
    if (n == 0) return SCDA_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n) * 4), dim3(256), 0, as_stream(stream), a, b, y, n, alpha, beta);
    return launch_status("axpby_kernel");
}

SCDA_API int scda_dropout_mask_hip(uint8_t *mask, long long n, float p, uint64_t seed, void *stream) {
    NN_CHECK(mask && n >= 0 && p >= 0.f && p < 1.f, "scda_dropout_mask_hip")
    if (n == 0) return SCDA_OK;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_grid(n) * 4), dim3(256), 0, as_stream(stream), mask, n, p, seed);
    return launch_status("dropout_mask_kernel");
}

SCDA_API int scda_dropout_apply_hip(const float *x, const uint8_t *mask, float *y, long long n, float scale, void *stream) {
    NN_CHECK(x && mask && y && n >= 0, "scda_dropout_apply_hip")
    if (n == 0) return SCDA_OK;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(ew_grid(n) * 4), dim3(256), 0, as_stream(stream), x, mask, y, n, scale);
    return launch_status("dropout_apply_kernel");
}

#define BIAS_GRAD_MAX_SPLIT 128
SCDA_API int scda_dropout_seeded_hip(const float *x, float *y, long long n, float p, uint64_t seed, float scale,
                                     const float *relu_src, void *stream) {
    NN_CHECK(x && y && n >= 0 && p >= 0.f && p < 1.f, "scda_dropout_seeded_hip")
    if (n == 0) return SCDA_OK;
    hipLaunchKernelGGL(dropout_seeded_kernel, dim3(ew_grid(n) * 4), dim3(256), 0, as_stream(stream), x, y, n, p, seed, scale, relu_src);
    return launch_status("dropout_seeded_kernel");
}

SCDA_API int scda_sigmoid_bce_rows_fwd_hip(const float *x, const float *t, int t_rows, const float *w, int C, int n, float scale,
                                           int accumulate, float *prob_out, float *out1, void *stream) {
    NN_CHECK(x && t && out1 && C > 0 && n > 0 && (t_rows == 1 || t_rows == C), "scda_sigmoid_bce_rows_fwd_hip")
    hipLaunchKernelGGL(sigmoid_bce_rows_fwd_kernel, dim3(1), dim3(1024), 0, as_stream(stream), x, t, t_rows, w, C, n, scale, accumulate,
                       prob_out, out1);
    return launch_status("sigmoid_bce_rows_fwd_kernel");
}

SCDA_API int scda_sigmoid_bce_rows_bwd_hip(const float *prob, const float *t, int t_rows, const float *w, int C, int n, float scale,
                                           const float *grad_scalar, float *dx, void *stream) {
    NN_CHECK(prob && t && grad_scalar && dx && C > 0 && n > 0 && (t_rows == 1 || t_rows == C), "scda_sigmoid_bce_rows_bwd_hip")
    hipLaunchKernelGGL(sigmoid_bce_rows_bwd_kernel, dim3(ew_grid((long long)C * n)), dim3(256), 0, as_stream(stream), prob, t, t_rows, w,
                       C, n, scale, grad_scalar, dx);
    return launch_status("sigmoid_bce_rows_bwd_kernel");
}

SCDA_API size_t scda_bias_grad_workspace_bytes(int C) { return (size_t)C * BIAS_GRAD_MAX_SPLIT * sizeof(float); }

SCDA_API int scda_bias_grad_nchw_hip(const float *dy, float *db, int B, int C, int HW, int accumulate, float *ws,
                                     void *stream) {
    NN_CHECK(dy && db && ws && B > 0 && C > 0 && HW > 0, "scda_bias_grad_nchw_hip")
    // ~2048 workgroups in total, at least 2048 pixels per slice
    int nsplit = 2048 / C;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > BIAS_GRAD_MAX_SPLIT) nsplit = BIAS_GRAD_MAX_SPLIT;
    int slice = (HW + nsplit - 1) / nsplit;
    if (slice < 2048) slice = 2048;
    slice = (slice + 3) & ~3;
    nsplit = (HW + slice - 1) / slice;
    hipLaunchKernelGGL(bias_grad_partial_kernel, dim3(C, nsplit), dim3(256), 0, as_stream(stream), dy, ws, B, C, HW, slice);
    int rc = launch_status("bias_grad_partial_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), (const float *)ws, db, C, nsplit, accumulate);
    return launch_status("bias_grad_finish_kernel");
}

SCDA_API int scda_colsum_hip(const float *dy, float *db, int M, int N, int accumulate, void *stream) {
    NN_CHECK(dy && db && M > 0 && N > 0, "scda_colsum_hip")
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 32)), dim3(256), 0, as_stream(stream), dy, db, M, N, accumulate);
    return launch_status("colsum_kernel");
}

SCDA_API int scda_softmax_ce_fwd_hip(const float *logits, const int64_t *targets, int R, int C, int ignore_index,
                                     float *probs, float *out2, void *stream) {
    NN_CHECK(logits && targets && probs && out2 && R > 0 && C > 0, "scda_softmax_ce_fwd_hip")
    hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3(1), dim3(1024), 0, as_stream(stream), logits, targets, R, C, ignore_index, probs, out2);
    return launch_status("softmax_ce_fwd_kernel");
}

SCDA_API int scda_softmax_ce_bwd_hip(const float *probs, const int64_t *targets, int R, int C, int ignore_index,
                                     const float *fwd_out2, const float *grad_scalar, float *dlogits, void *stream) {
    NN_CHECK(probs && targets && fwd_out2 && grad_scalar && dlogits && R > 0 && C > 0, "scda_softmax_ce_bwd_hip")
    hipLaunchKernelGGL(softmax_ce_bwd_kernel, dim3(ew_grid((long long)R * C)), dim3(256), 0, as_stream(stream), probs, targets, R, C,
                       ignore_index, fwd_out2, grad_scalar, dlogits);
    return launch_status("softmax_ce_bwd_kernel");
}

SCDA_API int scda_row_softmax_hip(const float *x, float *y, int R, int C, void *stream) {
    NN_CHECK(x && y && R > 0 && C > 0, "scda_row_softmax_hip")
    hipLaunchKernelGGL(row_softmax_kernel, dim3(ew_grid(R)), dim3(256), 0, as_stream(stream), x, y, R, C);
    return launch_status("row_softmax_kernel");
}

SCDA_API int scda_accuracy_hip(const float *logits, const int64_t *targets, int R, int C, int ignore_index, float *out1,
                               void *stream) {
    NN_CHECK(logits && targets && out1 && R > 0 && C > 0, "scda_accuracy_hip")
    hipLaunchKernelGGL(accuracy_kernel, dim3(1), dim3(1024), 0, as_stream(stream), logits, targets, R, C, ignore_index, out1);
    return launch_status("accuracy_kernel");
}

#define SL1_BLOCKS 512
SCDA_API size_t scda_smooth_l1_workspace_bytes(void) { return SL1_BLOCKS * sizeof(float); }

SCDA_API int scda_smooth_l1_fwd_hip(const float *pred, const float *mask, const float *target, long long n, float sigma,
                                    float scale, float *partial_ws, float *out1, void *stream) {
    NN_CHECK(pred && target && partial_ws && out1 && n > 0, "scda_smooth_l1_fwd_hip")
    int blocks = ew_grid(n);
    if (blocks > SL1_BLOCKS) blocks = SL1_BLOCKS;
    hipLaunchKernelGGL(smooth_l1_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), pred, mask, target, n, sigma * sigma, partial_ws);
    int rc = launch_status("smooth_l1_fwd_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const float *)partial_ws, blocks, scale, out1);
    return launch_status("finish_sum_kernel");
}

SCDA_API int scda_smooth_l1_bwd_hip(const float *pred, const float *mask, const float *target, long long n, float sigma,
                                    float scale, const float *grad_scalar, float *dpred, void *stream) {
    NN_CHECK(pred && target && grad_scalar && dpred && n > 0, "scda_smooth_l1_bwd_hip")
    hipLaunchKernelGGL(smooth_l1_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), pred, mask, target, n, sigma * sigma, scale,
                       grad_scalar, dpred);
    return launch_status("smooth_l1_bwd_kernel");
}

static unsigned drop_threshold(float p) {
    const double t = (double)p * 4294967296.0;
    return (unsigned)(t > 4294967295.0 ? 4294967295.0 : t);
}

static int instnorm_fwd_launch(const float *x, float *y, float *mean, float *rstd, int planes, int HW, float eps, int act, float slope,
                               const DropTail dt, void *stream) {
    const bool al = ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)dt.residual)) & 15) == 0;
#define INORM_FWD(V) hipLaunchKernelGGL(instnorm_fwd_kernel<V>, dim3(planes), dim3(1024), 0, as_stream(stream), x, y, mean, rstd, HW, eps, act, slope, dt)
    if (al && HW == 4096) INORM_FWD(1);
    else if (al && HW == 16384) INORM_FWD(4);
    else if (al && HW == 65536) INORM_FWD(16);
    else INORM_FWD(0);
#undef INORM_FWD
    return launch_status("instnorm_fwd_kernel");
}

static int instnorm_bwd_launch(const float *dy, const float *x, const float *mean, const float *rstd, float *dx, int planes, int HW,
                               int act, float slope, const DropTail dt, void *stream) {
    const bool al = ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0;
#define INORM_BWD(V, CX) hipLaunchKernelGGL((instnorm_bwd_kernel<V, CX>), dim3(planes), dim3(1024), 0, as_stream(stream), dy, x, mean, rstd, dx, HW, act, slope, dt)
    if (al && HW == 4096) INORM_BWD(1, true);
    else if (al && HW == 16384) INORM_BWD(4, true);
    else if (al && HW == 65536) INORM_BWD(16, false);
    else INORM_BWD(0, false);
#undef INORM_BWD
    return launch_status("instnorm_bwd_kernel");
}

SCDA_API int scda_instnorm_fwd_hip(const float *x, float *y, float *mean, float *rstd, int planes, int HW, float eps,
                                   int act, float slope, void *stream) {
    NN_CHECK(x && y && mean && rstd && planes > 0 && HW > 0, "scda_instnorm_fwd_hip")
    return instnorm_fwd_launch(x, y, mean, rstd, planes, HW, eps, act, slope, DropTail{nullptr, 0u, 0ull, 1.f, 0, nullptr}, stream);
}

SCDA_API int scda_instnorm_bwd_hip(const float *dy, const float *x, const float *mean, const float *rstd, float *dx,
                                   int planes, int HW, int act, float slope, void *stream) {
    NN_CHECK(dy && x && mean && rstd && dx && planes > 0 && HW > 0, "scda_instnorm_bwd_hip")
    return instnorm_bwd_launch(dy, x, mean, rstd, dx, planes, HW, act, slope, DropTail{nullptr, 0u, 0ull, 1.f, 0, nullptr}, stream);
}

// y = residual + dropout_p,seed(instance_norm(x)): the tail of a residual block in one launch (see DropTail)
SCDA_API int scda_instnorm_drop_add_fwd_hip(const float *x, const float *residual, float *y, float *mean, float *rstd, int planes,
                                            int HW, float eps, float p, uint64_t seed, float scale, void *stream) {
    NN_CHECK(x && residual && y && mean && rstd && planes > 0 && HW > 0 && p >= 0.f && p < 1.f, "scda_instnorm_drop_add_fwd_hip")
    return instnorm_fwd_launch(x, y, mean, rstd, planes, HW, eps, 0, 0.f, DropTail{residual, drop_threshold(p), seed, scale, 1, nullptr}, stream);
}

// ... with the seed in DEVICE memory (seed_dev: one uint64): the form a hipGraph records
SCDA_API int scda_instnorm_drop_add_fwd_dev_hip(const float *x, const float *residual, float *y, float *mean, float *rstd, int planes,
                                                int HW, float eps, float p, const uint64_t *seed_dev, float scale, void *stream) {
    NN_CHECK(x && residual && y && mean && rstd && seed_dev && planes > 0 && HW > 0 && p >= 0.f && p < 1.f, "scda_instnorm_drop_add_fwd_dev_hip")
    return instnorm_fwd_launch(x, y, mean, rstd, planes, HW, eps, 0, 0.f,
                               DropTail{residual, drop_threshold(p), 0ull, scale, 1, (const unsigned long long *)seed_dev}, stream);
}

// dx of the same: the dropout's mask (recomputed from the seed) applied to dy, then the norm's gradient
SCDA_API int scda_instnorm_drop_bwd_hip(const float *dy, const float *x, const float *mean, const float *rstd, float *dx, int planes,
                                        int HW, float p, uint64_t seed, float scale, void *stream) {
    NN_CHECK(dy && x && mean && rstd && dx && planes > 0 && HW > 0 && p >= 0.f && p < 1.f, "scda_instnorm_drop_bwd_hip")
    return instnorm_bwd_launch(dy, x, mean, rstd, dx, planes, HW, 0, 0.f, DropTail{nullptr, drop_threshold(p), seed, scale, 1, nullptr}, stream);
}

SCDA_API int scda_instnorm_drop_bwd_dev_hip(const float *dy, const float *x, const float *mean, const float *rstd, float *dx, int planes,
                                            int HW, float p, const uint64_t *seed_dev, float scale, void *stream) {
    NN_CHECK(dy && x && mean && rstd && dx && seed_dev && planes > 0 && HW > 0 && p >= 0.f && p < 1.f, "scda_instnorm_drop_bwd_dev_hip")
    return instnorm_bwd_launch(dy, x, mean, rstd, dx, planes, HW, 0, 0.f,
                               DropTail{nullptr, drop_threshold(p), 0ull, scale, 1, (const unsigned long long *)seed_dev}, stream);
}

SCDA_API size_t scda_batchnorm_workspace_bytes(int B, int C, int HW) {
    const int S = bn_slices(B, C, HW);
    return S > 1 ? (size_t)2 * C * S * sizeof(float) : 0;
}

SCDA_API int scda_batchnorm_fwd_hip(const float *x, float *y, const float *gamma, const float *beta, float *running_mean,
                                    float *running_var, float *save_mean, float *save_rstd, int B, int C, int HW,
                                    float eps, float momentum, int act, float slope, float *ws, void *stream) {
    NN_CHECK(x && y && gamma && beta && save_mean && save_rstd && B > 0 && C > 0 && HW > 0, "scda_batchnorm_fwd_hip")
    if (const int vpt = bn_plane_vpt(B, HW, x, y, nullptr)) {
#define BN_PLANE_FWD(V) hipLaunchKernelGGL(bn_plane_fwd_kernel<V>, dim3(C), dim3(1024), 0, as_stream(stream), x, y, gamma, beta, running_mean, running_var, save_mean, save_rstd, HW, eps, momentum, act, slope)
        if (vpt == 2) BN_PLANE_FWD(2); else if (vpt == 5) BN_PLANE_FWD(5); else if (vpt == 7) BN_PLANE_FWD(7); else if (vpt == 10) BN_PLANE_FWD(10); else BN_PLANE_FWD(16);
#undef BN_PLANE_FWD
        return launch_status("bn_plane_fwd_kernel");
    }
    const int S = bn_slices(B, C, HW);
    if (S > 1) {
        NN_CHECK(ws, "scda_batchnorm_fwd_hip (workspace)")
        const int BHW = B * HW, per = (BHW + S - 1) / S;
        hipStream_t st = as_stream(stream);
        float *psum = ws, *psq = ws + (size_t)C * S;
        hipLaunchKernelGGL(bn_slice_sum_kernel, dim3(C * S), dim3(256), 0, st, x, psum, C, HW, BHW, S, per);
        hipLaunchKernelGGL(bn_slice_sq_kernel, dim3(C * S), dim3(256), 0, st, x, psum, psq, C, HW, BHW, S, per);
        hipLaunchKernelGGL(bn_slice_apply_kernel, dim3(C * S), dim3(256), 0, st, x, y, gamma, beta, running_mean, running_var, save_mean,
                           save_rstd, psum, psq, C, HW, BHW, S, per, eps, momentum, act, slope);
        return launch_status("bn_slice kernels");
    }
    hipLaunchKernelGGL(batchnorm_fwd_kernel, dim3(C), dim3(256), 0, as_stream(stream), x, y, gamma, beta, running_mean, running_var,
                       save_mean, save_rstd, B, C, HW, eps, momentum, act, slope);
    return launch_status("batchnorm_fwd_kernel");
}

// relu(bn(x) + residual) of a ResNet block (models/mask_rcnn/resnet.py:95-104) in the batch norm's own pass, and its backward.
// Plane form only (batch 1: every layer of a batch-1 detector and the channel-major RoI head): scda_batchnorm_add_relu_ok() says
// whether a shape is served; callers run the two separate kernels otherwise.
SCDA_API int scda_batchnorm_add_relu_ok(int B, int HW) {
    static const float *aligned = nullptr;
    return bn_plane_vpt(B, HW, aligned, aligned, aligned, 10) != 0;
}

SCDA_API int scda_batchnorm_add_relu_fwd_hip(const float *x, const float *residual, float *y, const float *gamma, const float *beta,
                                             float *running_mean, float *running_var, float *save_mean, float *save_rstd, int B, int C,
                                             int HW, float eps, float momentum, void *stream) {
    NN_CHECK(x && residual && y && gamma && beta && save_mean && save_rstd && B > 0 && C > 0 && HW > 0, "scda_batchnorm_add_relu_fwd_hip")
    const int vpt = bn_plane_vpt(B, HW, x, y, residual, 10);
    NN_CHECK(vpt, "scda_batchnorm_add_relu_fwd_hip (shape / alignment not served: see scda_batchnorm_add_relu_ok)")
#define BN_PLANE_FWD(V) hipLaunchKernelGGL(bn_plane_fwd_kernel<V>, dim3(C), dim3(1024), 0, as_stream(stream), x, y, gamma, beta, running_mean, running_var, save_mean, save_rstd, HW, eps, momentum, 0, 0.f, residual)
    if (vpt == 2) BN_PLANE_FWD(2); else if (vpt == 5) BN_PLANE_FWD(5); else if (vpt == 7) BN_PLANE_FWD(7); else BN_PLANE_FWD(10);
#undef BN_PLANE_FWD
    return launch_status("bn_plane_fwd_kernel (residual join)");
}

SCDA_API int scda_batchnorm_add_relu_bwd_hip(const float *dy, const float *x, const float *y, const float *gamma, const float *beta,
                                             const float *save_mean, const float *save_rstd, float *dx, float *d_residual,
                                             float *dgamma, float *dbeta, int B, int C, int HW, int accumulate, void *stream) {
    NN_CHECK(dy && x && y && gamma && beta && save_mean && save_rstd && d_residual && dgamma && dbeta && B > 0 && C > 0 && HW > 0,
             "scda_batchnorm_add_relu_bwd_hip")
    const int vpt = bn_plane_vpt(B, HW, x, dy, dx, 10);
    NN_CHECK(vpt && !(((uintptr_t)y | (uintptr_t)d_residual) & 15),
             "scda_batchnorm_add_relu_bwd_hip (shape / alignment not served: see scda_batchnorm_add_relu_ok)")
#define BN_PLANE_BWD(V) hipLaunchKernelGGL(bn_plane_bwd_kernel<V>, dim3(C), dim3(1024), 0, as_stream(stream), dy, x, gamma, beta, save_mean, save_rstd, dx, dgamma, dbeta, HW, 0, 0.f, accumulate, y, d_residual)
    if (vpt == 2) BN_PLANE_BWD(2); else if (vpt == 5) BN_PLANE_BWD(5); else if (vpt == 7) BN_PLANE_BWD(7); else BN_PLANE_BWD(10);
#undef BN_PLANE_BWD
    return launch_status("bn_plane_bwd_kernel (residual join)");
}

SCDA_API int scda_batchnorm_bwd_hip(const float *dy, const float *x, const float *gamma, const float *beta,
                                    const float *save_mean, const float *save_rstd, float *dx, float *dgamma,
                                    float *dbeta, int B, int C, int HW, int act, float slope, int accumulate,
                                    float *ws, void *stream) {
    NN_CHECK(dy && x && gamma && beta && save_mean && save_rstd && dgamma && dbeta && B > 0 && C > 0 && HW > 0, "scda_batchnorm_bwd_hip")
    if (const int vpt = bn_plane_vpt(B, HW, x, dy, dx, 10)) {   // backward keeps TWO planes in registers: 16 float4 each would spill
#define BN_PLANE_BWD(V) hipLaunchKernelGGL(bn_plane_bwd_kernel<V>, dim3(C), dim3(1024), 0, as_stream(stream), dy, x, gamma, beta, save_mean, save_rstd, dx, dgamma, dbeta, HW, act, slope, accumulate)
        if (vpt == 2) BN_PLANE_BWD(2); else if (vpt == 5) BN_PLANE_BWD(5); else if (vpt == 7) BN_PLANE_BWD(7); else BN_PLANE_BWD(10);
#undef BN_PLANE_BWD
        return launch_status("bn_plane_bwd_kernel");
    }
    const int S = bn_slices(B, C, HW);
    if (S > 1) {
        NN_CHECK(ws, "scda_batchnorm_bwd_hip (workspace)")
        const int BHW = B * HW, per = (BHW + S - 1) / S;
        hipStream_t st = as_stream(stream);
        hipLaunchKernelGGL(bn_slice_grad_sums_kernel, dim3(C * S), dim3(256), 0, st, dy, x, gamma, beta, save_mean, save_rstd, ws, C, HW, BHW,
                           S, per, act, slope);
        hipLaunchKernelGGL(bn_slice_dx_kernel, dim3(C * S), dim3(256), 0, st, dy, x, gamma, beta, save_mean, save_rstd, ws, dx, dgamma, dbeta,
                           C, HW, BHW, S, per, act, slope, accumulate);
        return launch_status("bn_slice backward kernels");
    }
    hipLaunchKernelGGL(batchnorm_bwd_kernel, dim3(C), dim3(256), 0, as_stream(stream), dy, x, gamma, beta, save_mean, save_rstd, dx,
                       dgamma, dbeta, B, C, HW, act, slope, accumulate);
    return launch_status("batchnorm_bwd_kernel");
}

SCDA_API int scda_batchnorm_eval_hip(const float *x, const float *dy_or_null, float *out, const float *gamma, const float *beta,
                                     const float *running_mean, const float *running_var, int B, int C, int HW, float eps,
                                     int act, float slope, void *stream) {
    NN_CHECK(x && out && gamma && beta && running_mean && running_var && B > 0 && C > 0 && HW > 0, "scda_batchnorm_eval_hip")
    const long long total = (long long)B * C * HW;
    hipLaunchKernelGGL(batchnorm_eval_kernel, dim3(ew_grid(total) * 4), dim3(256), 0, as_stream(stream), x, out, gamma, beta,
                       running_mean, running_var, total, C, HW, eps, act, slope, dy_or_null);
    return launch_status("batchnorm_eval_kernel");
}

static float up_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

SCDA_API int scda_upsample2x_fwd_hip(const float *x, float *y, int planes, int IH, int IW, void *stream) {
    NN_CHECK(x && y && planes > 0 && IH > 0 && IW > 0, "scda_upsample2x_fwd_hip")
    const int OH = IH * 2, OW = IW * 2;
    if ((OH % 8) == 0)
        hipLaunchKernelGGL(upsample2_fwd_kernel<8>, dim3(planes * OH / 8), dim3(256), 0, as_stream(stream), x, y, IH, IW, OH, OW,
                           up_scale(IH, OH), up_scale(IW, OW));
    else
        hipLaunchKernelGGL(upsample2_fwd_kernel<2>, dim3(planes * OH / 2), dim3(256), 0, as_stream(stream), x, y, IH, IW, OH, OW,
                           up_scale(IH, OH), up_scale(IW, OW));
    return launch_status("upsample2_fwd_kernel");
}

SCDA_API int scda_upsample2x_bwd_hip(const float *dy, float *dx, int planes, int IH, int IW, void *stream) {
    NN_CHECK(dy && dx && planes > 0 && IH > 1 && IW > 1, "scda_upsample2x_bwd_hip")
    const int OH = IH * 2, OW = IW * 2;
    if (OW > kUpMaxOW) { set_error("scda_upsample2x_bwd_hip: rows wider than %d are not supported", kUpMaxOW); return SCDA_EINVAL; }
    const bool row_at_a_time = getenv("SCDA_UPSAMPLE_BWD_ROWWISE") != nullptr;   // A/B knob (read per launch: the test compares both kernels)
    if (!row_at_a_time && (IH % 8) == 0 && (OW % 4) == 0 && OW <= kUpRowsOW && ((((uintptr_t)dy) & 15) == 0))
        hipLaunchKernelGGL(upsample2_bwd_rows_kernel<8>, dim3(planes * IH / 8), dim3(256), 0, as_stream(stream), dy, dx, IH, IW, OH, OW,
                           up_scale(IH, OH), up_scale(IW, OW));
    else if ((IH % 4) == 0)
        hipLaunchKernelGGL(upsample2_bwd_kernel<4>, dim3(planes * IH / 4), dim3(256), 0, as_stream(stream), dy, dx, IH, IW, OH, OW,
                           up_scale(IH, OH), up_scale(IW, OW));
    else
        hipLaunchKernelGGL(upsample2_bwd_kernel<1>, dim3(planes * IH), dim3(256), 0, as_stream(stream), dy, dx, IH, IW, OH, OW,
                           up_scale(IH, OH), up_scale(IW, OW));
    return launch_status("upsample2_bwd_kernel");
}

// instance norm (+ act / + dropout-and-add tail) and the bilinear x2 behind it as ONE launch: y2 [planes, 2 IH, 2 IW]
SCDA_API int scda_instnorm_up2_supported(int IH, int IW) {
    const long long hw = (long long)IH * IW;
    return IH > 1 && IW > 1 && (hw == 4096 || hw == 16384) && (IW % 32) == 0 && !getenv("SCDA_NO_NORM_UP_FUSION");
}

static int instnorm_up2_launch(const float *x, float *y2, float *mean, float *rstd, int planes, int IH, int IW, float eps, int act,
                               float slope, const DropTail dt, void *stream, const char *who) {
    if (!scda_instnorm_up2_supported(IH, IW)) { set_error("%s: planes of %d x %d are not supported (scda_instnorm_up2_supported)", who, IH, IW); return SCDA_EINVAL; }
    if (((((uintptr_t)x) | ((uintptr_t)y2) | ((uintptr_t)dt.residual)) & 15) != 0) { set_error("%s: x, y2 and residual must be 16-byte aligned", who); return SCDA_EINVAL; }
    const float sh = up_scale(IH, 2 * IH), sw = up_scale(IW, 2 * IW);
    if ((long long)IH * IW == 4096)
        hipLaunchKernelGGL(instnorm_up2_fwd_kernel<1>, dim3(planes), dim3(1024), 0, as_stream(stream), x, y2, mean, rstd, IH, IW, eps, act, slope, dt, sh, sw);
    else
        hipLaunchKernelGGL(instnorm_up2_fwd_kernel<4>, dim3(planes), dim3(1024), 0, as_stream(stream), x, y2, mean, rstd, IH, IW, eps, act, slope, dt, sh, sw);
    return launch_status("instnorm_up2_fwd_kernel");
}

SCDA_API int scda_instnorm_up2_fwd_hip(const float *x, float *y2, float *mean, float *rstd, int planes, int IH, int IW, float eps,
                                       int act, float slope, void *stream) {
    NN_CHECK(x && y2 && mean && rstd && planes > 0, "scda_instnorm_up2_fwd_hip")
    return instnorm_up2_launch(x, y2, mean, rstd, planes, IH, IW, eps, act, slope, DropTail{nullptr, 0u, 0ull, 1.f, 0, nullptr}, stream,
                               "scda_instnorm_up2_fwd_hip");
}

SCDA_API int scda_instnorm_drop_add_up2_fwd_hip(const float *x, const float *residual, float *y2, float *mean, float *rstd, int planes,
                                                int IH, int IW, float eps, float p, uint64_t seed, float scale, void *stream) {
    NN_CHECK(x && residual && y2 && mean && rstd && planes > 0 && p >= 0.f && p < 1.f, "scda_instnorm_drop_add_up2_fwd_hip")
    return instnorm_up2_launch(x, y2, mean, rstd, planes, IH, IW, eps, 0, 0.f, DropTail{residual, drop_threshold(p), seed, scale, 1, nullptr}, stream,
                               "scda_instnorm_drop_add_up2_fwd_hip");
}

SCDA_API int scda_instnorm_drop_add_up2_fwd_dev_hip(const float *x, const float *residual, float *y2, float *mean, float *rstd, int planes,
                                                    int IH, int IW, float eps, float p, const uint64_t *seed_dev, float scale,
                                                    void *stream) {
    NN_CHECK(x && residual && y2 && mean && rstd && seed_dev && planes > 0 && p >= 0.f && p < 1.f, "scda_instnorm_drop_add_up2_fwd_dev_hip")
    return instnorm_up2_launch(x, y2, mean, rstd, planes, IH, IW, eps, 0, 0.f, DropTail{residual, drop_threshold(p), 0ull, scale, 1, (const unsigned long long *)seed_dev}, stream,
                               "scda_instnorm_drop_add_up2_fwd_dev_hip");
}

static int instnorm_up2_bwd_launch(const float *dy2, const float *x, const float *mean, const float *rstd, float *dx, float *dsmall,
                                   int planes, int IH, int IW, int act, float slope, const DropTail dt, void *stream, const char *who) {
    if (!scda_instnorm_up2_supported(IH, IW)) { set_error("%s: planes of %d x %d are not supported (scda_instnorm_up2_supported)", who, IH, IW); return SCDA_EINVAL; }
    if (((((uintptr_t)dy2) | ((uintptr_t)x) | ((uintptr_t)dx) | ((uintptr_t)dsmall)) & 15) != 0) { set_error("%s: dy2, x, dx and the residual's gradient must be 16-byte aligned", who); return SCDA_EINVAL; }
    const float sh = up_scale(IH, 2 * IH), sw = up_scale(IW, 2 * IW);
    if ((long long)IH * IW == 4096)
        hipLaunchKernelGGL(instnorm_up2_bwd_kernel<1>, dim3(planes), dim3(1024), 0, as_stream(stream), dy2, x, mean, rstd, dx, dsmall, IH, IW, act, slope, dt, sh, sw);
    else
        hipLaunchKernelGGL(instnorm_up2_bwd_kernel<4>, dim3(planes), dim3(1024), 0, as_stream(stream), dy2, x, mean, rstd, dx, dsmall, IH, IW, act, slope, dt, sh, sw);
    return launch_status("instnorm_up2_bwd_kernel");
}

SCDA_API int scda_instnorm_up2_bwd_hip(const float *dy2, const float *x, const float *mean, const float *rstd, float *dx, int planes,
                                       int IH, int IW, int act, float slope, void *stream) {
    NN_CHECK(dy2 && x && mean && rstd && dx && planes > 0, "scda_instnorm_up2_bwd_hip")
    return instnorm_up2_bwd_launch(dy2, x, mean, rstd, dx, nullptr, planes, IH, IW, act, slope, DropTail{nullptr, 0u, 0ull, 1.f, 0, nullptr}, stream,
                                   "scda_instnorm_up2_bwd_hip");
}

SCDA_API int scda_instnorm_drop_up2_bwd_hip(const float *dy2, const float *x, const float *mean, const float *rstd, float *dx,
                                            float *dresidual, int planes, int IH, int IW, float p, uint64_t seed, float scale, void *stream) {
    NN_CHECK(dy2 && x && mean && rstd && dx && dresidual && planes > 0 && p >= 0.f && p < 1.f, "scda_instnorm_drop_up2_bwd_hip")
    return instnorm_up2_bwd_launch(dy2, x, mean, rstd, dx, dresidual, planes, IH, IW, 0, 0.f, DropTail{nullptr, drop_threshold(p), seed, scale, 1, nullptr},
                                   stream, "scda_instnorm_drop_up2_bwd_hip");
}

SCDA_API int scda_instnorm_drop_up2_bwd_dev_hip(const float *dy2, const float *x, const float *mean, const float *rstd, float *dx,
                                                float *dresidual, int planes, int IH, int IW, float p, const uint64_t *seed_dev, float scale,
                                                void *stream) {
    NN_CHECK(dy2 && x && mean && rstd && dx && dresidual && seed_dev && planes > 0 && p >= 0.f && p < 1.f, "scda_instnorm_drop_up2_bwd_dev_hip")
    return instnorm_up2_bwd_launch(dy2, x, mean, rstd, dx, dresidual, planes, IH, IW, 0, 0.f,
                                   DropTail{nullptr, drop_threshold(p), 0ull, scale, 1, (const unsigned long long *)seed_dev}, stream,
                                   "scda_instnorm_drop_up2_bwd_dev_hip");
}

SCDA_API int scda_bce_fwd_hip(const float *p, const float *t, int n, float *out1, void *stream) {
    NN_CHECK(p && t && out1 && n > 0, "scda_bce_fwd_hip")
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(1), dim3(1024), 0, as_stream(stream), p, t, n, out1);
    return launch_status("bce_fwd_kernel");
}

SCDA_API int scda_bce_bwd_hip(const float *p, const float *t, int n, const float *grad_scalar, float *dp, void *stream) {
    NN_CHECK(p && t && grad_scalar && dp && n > 0, "scda_bce_bwd_hip")
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), p, t, n, grad_scalar, dp);
    return launch_status("bce_bwd_kernel");
}

SCDA_API int scda_gap_fwd_hip(const float *x, float *y, int planes, int HW, void *stream) {
    NN_CHECK(x && y && planes > 0 && HW > 0, "scda_gap_fwd_hip")
    if (HW <= 64 && planes >= 4096)
        hipLaunchKernelGGL(gap_fwd_small_kernel, dim3(ew_grid(planes)), dim3(256), (size_t)256 * HW * sizeof(float), as_stream(stream), x, y,
                           (long long)planes, HW);
    else
        hipLaunchKernelGGL(gap_fwd_kernel, dim3(planes), dim3(256), 0, as_stream(stream), x, y, HW);
    return launch_status("gap_fwd_kernel");
}

SCDA_API int scda_avg2x2s1_fwd_hip(const float *x, float *y, int planes, int H, int W, void *stream) {
    NN_CHECK(x && y && planes > 0 && H > 0 && W > 0, "scda_avg2x2s1_fwd_hip")
    const long long total = (long long)planes * H * W;
    hipLaunchKernelGGL(avg2x2s1_fwd_kernel, dim3(ew_grid(total) * 4), dim3(256), 0, as_stream(stream), x, y, total, H, W);
    return launch_status("avg2x2s1_fwd_kernel");
}

SCDA_API int scda_avg2x2s1_bwd_hip(const float *dy, float *dx, int planes, int H, int W, void *stream) {
    NN_CHECK(dy && dx && planes > 0 && H > 0 && W > 0, "scda_avg2x2s1_bwd_hip")
    const long long total = (long long)planes * (H + 1) * (W + 1);
    hipLaunchKernelGGL(avg2x2s1_bwd_kernel, dim3(ew_grid(total) * 4), dim3(256), 0, as_stream(stream), dy, dx, total, H, W);
    return launch_status("avg2x2s1_bwd_kernel");
}

SCDA_API int scda_gap_bwd_hip(const float *dy, float *dx, int planes, int HW, void *stream) {
    NN_CHECK(dy && dx && planes > 0 && HW > 0, "scda_gap_bwd_hip")
    const long long total = (long long)planes * HW;
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, as_stream(stream), dy, dx, total, HW);
    return launch_status("gap_bwd_kernel");
}

SCDA_API int scda_row_mean_hip(const float *x, float *y, int R, int C, void *stream) {
    NN_CHECK(x && y && R > 0 && C > 0, "scda_row_mean_hip")
    hipLaunchKernelGGL(row_mean_kernel, dim3(R), dim3(256), 0, as_stream(stream), x, y, C);
    return launch_status("row_mean_kernel");
}

// max_blocks > 0 sets the cap of the grid (grid-stride kernel) explicitly; 0 = the measured best.
SCDA_API int scda_adam_limited_hip(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int step, int max_blocks, void *stream) {
    NN_CHECK(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1 && max_blocks >= 0, "scda_adam_hip")
    if (n == 0) return SCDA_OK;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) {
        set_error("scda_adam_hip: buffers must be 16-byte aligned");
        return SCDA_EINVAL;
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    // Two resident workgroups per CU stream the bucket fastest: 137 M parameters take 0.65 ms (5.9 TB/s) with 512 workgroups,
    // 0.81 ms (4.7 TB/s) with the element-wise default of 4096 (scripts/time_adam.py on MI355X).
    int blocks = ew_grid(n / 4 + 1) * 2;
    if (max_blocks == 0) max_blocks = 512;
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2));
    return launch_status("adam_kernel");
}

SCDA_API int scda_adam_hip(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, float lr,
                           float beta1, float beta2, float eps, float weight_decay, int step, void *stream) {
    return scda_adam_limited_hip(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, 0, stream);
}
