// Probe for the exact-product split: an fp32 value is the sum of THREE bf16 pieces (8 + 8 + 8 significand bits, by truncation:
// hi = x & 0xffff0000, mid = (x - hi) & 0xffff0000, lo = x - hi - mid -- every subtraction exact), every bf16 x bf16 product is exact
// in fp32 (16 significand bits), so the nine piece products of a x b add up to the exact 48-bit product.  Question for the hardware:
// what does v_mfma_f32_32x32x16_bf16 do with the 16 products + C it sums -- is C[M][N] = sum_k a*b through nine such MFMAs per 16 k
// at least as close to an fp64 GEMM as the fp32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain) that the library uses today?
//   hipcc --offload-arch=gfx950 -O3 -o bf16x9_probe bf16x9_probe.hip && ./bf16x9_probe
// Prints (1) single-instruction diagnostics of the internal accumulation, (2) error statistics against fp64 on FC6 / FC7-sized K for
// the fp32 MFMA and for the nine-product form in several term orders, (3) a bare-loop rate of the nine-product form with the
// in-register split (VALU beside the matrix pipe).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned hi16(float x) { return __float_as_uint(x) & 0xffff0000u; }
// pieces of 8 consecutive-k values -> three packed bf16x8 fragments
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 &h, u32x4 &m, u32x4 &l) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        const unsigned h0 = hi16(x0), h1 = hi16(x1);
        const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
        const unsigned m0 = hi16(r0), m1 = hi16(r1);
        const float l0 = r0 - __uint_as_float(m0), l1 = r1 - __uint_as_float(m1);
        h[p] = (h0 >> 16) | h1;
        m[p] = (m0 >> 16) | m1;
        l[p] = (__float_as_uint(l0) >> 16) | (__float_as_uint(l1) & 0xffff0000u);
    }
}
#define MF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), C_, 0, 0, 0)

// one wave = one 32 x 32 tile of C = A[M][K] * B[N][K]^T (both K-contiguous).  MODE 0: fp32 MFMA (k-ordered fmaf chain);
// 1: nine bf16 products per 16 k, small terms first, one accumulator; 2: large terms first; 3: hi*hi in one accumulator, the eight
// others in a second, summed at the end; 4: six products (lo*lo, lo*mid, mid*lo dropped -- NOT a candidate, only to show what they carry)
template <int MODE>
__global__ void tile_gemm(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int N, int K) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const float *a = A + (size_t)(m0 + i) * K, *b = B + (size_t)(n0 + i) * K;
    f32x16 acc = {0}, acc2 = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k0 + 2 * t + h], b[k0 + 2 * t + h], acc, 0, 0, 0);
        } else {
            float xa[8], xb[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) { xa[t] = a[k0 + 8 * h + t]; xb[t] = b[k0 + 8 * h + t]; }
            u32x4 ah, am, al, bh, bm, bl;
            split8(xa, ah, am, al);
            split8(xb, bh, bm, bl);
            if (MODE == 1) { MF(al, bl, acc); MF(al, bm, acc); MF(am, bl, acc); MF(al, bh, acc); MF(ah, bl, acc); MF(am, bm, acc); MF(am, bh, acc); MF(ah, bm, acc); MF(ah, bh, acc); }
            if (MODE == 2) { MF(ah, bh, acc); MF(ah, bm, acc); MF(am, bh, acc); MF(am, bm, acc); MF(ah, bl, acc); MF(al, bh, acc); MF(am, bl, acc); MF(al, bm, acc); MF(al, bl, acc); }
            if (MODE == 3) { MF(al, bl, acc2); MF(al, bm, acc2); MF(am, bl, acc2); MF(al, bh, acc2); MF(ah, bl, acc2); MF(am, bm, acc2); MF(am, bh, acc2); MF(ah, bm, acc2); MF(ah, bh, acc); }
            if (MODE == 4) { MF(al, bh, acc); MF(ah, bl, acc); MF(am, bm, acc); MF(am, bh, acc); MF(ah, bm, acc); MF(ah, bh, acc); }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        C[(size_t)(m0 + row) * N + n0 + i] = acc[r] + acc2[r];
    }
}

// (1) one instruction: products p_k = a_k * b_k chosen so that the answer tells how the 16 products and C are summed
__global__ void one_mfma(const float *a16, const float *b16, float c0, float *out) {
    const int lane = threadIdx.x, h = lane >> 5;
    float xa[8], xb[8];
    for (int t = 0; t < 8; ++t) { xa[t] = a16[8 * h + t]; xb[t] = b16[8 * h + t]; }
    u32x4 ah, am, al, bh, bm, bl;
    split8(xa, ah, am, al);
    split8(xb, bh, bm, bl);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c0;
    MF(ah, bh, acc);
    if (lane == 0) out[0] = acc[0];
}

// (3) rate: per k-step a wave splits (TM + TN) * 8 fresh values (from registers rotated by a cheap VALU op so that nothing folds)
// and issues TM * TN * 9 MFMAs -- the in-register form of a 64 x 64 wave tile
template <int TM, int TN, bool SPLIT>
__global__ __launch_bounds__(512) void rate(float *out, int iters, float seed) {
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float xa[TM][8], xb[TN][8];
    for (int i = 0; i < TM; ++i) for (int t = 0; t < 8; ++t) xa[i][t] = seed + threadIdx.x * 0.001f + t + i;
    for (int j = 0; j < TN; ++j) for (int t = 0; t < 8; ++t) xb[j][t] = seed - threadIdx.x * 0.002f - t - j;
    u32x4 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
    for (int i = 0; i < TM; ++i) split8(xa[i], ah[i], am[i], al[i]);
    for (int j = 0; j < TN; ++j) split8(xb[j], bh[j], bm[j], bl[j]);
    for (int it = 0; it < iters; ++it) {
        if (SPLIT) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int t = 0; t < 8; ++t) xa[i][t] = __uint_as_float(__float_as_uint(xa[i][t]) ^ (unsigned)it);
                split8(xa[i], ah[i], am[i], al[i]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int t = 0; t < 8; ++t) xb[j][t] = __uint_as_float(__float_as_uint(xb[j][t]) ^ (unsigned)it);
                split8(xb[j], bh[j], bm[j], bl[j]);
            }
        }
        // product-major: the MFMAs on one accumulator are TM * TN instructions apart
#define ALL_TILES(PA_, PB_)                                      \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)           \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) MF(PA_[i], PB_[j], acc[i][j]);
        ALL_TILES(al, bl) ALL_TILES(al, bm) ALL_TILES(am, bl) ALL_TILES(al, bh) ALL_TILES(ah, bl) ALL_TILES(am, bm)
        ALL_TILES(am, bh) ALL_TILES(ah, bm) ALL_TILES(ah, bh)
#undef ALL_TILES
    }
    float s = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// (4) what the matrix pipe leaves for the other instructions of the SAME waves: 36 MFMAs per iteration on eight accumulators (the
// kernel's dependency distance: four instructions) with NV independent VALU operations / ND LDS reads fenced in behind every MFMA
template <int NV, int ND>
__global__ __launch_bounds__(512) void slots(float *out, int iters, float seed) {
    __shared__ float lds[8192];
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 a = {threadIdx.x, 2u, 3u, 4u}, b = {5u, 6u, threadIdx.x * 3u, 8u};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    lds[threadIdx.x] = seed; lds[threadIdx.x + 512] = seed;
    __syncthreads();
    float d = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 36; ++k) {
            asm volatile("" : "+v"(acc[k & 7]));
            MF(a, b, acc[k & 7]);
            asm volatile("" : "+v"(acc[k & 7]));
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = (k * NV + n) & 7;
                if (n & 1) v[q] = v[q] - __uint_as_float(__float_as_uint(v[(q + 3) & 7]) & 0xffff0000u);
                else v[q] = __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(v[q]), __float_as_uint(v[(q + 5) & 7]), 0x07060302u));
                asm volatile("" : "+v"(v[q]));
            }
#pragma unroll
            for (int n = 0; n < ND; ++n) { d += lds[(threadIdx.x + 64 * (k + n)) & 8191]; asm volatile("" : "+v"(d)); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = d;
    for (int i = 0; i < 8; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; s += v[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double frand() { return (double)rand() / RAND_MAX; }
static float gauss() { return (float)(sqrt(-2.0 * log(frand() + 1e-300)) * cos(6.283185307179586 * frand())); }

template <int MODE>
static void stats(const char *name, const float *dA, const float *dB, float *dC, const std::vector<double> &ref, const std::vector<double> &absref, int M, int N, int K) {
    hipMemset(dC, 0, (size_t)M * N * 4);
    tile_gemm<MODE><<<dim3(N / 32, M / 32), 64>>>(dA, dB, dC, N, K);
    std::vector<float> c((size_t)M * N);
    hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, sq = 0, mxr = 0;
    for (size_t i = 0; i < c.size(); ++i) {
        const double e = fabs((double)c[i] - ref[i]) / absref[i];      // relative to sum_k |a b|: the scale round-off is proportional to
        mx = fmax(mx, e); sq += e * e;
        mxr = fmax(mxr, fabs((double)c[i] - ref[i]));
    }
    printf("  %-46s max %.3e  rms %.3e  (x sum|ab|)   max abs %.3e\n", name, mx, sqrt(sq / c.size()), mxr);
}

int main() {
    // ---- (1) single instruction ----
    {
        float *da, *db, *dout; hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
        struct { const char *what; float a[16]; float c; double exact; } T[] = {
            {"2^24 + 15 x 1, C = 0        (fmaf chain: 16777216; one rounding of the exact sum: 16777232)", {16777216.f, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, 0.f, 16777231.0},
            {"15 x 1 then 2^24, C = 0     (fmaf chain: 16777232 either way? no: 15 + 2^24 = 16777231 -> 16777232)", {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 16777216.f}, 0.f, 16777231.0},
            {"16 x 1, C = 2^24            (C added first and rounded per product: 16777216; exact: 16777232)", {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, 16777216.f, 16777232.0},
            {"2^24, -2^24, 14 x 1.5, C = 0 (cancellation inside: exact 21)", {16777216.f, -16777216.f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f, 1.5f}, 0.f, 21.0},
            {"2^40, 15 x 1, C = -2^40     (alignment width: exact 15)", {1099511627776.f, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, -1099511627776.f, 15.0},
            {"2^30, 15 x 1, C = -2^30     (exact 15)", {1073741824.f, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, -1073741824.f, 15.0},
            {"2^26, 15 x 1, C = -2^26     (exact 15)", {67108864.f, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, -67108864.f, 15.0},
        };
        printf("(1) one v_mfma_f32_32x32x16_bf16 (hi pieces only; the values are exact bf16): what comes out\n");
        for (auto &t : T) {
            float b[16]; for (int k = 0; k < 16; ++k) b[k] = 1.f;
            hipMemcpy(da, t.a, 64, hipMemcpyHostToDevice); hipMemcpy(db, b, 64, hipMemcpyHostToDevice);
            one_mfma<<<1, 64>>>(da, db, t.c, dout);
            float r; hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
            printf("  %-100s -> %.1f (exact %.1f)\n", t.what, (double)r, t.exact);
        }
    }
    // ---- (2) GEMM error statistics against fp64 ----
    const int M = 128, N = 128;
    for (int K : {4096, 25088}) {
        for (int dist = 0; dist < 2; ++dist) {
            srand(1234 + K + dist);
            std::vector<float> A((size_t)M * K), B((size_t)N * K);
            // dist 0: N(0,1) x N(0, 1/sqrt(K)) (FC weights against features); dist 1: post-ReLU features (half zeros, positive) x weights
            for (auto &v : A) { v = gauss(); if (dist == 1) v = v > 0 ? v : 0.f; }
            for (auto &v : B) v = gauss() / sqrtf((float)K);
            std::vector<double> ref((size_t)M * N), absref((size_t)M * N);
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) {
                    double s = 0, sa = 0;
                    const float *a = &A[(size_t)m * K], *b = &B[(size_t)n * K];
                    for (int k = 0; k < K; ++k) { const double p = (double)a[k] * (double)b[k]; s += p; sa += fabs(p); }
                    ref[(size_t)m * N + n] = s; absref[(size_t)m * N + n] = sa > 0 ? sa : 1.0;
                }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            printf("(2) C[%d][%d], K = %d, %s: error / sum_k|a b| against an fp64 GEMM\n", M, N, K, dist ? "post-ReLU features x weights" : "normal x normal / sqrt(K)");
            stats<0>("fp32 MFMA 32x32x2 (today's kernels: fmaf chain)", dA, dB, dC, ref, absref, M, N, K);
            stats<1>("bf16 x 9, small terms first, one accumulator", dA, dB, dC, ref, absref, M, N, K);
            stats<2>("bf16 x 9, large terms first, one accumulator", dA, dB, dC, ref, absref, M, N, K);
            stats<3>("bf16 x 9, hi*hi apart from the eight others", dA, dB, dC, ref, absref, M, N, K);
            stats<4>("bf16 x 6 (three smallest dropped: NOT a candidate)", dA, dB, dC, ref, absref, M, N, K);
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    }
    // ---- (3) bare-loop rate ----
    {
        float *out; hipMalloc(&out, (size_t)256 * 512 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 2000;
        auto run = [&](auto kern, const char *name, int tm, int tn) {
            kern<<<256, 512>>>(out, 10, 1.f);
            hipEventRecord(e0);
            kern<<<256, 512>>>(out, iters, 1.f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double prod_flop = 256.0 * 8 * iters * tm * tn * 2.0 * 32 * 32 * 16;     // fp32-equivalent work (one product per a*b)
            printf("  %-64s %.3f ms  %.1f TFLOP/s fp32-equivalent (x9 = %.0f bf16 TFLOP/s)\n", name, ms, prod_flop / ms / 1e9, 9 * prod_flop / ms / 1e9);
        };
        printf("(3) bare loops, 8 waves per CU (two per SIMD), 256 workgroups:\n");
        run(rate<2, 2, false>, "64 x 64 wave tile, 36 MFMAs per k16, no split (pure matrix pipe)", 2, 2);
        run(rate<2, 2, true>, "64 x 64 wave tile, 36 MFMAs + in-register split of 32 values", 2, 2);
        run(rate<2, 4, false>, "64 x 128 wave tile, 72 MFMAs per k16, no split", 2, 4);
        run(rate<2, 4, true>, "64 x 128 wave tile, 72 MFMAs + in-register split of 48 values", 2, 4);
    }
    {
        float *out; hipMalloc(&out, (size_t)256 * 512 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 2000;
        auto run = [&](auto kern, int nv, int nd) {
            kern<<<256, 512>>>(out, 10, 1.f);
            hipEventRecord(e0);
            kern<<<256, 512>>>(out, iters, 1.f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  %d VALU + %d ds_read_b32 behind every MFMA: %.3f ms = %.2f us per 72 MFMAs of a SIMD (bare %s)\n", nv, nd, ms, ms * 1e3 / iters, nv + nd ? "see 0 + 0" : "loop");
        };
        printf("(4) issue slots beside the MFMA stream, 8 waves per CU (two per SIMD), 36 MFMAs per wave and iteration:\n");
        run(slots<0, 0>, 0, 0); run(slots<1, 0>, 1, 0); run(slots<2, 0>, 2, 0); run(slots<3, 0>, 3, 0); run(slots<4, 0>, 4, 0); run(slots<6, 0>, 6, 0);
        run(slots<0, 1>, 0, 1); run(slots<2, 1>, 2, 1);
    }
    return 0;
}
