// mfma_tile.h -- shared device code for the fp32 MFMA GEMM family (gfx950).
//
// All dense contractions of the SCDA step (VGG/RPN/decoder/discriminator convs,
// FC6/FC7/heads) run on v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate, exact
// fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak.
// The reference computes in fp32 (cuDNN/cuBLAS), so this is the parity dtype.
//
// Workgroup = 256 threads = 4 waves in a 2x2 arrangement; workgroup tile BM x BN,
// K-step BK=16; each wave owns a (BM/2)x(BN/2) sub-tile made of 32x32 MFMA tiles.
// LDS holds A as [k][m] and B as [k][n] (k-major) so that the MFMA operand fetch
//   A[i = lane&31][k = lane>>5],  B[k = lane>>5][j = lane&31]
// is one conflict-free ds_read_b32 per operand (two 32-lane groups, each reading
// 32 consecutive words).
#pragma once
#include "common.h"

namespace scda {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;

template <int BM, int BN, int BKT = BK>
struct TileCfg {
    static constexpr int LDA = BM + 4;  // +4 words: keeps rows 16B aligned, skews banks
    static constexpr int LDB = BN + 4;
    static constexpr int WM = BM / 2, WN = BN / 2;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int A_ELEMS = BKT * BM / 256;  // elements each thread stages per K-step
    static constexpr int B_ELEMS = BKT * BN / 256;
    static constexpr int LDS_FLOATS = 2 * BKT * (LDA + LDB);
};

// integer division by a runtime-constant divisor; power-of-two divisors (every
// VGG / decoder extent) take the shift path
struct Div {
    int d;
    int shift;  // >=0: power of two
    __host__ __device__ Div() : d(1), shift(0) {}
    __host__ explicit Div(int dd) : d(dd), shift(-1) {
        if (dd > 0 && (dd & (dd - 1)) == 0) {
            shift = 0;
            while ((1 << shift) < dd) ++shift;
        }
    }
    __device__ __forceinline__ int div(int n) const { return shift >= 0 ? (n >> shift) : (n / d); }
    __device__ __forceinline__ void divmod(int n, int &q, int &r) const {
        q = div(n);
        r = n - q * d;
    }
};

template <int BM, int BN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[BM / 64][BN / 64]) {
#pragma unroll
    for (int i = 0; i < BM / 64; ++i)
#pragma unroll
        for (int j = 0; j < BN / 64; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// one BK-deep slab of MFMAs out of LDS
template <int BM, int BN, int BKT = BK>
__device__ __forceinline__ void mma_slab(const float *__restrict__ As, const float *__restrict__ Bs,
                                         f32x16 (&acc)[BM / 64][BN / 64], const int wm,
                                         const int wn, const int lane) {
    using T = TileCfg<BM, BN, BKT>;
    const int lr = lane & 31, lk = lane >> 5;
    const float *ap = As + lk * T::LDA + wm * T::WM + lr;
    const float *bp = Bs + lk * T::LDB + wn * T::WN + lr;
    // operand fragments double-buffered in registers: the LDS reads of K-pair kp+1 are issued before the MFMAs of kp
    float a[2][T::TM], b[2][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i) a[0][i] = ap[i * 32];
#pragma unroll
    for (int j = 0; j < T::TN; ++j) b[0][j] = bp[j * 32];
#pragma unroll
    for (int kp = 0; kp < BKT / 2; ++kp) {
        const int cur = kp & 1, nxt = cur ^ 1;
        if (kp + 1 < BKT / 2) {
#pragma unroll
            for (int i = 0; i < T::TM; ++i) a[nxt][i] = ap[(2 * kp + 2) * T::LDA + i * 32];
#pragma unroll
            for (int j = 0; j < T::TN; ++j) b[nxt][j] = bp[(2 * kp + 2) * T::LDB + j * 32];
        }
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    }
}

// C/D fragment coordinates of the 32x32 MFMA: reg r of lane l holds
//   row = (r&3) + 8*(r>>2) + 4*(l>>5),  col = l&31
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_LEAKY) return v > 0.f ? v : v * slope;
    return v;
}

}  // namespace scda
