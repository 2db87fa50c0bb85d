// wino_pack.h -- the Winograd kernel's transformed-filter layout (conv_wino.hip), shared with the batched weight re-pack of
// conv_gemm.hip (one launch for all conv weights of an optimiser group).
//   U_ab[m][c] = (G g G^T)[a][b],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
//   g = w[m][c] (forward: M = Cout rows, C = Cin reduced) or the data gradient's filter w[c][m] rotated by 180 degrees
//       (M = Cin rows, C = Cout reduced)
//   index = (((xi * n_mbg + m / 32) * (C / 8) + c / 8) * 64 + (c & 1) * 32 + m % 32) * 4 + (c % 8) / 2      rows m >= M are zero
// i.e. one (position, 32-row block, 8-channel slab) is 256 consecutive floats in the MFMA A-fragment order: lane = (c & 1) * 32 +
// m % 32 holds the four K-pairs of its row as one float4.
#pragma once
#include "common.h"

namespace scda {

constexpr int WINO_BK = 8;   // channels per K-slab of conv_wino_kernel

__host__ __device__ inline long long wino_packed_elems(int M, int C) { return 16LL * ((M + 63) / 64 * 64) * C; }
// tiles of pack_tile_wino: one per (32-row block, slab)
__host__ __device__ inline long long wino_pack_tiles(int M, int C) { return (long long)((M + 63) / 64 * 2) * (C / WINO_BK); }

// the 16 transformed values of one filter g[3][3]
__device__ __forceinline__ void wino_filter_transform(const float (&g)[9], float (&u)[16]) {
    float r[4][3];   // G g
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        r[0][v] = g[v];
        r[1][v] = 0.5f * (g[v] + g[3 + v] + g[6 + v]);
        r[2][v] = 0.5f * (g[v] - g[3 + v] + g[6 + v]);
        r[3][v] = g[6 + v];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        u[a * 4 + 0] = r[a][0];
        u[a * 4 + 1] = 0.5f * (r[a][0] + r[a][1] + r[a][2]);
        u[a * 4 + 2] = 0.5f * (r[a][0] - r[a][1] + r[a][2]);
        u[a * 4 + 3] = r[a][2];
    }
}

// One workgroup (256 threads) = tile t = (32-row block mbg, slab s): thread (m_l = tid / 8, c_l = tid % 8) reads its filter (the
// 8 channels of a row are 72 consecutive floats forward; the 32 rows of a channel 288 consecutive floats for the data gradient),
// transforms it, and the 16 x 256 results leave through LDS as sixteen 1-KB runs.  tile: 16 * 256 floats of LDS.
__device__ __forceinline__ void pack_tile_wino(const float *__restrict__ w, float *__restrict__ out, const int Cout, const int Cin,
                                               const int for_dgrad, const int t, float *tile) {
    const int M = for_dgrad ? Cin : Cout, C = for_dgrad ? Cout : Cin;
    const int n_mbg = (M + 63) / 64 * 2, n_slab = C / WINO_BK;
    const int mbg = t / n_slab, s = t - mbg * n_slab;
    const int tid = threadIdx.x;
    // read order follows the source's fastest dimension; the thread's (row, channel) follows from it
    const int m_l = for_dgrad ? (tid & 31) : (tid >> 3), c_l = for_dgrad ? (tid >> 5) : (tid & 7);
    const int m = mbg * 32 + m_l, c = s * WINO_BK + c_l;
    float g[9], u[16];
    if (m < M) {
        const float *src = for_dgrad ? w + ((size_t)c * Cin + m) * 9 : w + ((size_t)m * Cin + c) * 9;
#pragma unroll
        for (int r = 0; r < 9; ++r) g[for_dgrad ? 8 - r : r] = src[r];     // data gradient: rotated by 180 degrees
        wino_filter_transform(g, u);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) u[i] = 0.f;
    }
    const int slot = ((c_l & 1) * 32 + m_l) * 4 + (c_l >> 1);
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) tile[xi * 256 + slot] = u[xi];
    __syncthreads();
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) out[((size_t)(xi * n_mbg + mbg) * n_slab + s) * 256 + tid] = tile[xi * 256 + tid];
}

}  // namespace scda
