// prototype: implicit-GEMM 3x3 s1 p1 conv forward with DIRECT-TO-LDS operand staging (global_load_lds), a 3-stage LDS ring,
// counted vmcnt and ONE barrier per K-slab -- against the register-staged kernel shape that conv_gemm.hip ships.
//   hipcc --offload-arch=gfx950 -O3 -o conv_glds conv_glds.hip && ./conv_glds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../scda_amd/csrc/mfma_tile.h"
using namespace scda;
namespace scda { void set_error(const char*, ...) {} void prof_begin(int,double,hipStream_t){} void prof_end(hipStream_t){} }

struct G { int C, H, W, M, N, K; };
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// ---- baseline: register staged, K order channel-block major, weights [M][K] ------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256) void kbase(const float* __restrict__ Wm, const float* __restrict__ X, float* __restrict__ Y, G g, const float* zp) {
    constexpr int BKK = 16, LDA = BM + 4, LDB = BN + 4, WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int AE = BKK * BM / 256, BE = BKK * BN / 256, KS = 256 / BN;
    __shared__ __attribute__((aligned(16))) float lds[2 * BKK * (LDA + LDB)];
    auto As = [&](int b) { return lds + b * (BKK * LDA); };
    auto Bs = [&](int b) { return lds + 2 * BKK * LDA + b * (BKK * LDB); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nb = tid % BN, kb = tid / BN;
    const int n = n0 + nb, py = n / g.W, px = n % g.W, plane = g.H * g.W;
    float ar[AE], br[BE];
    const int qa = tid & 3, rva = tid >> 2;
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < AE / 4; ++j) {
            const int m = m0 + rva + 64 * j;
            const float4 v = *reinterpret_cast<const float4*>(Wm + (size_t)m * g.K + k0 + 4 * qa);
            ar[4 * j] = v.x; ar[4 * j + 1] = v.y; ar[4 * j + 2] = v.z; ar[4 * j + 3] = v.w;
        }
        const int sl = k0 / 16, cb = sl / 9, r = sl - cb * 9, kh = r / 3, kw = r - kh * 3;
        const int iy = py + kh - 1, ix = px + kw - 1;
        const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        const float* src = ok ? X + (size_t)(cb * 16 + kb) * plane + iy * g.W + ix : zp;
        const size_t st = ok ? (size_t)KS * plane : 0;
#pragma unroll
        for (int j = 0; j < BE; ++j) br[j] = src[j * st];
    };
    auto sstore = [&](int b) {
#pragma unroll
        for (int j = 0; j < AE / 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) As(b)[(4 * qa + i) * LDA + rva + 64 * j] = ar[4 * j + i];
#pragma unroll
        for (int j = 0; j < BE; ++j) Bs(b)[(kb + KS * j) * LDB + nb] = br[j];
    };
    f32x16 acc[TM][TN];
    zero_acc<BM, BN>(acc);
    gload(0); sstore(0); __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < g.K; k0 += BKK) {
        const bool more = k0 + BKK < g.K;
        if (more) gload(k0 + BKK);
        mma_slab<BM, BN>(As(buf), Bs(buf), acc, wm, wn, lane);
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    const int lr = lane & 31;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nn = n0 + wn * WN + j * 32 + lr;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[(size_t)(m0 + wm * WM + i * 32 + frag_row(r, lane)) * g.N + nn] = acc[i][j][r];
    }
}

// ---- direct-to-LDS: weights [K][M] (M contiguous), 128x128 tile, BK 16, NST-stage ring --------------------------------
__device__ __forceinline__ void glds4(const float* src, float* lds_dst_wave_uniform, bool use_asm) {
    if (use_asm) {
        unsigned keep;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_dst_wave_uniform);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    } else {
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)lds_dst_wave_uniform, 4, 0, 0);
    }
}
__device__ __forceinline__ void glds16(const float* src, float* lds_dst_wave_uniform, bool use_asm) {
    if (use_asm) {
        unsigned keep;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_dst_wave_uniform);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    } else {
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)lds_dst_wave_uniform, 16, 0, 0);
    }
}

template <int NST, bool ASM, int PIPE>
__global__ __launch_bounds__(256) void kglds(const float* __restrict__ Wt, const float* __restrict__ X, float* __restrict__ Y, G g, const float* zp) {
    constexpr int BM = 128, BN = 128, BKK = 16, LD = 128, STAGE = BKK * 2 * LD, WM = 64, WN = 64, TM = 2, TN = 2;
    __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int half = wave & 1, kg = wave >> 1;
    const int n = n0 + half * 64 + lane, py = n / g.W, px = n % g.W, plane = g.H * g.W;
    const int nslab = g.K / BKK;
    const float* wsrc = Wt + (size_t)(4 * wave + (lane >> 5)) * g.M + m0 + (lane & 31) * 4;

    auto issue = [&](int s, int buf) {
        float* Ab = lds + buf * STAGE;
        float* Bb = Ab + BKK * LD;
        const float* wa = wsrc + (size_t)s * BKK * g.M;
        glds16(wa, Ab + (4 * wave) * LD, ASM);
        glds16(wa + (size_t)2 * g.M, Ab + (4 * wave + 2) * LD, ASM);
        const int cb = s / 9, r = s - cb * 9, kh = r / 3, kw = r - kh * 3;
        const int iy = py + kh - 1, ix = px + kw - 1;
        const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        const float* src = ok ? X + (size_t)(cb * 16 + kg * 8) * plane + iy * g.W + ix : zp;
        const size_t st = ok ? (size_t)plane : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) glds4(src + j * st, Bb + (kg * 8 + j) * LD + half * 64, ASM);
    };

    f32x16 acc[TM][TN];
    zero_acc<BM, BN>(acc);
    const int lr = lane & 31, lk = lane >> 5;

    constexpr int DIST = PIPE == 3 ? 3 : 2;
    issue(0, 0);
    if (nslab > 1) issue(1, 1);
    if (DIST == 3 && nslab > 2) issue(2, 2);
    int buf = 0, nbuf = DIST % NST;
    for (int s = 0; s < nslab; ++s) {
        if (PIPE == 2) {
            if (s + 2 < nslab) { issue(s + 2, nbuf); asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); }
            else if (s + 1 < nslab) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (DIST == 3) {
            if (s + 2 < nslab) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else if (s + 1 < nslab) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (s + 1 < nslab) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (PIPE != 2 && s + DIST < nslab) issue(s + DIST, nbuf);
        const float* ap = lds + buf * STAGE + lk * LD + wm * WM + lr;
        const float* bp = lds + buf * STAGE + BKK * LD + lk * LD + wn * WN + lr;
        if (PIPE == 0) {
#pragma unroll
            for (int kp = 0; kp < BKK / 2; ++kp) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = ap[(2 * kp) * LD + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = bp[(2 * kp) * LD + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // fragment double buffering: the LDS reads of K-pair kp+1 are issued before the MFMAs of K-pair kp
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = ap[i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = bp[j * 32];
#pragma unroll
            for (int kp = 0; kp < BKK / 2; ++kp) {
                const int cur = kp & 1, nxt = cur ^ 1;
                if (kp + 1 < BKK / 2) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[nxt][i] = ap[(2 * kp + 2) * LD + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[nxt][j] = bp[(2 * kp + 2) * LD + j * 32];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            }
        }
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nn = n0 + wn * WN + j * 32 + lr;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[(size_t)(m0 + wm * WM + i * 32 + frag_row(r, lane)) * g.N + nn] = acc[i][j][r];
    }
}

// ---- 8 waves, 256 x 128 tile: the gathered B operand is shared by twice as many output channels ------------------------
template <int NST>
__global__ __launch_bounds__(512) void kglds8(const float* __restrict__ Wt, const float* __restrict__ X, float* __restrict__ Y, G g, const float* zp) {
    constexpr int BM = 256, BN = 128, BKK = 16, STAGE = BKK * (BM + BN), WM = 64, WN = 64, TM = 2, TN = 2;
    __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7
    const int wm = wave >> 1, wn = wave & 1;                       // 4 x 2
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    // B: 16 rows x 2 halves = 32 instr / 8 waves = 4 per wave: wave -> half = wave&1, rows (wave>>1)*4 .. +3
    const int half = wave & 1, kg = wave >> 1;
    const int n = n0 + half * 64 + lane, py = n / g.W, px = n % g.W, plane = g.H * g.W;
    const int nslab = g.K / BKK;
    // A: 16 rows x 256 floats = 16 x4-instr (one row each: 64 lanes x 4) / 8 waves = 2 per wave: rows 2*wave, 2*wave+1
    const float* wsrc = Wt + (size_t)(2 * wave) * g.M + m0 + lane * 4;
    auto issue = [&](int s, int buf) {
        float* Ab = lds + buf * STAGE;
        float* Bb = Ab + BKK * BM;
        const float* wa = wsrc + (size_t)s * BKK * g.M;
        __builtin_amdgcn_global_load_lds((glb_void*)wa, (lds_void*)(Ab + (2 * wave) * BM), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(wa + g.M), (lds_void*)(Ab + (2 * wave + 1) * BM), 16, 0, 0);
        const int cb = s / 9, r = s - cb * 9, kh = r / 3, kw = r - kh * 3;
        const int iy = py + kh - 1, ix = px + kw - 1;
        const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        const float* src = ok ? X + (size_t)(cb * 16 + kg * 4) * plane + iy * g.W + ix : zp;
        const size_t st = ok ? (size_t)plane : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((glb_void*)(src + j * st), (lds_void*)(Bb + (kg * 4 + j) * BN + half * 64), 4, 0, 0);
    };
    f32x16 acc[TM][TN];
    zero_acc<128, 128>(acc);
    const int lr = lane & 31, lk = lane >> 5;
    issue(0, 0);
    if (nslab > 1) issue(1, 1);
    int buf = 0, nbuf = 2;
    for (int s = 0; s < nslab; ++s) {
        if (s + 2 < nslab) { issue(s + 2, nbuf); asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
        else if (s + 1 < nslab) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float* ap = lds + buf * STAGE + lk * BM + wm * WM + lr;
        const float* bp = lds + buf * STAGE + BKK * BM + lk * BN + wn * WN + lr;
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = ap[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = bp[j * 32];
#pragma unroll
        for (int kp = 0; kp < BKK / 2; ++kp) {
            const int cur = kp & 1, nxt = cur ^ 1;
            if (kp + 1 < BKK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[nxt][i] = ap[(2 * kp + 2) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[nxt][j] = bp[(2 * kp + 2) * BN + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
        }
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nn = n0 + wn * WN + j * 32 + lr;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[(size_t)(m0 + wm * WM + i * 32 + frag_row(r, lane)) * g.N + nn] = acc[i][j][r];
    }
}

template <typename F>
float timeit(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(a);
    const int it = 20;
    for (int i = 0; i < it; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / it;
}

int main() {
    struct { const char* n; int C, H, W, M; } L[] = {{"conv2_2", 128, 256, 512, 128}, {"conv3_2", 256, 128, 256, 256}, {"conv4_2", 512, 64, 128, 512}, {"conv5_x", 512, 32, 64, 512}};
    float* zp; hipMalloc(&zp, 1 << 16); hipMemset(zp, 0, 1 << 16);
    for (auto& l : L) {
        G g{l.C, l.H, l.W, l.M, l.H * l.W, l.C * 9};
        float *W, *Wt, *X, *Y, *Y2;
        hipMalloc(&W, (size_t)g.M * g.K * 4); hipMalloc(&Wt, (size_t)g.M * g.K * 4); hipMalloc(&X, (size_t)g.C * g.N * 4);
        hipMalloc(&Y, (size_t)g.M * g.N * 4); hipMalloc(&Y2, (size_t)g.M * g.N * 4);
        std::vector<float> h((size_t)g.C * g.N); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
        hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> hw((size_t)g.M * g.K), hwt((size_t)g.M * g.K);
        for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) % 1000) / 10000.f - 0.05f;
        for (int m = 0; m < g.M; ++m) for (int k = 0; k < g.K; ++k) hwt[(size_t)k * g.M + m] = hw[(size_t)m * g.K + k];
        hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(Wt, hwt.data(), hwt.size() * 4, hipMemcpyHostToDevice);
        printf("%s  M=%d N=%d K=%d\n", l.n, g.M, g.N, g.K);
        const double fl = 2.0 * g.M * g.N * g.K;
        dim3 grid(g.N / 128, g.M / 128);
        float ms = timeit([&] { hipLaunchKernelGGL((kbase<128, 128>), grid, dim3(256), 0, 0, W, X, Y, g, zp); });
        printf("  %-40s %7.3f ms  %6.1f TF\n", "register staged (shipping shape)", ms, fl / ms / 1e9);
        std::vector<float> y0((size_t)g.M * g.N), y1((size_t)g.M * g.N);
        hipMemcpy(y0.data(), Y, y0.size() * 4, hipMemcpyDeviceToHost);
        auto check = [&](const char* nm, float t) {
            hipMemcpy(y1.data(), Y2, y1.size() * 4, hipMemcpyDeviceToHost);
            size_t bad = 0; for (size_t i = 0; i < y0.size(); ++i) bad += memcmp(&y0[i], &y1[i], 4) != 0;
            printf("  %-40s %7.3f ms  %6.1f TF   mismatching outputs: %zu\n", nm, t, fl / t / 1e9, bad);
            hipMemset(Y2, 0, y1.size() * 4);
        };
        ms = timeit([&] { hipLaunchKernelGGL((kglds<3, false, 0>), grid, dim3(256), 0, 0, Wt, X, Y2, g, zp); });
        check("glds builtin, 3 stages, 1 barrier", ms);
        ms = timeit([&] { hipLaunchKernelGGL((kglds<3, false, 1>), grid, dim3(256), 0, 0, Wt, X, Y2, g, zp); });
        check("glds builtin, 3 stages, frag dbuf", ms);
        ms = timeit([&] { hipLaunchKernelGGL((kglds<4, false, 1>), grid, dim3(256), 0, 0, Wt, X, Y2, g, zp); });
        check("glds builtin, 4 stages, frag dbuf", ms);
        ms = timeit([&] { hipLaunchKernelGGL((kglds<4, false, 2>), grid, dim3(256), 0, 0, Wt, X, Y2, g, zp); });
        check("glds 4 stages, issue before barrier", ms);
        ms = timeit([&] { hipLaunchKernelGGL((kglds<4, false, 3>), grid, dim3(256), 0, 0, Wt, X, Y2, g, zp); });
        check("glds 4 stages, prefetch distance 3", ms);
        if (g.M % 256 == 0) {
            dim3 grid8(g.N / 128, g.M / 256);
            ms = timeit([&] { hipLaunchKernelGGL((kglds8<4>), grid8, dim3(512), 0, 0, Wt, X, Y2, g, zp); });
            check("8 waves 256x128, 4 stages", ms);
            ms = timeit([&] { hipLaunchKernelGGL((kglds8<3>), grid8, dim3(512), 0, 0, Wt, X, Y2, g, zp); });
            check("8 waves 256x128, 3 stages", ms);
        }
        hipFree(W); hipFree(Wt); hipFree(X); hipFree(Y); hipFree(Y2);
    }
    return 0;
}
