// standalone ablation of the implicit-GEMM conv forward kernel (3x3 s1 p1): which phase bounds it?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../scda_amd/csrc/mfma_tile.h"
using namespace scda;
namespace scda { void set_error(const char*, ...) {} void prof_begin(int,double,hipStream_t){} void prof_end(hipStream_t){} }

struct G { int C, H, W, M, N, K; };

// VAR: 0 baseline, 1 no B gather (constant), 2 no global loads at all, 3 no MFMA, 4 no LDS store+barrier in loop
template <int BM, int BN, int BKK, int VAR>
__global__ __launch_bounds__(256) void k(const float* __restrict__ Wm, const float* __restrict__ X, float* __restrict__ Y, G g) {
    constexpr int LDA = BM + 4, LDB = BN + 4, WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int AE = BKK * BM / 256, BE = BKK * BN / 256, KS = 256 / BN;
    __shared__ __attribute__((aligned(16))) float lds[2 * BKK * (LDA + LDB)];
    auto As = [&](int b) { return lds + b * (BKK * LDA); };
    auto Bs = [&](int b) { return lds + 2 * BKK * LDA + b * (BKK * LDB); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int ka = tid % BKK, ra = tid / BKK;     // A: lanes along K
    const int nb = tid % BN, kb = tid / BN;
    const int n = n0 + nb, py = n / g.W, px = n % g.W, plane = g.H * g.W;
    float ar[AE], br[BE];
    const int qa = tid & 3, rva = tid >> 2;
    auto gload = [&](int k0) {
        if (VAR == 2) {
#pragma unroll
            for (int j = 0; j < AE; ++j) ar[j] = 1.f;
        } else {
#pragma unroll
            for (int j = 0; j < AE / 4; ++j) {
                const int m = m0 + rva + 64 * j;
                const float4 v = *reinterpret_cast<const float4*>(Wm + (size_t)m * g.K + k0 + 4 * qa + (BKK == 32 ? 16 * (j & 1) : 0) * 0);
                ar[4 * j] = v.x; ar[4 * j + 1] = v.y; ar[4 * j + 2] = v.z; ar[4 * j + 3] = v.w;
            }
        }
        const int r = k0 / g.C, c0 = k0 - r * g.C, kh = r / 3, kw = r - kh * 3;
        const int iy = py + kh - 1, ix = px + kw - 1;
        const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        const float* src = X + (size_t)(c0 + kb) * plane + (ok ? iy * g.W + ix : 0);
#pragma unroll
        for (int j = 0; j < BE; ++j) br[j] = (VAR == 1 || VAR == 2) ? 1.f : (ok ? src[(size_t)(KS * j) * plane] : 0.f);
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
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    gload(0); sstore(0); __syncthreads();
    int buf = 0;
    const int lr = lane & 31, lk = lane >> 5;
    for (int k0 = 0; k0 < g.K; k0 += BKK) {
        const bool more = k0 + BKK < g.K;
        if (more) gload(k0 + BKK);
        const float* ap = As(buf) + lk * LDA + wm * WM + lr;
        const float* bp = Bs(buf) + lk * LDB + wn * WN + lr;
#pragma unroll
        for (int kp = 0; kp < BKK / 2; ++kp) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = ap[(2 * kp) * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bp[(2 * kp) * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (VAR == 3) { acc[i][j][0] += a[i] * b[j]; }
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                }
        }
        if (VAR != 4) { if (more) sstore(buf ^ 1); __syncthreads(); buf ^= 1; }
        else { asm volatile("" :: "v"(ar[0]), "v"(br[0])); }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nn = n0 + wn * WN + j * 32 + lr;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + frag_row(r, lane);
                Y[(size_t)m * g.N + nn] = acc[i][j][r];
            }
    }
}

template <int BM, int BN, int BKK, int VAR>
void run(const char* name, G g, float* W, float* X, float* Y) {
    dim3 grid(g.N / BN, g.M / BM);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<BM, BN, BKK, VAR>), grid, dim3(256), 0, 0, W, X, Y, g);
    hipEventRecord(a);
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k<BM, BN, BKK, VAR>), grid, dim3(256), 0, 0, W, X, Y, g);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
    printf("  %-34s tile %3dx%3d BK %2d blocks %5d : %7.3f ms  %6.1f TF\n", name, BM, BN, BKK, grid.x * grid.y, ms, 2.0 * g.M * g.N * g.K / ms / 1e9);
}

int main() {
    struct { const char* n; int C, H, W, M; } L[] = {{"conv1_2", 64, 512, 1024, 64}, {"conv2_2", 128, 256, 512, 128}, {"conv3_2", 256, 128, 256, 256}, {"conv4_2", 512, 64, 128, 512}, {"conv5_x", 512, 32, 64, 512}};
    for (auto& l : L) {
        G g{l.C, l.H, l.W, l.M, l.H * l.W, l.C * 9};
        float *W, *X, *Y;
        hipMalloc(&W, (size_t)g.M * g.K * 4); hipMalloc(&X, (size_t)g.C * g.N * 4); hipMalloc(&Y, (size_t)g.M * g.N * 4);
        std::vector<float> h((size_t)g.C * g.N); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
        hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> hw((size_t)g.M * g.K); for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) % 1000) / 10000.f - 0.05f;
        hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        printf("%s  M=%d N=%d K=%d\n", l.n, g.M, g.N, g.K);
        if (g.M >= 128) {
            run<128, 128, 16, 0>("baseline", g, W, X, Y);
            run<128, 128, 16, 1>("no B gather", g, W, X, Y);
            run<128, 128, 16, 2>("no global loads", g, W, X, Y);
            run<128, 128, 16, 3>("no MFMA", g, W, X, Y);
            run<128, 128, 16, 4>("no LDS store/barrier", g, W, X, Y);
            run<128, 64, 16, 0>("tile 128x64", g, W, X, Y);
            run<64, 64, 16, 0>("tile 64x64", g, W, X, Y);
            run<64, 128, 16, 0>("tile 64x128", g, W, X, Y);
        } else {
            run<64, 128, 16, 0>("baseline 64x128", g, W, X, Y);
            run<64, 128, 16, 1>("no B gather", g, W, X, Y);
            run<64, 128, 16, 3>("no MFMA", g, W, X, Y);
            run<64, 64, 16, 0>("tile 64x64", g, W, X, Y);
        }
        hipFree(W); hipFree(X); hipFree(Y);
    }
    return 0;
}
