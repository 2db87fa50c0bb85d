// Micro-benchmark: the conv epilogue's store pattern.  A 64 x 256 output tile per workgroup (4 waves, each 64 rows x 64 pixels of an
// NCHW tensor [64, 512*1024]); variant A: as the MFMA accumulator layout dictates -- one dword per lane, a store instruction covers two
// rows x 32 pixels (2 x 128 B); variant B: 16 bytes per lane, an instruction covers 4 rows x 64 pixels (4 x 256 B), what an LDS
// transpose in front of the stores would give.   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int PLANE = 512 * 1024, M = 64;
__global__ __launch_bounds__(256) void store_dword(float *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, g = lane >> 5;
    const int n0 = blockIdx.x * 256 + wave * 64;
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 2; ++i)
            for (int r = 0; r < 16; ++r) {
                const int m = i * 32 + (r >> 2) * 8 + g * 4 + (r & 3);
                out[(size_t)m * PLANE + n0 + j * 32 + lr] = (float)(m + lane);
            }
}
__global__ __launch_bounds__(256) void store_x4(float *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n0 = blockIdx.x * 256 + wave * 64;
    for (int it = 0; it < 16; ++it) {
        const int m = it * 4 + (lane >> 4);
        float4 v = {(float)m, (float)lane, 0.f, 1.f};
        *reinterpret_cast<float4 *>(out + (size_t)m * PLANE + n0 + (lane & 15) * 4) = v;
    }
}
int main() {
    float *out; hipMalloc(&out, (size_t)M * PLANE * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 10; ++k) { if (v == 0) store_dword<<<PLANE / 256, 256>>>(out); else store_x4<<<PLANE / 256, 256>>>(out); }
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per 134 MB  (%.2f TB/s)\n", v == 0 ? "dword stores (accumulator layout)" : "16-byte stores", ms * 100, 134.2e6 / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
