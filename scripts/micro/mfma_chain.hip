// Micro-benchmark: how fast does one wave issue DEPENDENT v_mfma_f32_32x32x2_f32 (one accumulator) against 2 / 4 independent ones,
// with 1, 2 or 4 such waves per SIMD?   hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void chain(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
static void run(int waves_per_simd) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 4000;   // one workgroup per CU, waves_per_simd waves on each SIMD
    float *out; hipMalloc(&out, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chain<NACC><<<blocks, threads>>>(out, 10, 1.f, 1.f);
    hipEventRecord(e0);
    chain<NACC><<<blocks, threads>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * 16 * 4096.0;
    printf("accumulators %d, waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", NACC, waves_per_simd, ms, flop / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 4; w *= 2) { run<1>(w); run<2>(w); run<4>(w); }
    return 0;
}
