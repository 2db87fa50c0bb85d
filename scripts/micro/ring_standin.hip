// ring_standin.hip -- measurement aid for scripts/allreduce_contention.py (not part of the product): a PERSISTENT streaming kernel
// shaped like a collective's ring kernels -- a fixed, small number of workgroups (one per "channel") that stay resident for the
// whole transfer and stream a buffer through HBM -- instead of an element-wise torch kernel, whose ~100 k short workgroups behave
// nothing like RCCL beside this library's one-workgroup-per-CU convolution launches.
//   heavy = 0: 512 threads, no LDS: fits on a CU beside a Winograd workgroup (128 KB LDS, 83 % of the registers)
//   heavy = 1: declares 64 KB of LDS: cannot share a CU with one (RCCL's kernels at ~128 registers per lane cannot either)
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/micro/ring_standin.hip -o scripts/micro/build/libring_standin.so
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int HEAVY>
__global__ __launch_bounds__(512) void standin_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4, int passes) {
    __shared__ float pad[HEAVY ? 16384 : 1];
    if (HEAVY && threadIdx.x == 0) pad[blockIdx.x & 16383] = 1.f;
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    for (int p = 0; p < passes; ++p)
        for (size_t i = lo + threadIdx.x; i < hi; i += 4 * 512) {       // four 16-byte loads in flight per lane
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + u * 512 < hi) v[u] = src[i + u * 512];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + u * 512 < hi) dst[i + u * 512] = v[u];
        }
    if (HEAVY && pad[0] == 123.f) dst[0].x = 0.f;
}

extern "C" int standin_launch(const void *src, void *dst, size_t floats, int blocks, int passes, int heavy, void *stream) {
    if (heavy) hipLaunchKernelGGL(standin_kernel<1>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const float4 *)src, (float4 *)dst, floats / 4, passes);
    else hipLaunchKernelGGL(standin_kernel<0>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const float4 *)src, (float4 *)dst, floats / 4, passes);
    return (int)hipGetLastError();
}
