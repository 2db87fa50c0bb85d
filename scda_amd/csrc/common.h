// common.h -- shared helpers for the gfx950 kernels of libscda_ops.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/scda_ops.h"

#define SCDA_API extern "C" __attribute__((visibility("default")))

namespace scda {

void set_error(const char *fmt, ...);

// launch epilogue: convert hipGetLastError() into the C-ABI status
static inline int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SCDA_ELAUNCH;
    }
    return SCDA_OK;
}

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// grid size for grid-stride elementwise kernels: enough blocks to fill
// 256 CUs x 8 blocks, never more than the work needs
static inline int ew_grid(long long n, int block = 256) {
    long long g = (n + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

constexpr int kWave = 64;  // gfx950 wavefront

// 4 KB of device zeros (per device, allocated on first use, never freed).  Gathers whose lane falls outside the
// tensor read from here instead of being predicated: with no select after the load there is nothing that needs the
// loaded value before the LDS store, so the s_waitcnt lands AFTER the MFMAs of the current K-slab.
const float *zero_page();

// ---- optional per-launch timing of the GEMM-class kernels (scda_prof_* in the C ABI) ----
// When enabled, a hipEvent pair is recorded on the launch stream around the tagged kernel;
// scda_prof_collect() (after a device sync) turns them into per-kernel count / time / flops.
// kernel classes mirror the template instantiations rocprofv3 lists:
//   conv_igemm_glds_kernel<BM,BN,KH,KW,S,DGRAD> (classes follow the INSTANTIATION that ran: tile rows 64, or 128|256): id = ((DGRAD*2 + (BM==64)) * 3 + shape), shape 0: 3x3 s1, 1: 3x3 s2, 2: 1x1
//   conv_wgrad_kernel<*,*,KH,KW,S>:          id = 12 + shape ;   gemm_kernel<*>: id = 15
//   conv_igemm_kernel<*> (the gather kernel of the layers whose channel count is not a multiple of 16): id = 16
//   conv_wino_kernel (Winograd F(2x2,3x3), conv_wino.hip): forward id = 17, data gradient id = 18, weight gradient id = 19
enum { PK_CONV = 0, PK_CONV_WGRAD = 12, PK_GEMM = 15, PK_CONV_GATHER = 16, PK_WINO_FWD = 17, PK_WINO_DGRAD = 18, PK_WINO_WGRAD = 19, PK_COUNT = 20 };
static inline int prof_shape(int KH, int S) { return KH == 1 ? 2 : (S == 2 ? 1 : 0); }
void prof_begin(int kernel, double flops, hipStream_t st, double bytes = 0.0);   // bytes: algorithmic HBM bytes of the launch
void prof_end(hipStream_t st);

}  // namespace scda
