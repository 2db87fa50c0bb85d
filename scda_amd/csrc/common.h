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

}  // namespace scda
