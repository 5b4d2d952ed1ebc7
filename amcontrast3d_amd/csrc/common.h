// Shared helpers for the gfx950 kernels (wave64 only; no dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/amc3d.h"

#define AMC_API extern "C" __attribute__((visibility("default")))

namespace amc {

constexpr int kWave = 64;

void set_error(const char *fmt, ...);

// launch epilogue: pick up a launch-configuration error without synchronising
inline int launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

inline int bad_arg(const char *what)
{
    set_error("%s", what);
    return (int)hipErrorInvalidValue;
}

inline int div_up(long a, long b) { return (int)((a + b - 1) / b); }

// The reference's squared distance, evaluated as written (no contraction):
// (qx-x)*(qx-x) + (qy-y)*(qy-y) + (qz-z)*(qz-z)
__device__ __forceinline__ float dist2_ref(float qx, float qy, float qz, float x, float y, float z)
{
    const float dx = __fsub_rn(qx, x), dy = __fsub_rn(qy, y), dz = __fsub_rn(qz, z);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// Zero-fill as an ordinary kernel.  The library never uses hipMemsetAsync: under hipGraph replay a
// memset node was observed to race with the kernels captured after it on ROCm 7.2 (stale counters ->
// out-of-range scatter), whereas kernel nodes of one captured stream always run in order.
__global__ void fill_i32_kernel(int *__restrict__ p, int value, size_t count);
int fill_i32(int *p, int value, size_t count, hipStream_t stream);

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// number of set bits of `mask` below this lane
__device__ __forceinline__ int mbcnt(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

}  // namespace amc
