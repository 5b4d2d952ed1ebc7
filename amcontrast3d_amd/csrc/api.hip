// Library identification and per-thread error text for the C-ABI (include/amc3d.h).
#include <stdarg.h>

#include "common.h"

namespace amc {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
__global__ void fill_i32_kernel(int *__restrict__ p, int value, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        p[i] = value;
}

int fill_i32(int *p, int value, size_t count, hipStream_t stream)
{
    if (count == 0) return 0;
    const int blocks = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(blocks), dim3(256), 0, stream, p, value, count);
    return launch_status("fill_i32");
}
}  // namespace amc

AMC_API const char *amc3d_version(void) { return "amc3d-hip gfx950 1"; }
AMC_API const char *amc3d_last_error(void) { return amc::g_err; }
