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

// A HIP stream with a hardware queue of its own.  Ordinary streams of a process share at most four hardware queues
// (round-robin, per priority): a long latency-bound kernel such as furthest point sampling (milliseconds on 8
// workgroups) then stalls whatever stream happens to share its queue -- the training stream, a graph's internal
// branch, RCCL -- and which one that is changes with every stream anybody creates.  Streams created with a CU mask get
// a dedicated queue; the mask here enables every CU, so nothing else about the stream is special.
// first_cu / n_cus select the enabled bits [first_cu, first_cu + n_cus) of the CU mask (n_cus <= 0: every CU) -- a way to
// keep a background stream (neighbourhood geometry of the next batch) from spreading over the whole chip.
AMC_API int amc3d_stream_create_masked(void **stream, int first_cu, int n_cus)
{
    if (!stream) return amc::bad_arg("amc3d_stream_create_masked: null pointer");
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || cus <= 0) { amc::set_error("amc3d_stream_create_masked: %s", hipGetErrorString(e)); return (int)(e ? e : hipErrorUnknown); }
    if (n_cus <= 0) { first_cu = 0; n_cus = cus; }
    if (first_cu < 0 || first_cu + n_cus > cus) return amc::bad_arg("amc3d_stream_create_masked: CU range outside the device");
    uint32_t mask[32] = {0};
    const int words = (cus + 31) / 32;
    if (words > 32) return amc::bad_arg("amc3d_stream_create_masked: more than 1024 CUs");
    for (int c = first_cu; c < first_cu + n_cus; ++c) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t s = nullptr;
    e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) { amc::set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e)); return (int)e; }
    *stream = (void *)s;
    return 0;
}

AMC_API int amc3d_stream_create_dedicated(void **stream) { return amc3d_stream_create_masked(stream, 0, 0); }

// the same with an arbitrary mask: word i, bit j enables CU 32 i + j
AMC_API int amc3d_stream_create_cu_mask(void **stream, const unsigned int *mask, int words)
{
    if (!stream || !mask || words <= 0 || words > 32) return amc::bad_arg("amc3d_stream_create_cu_mask: bad argument");
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) { amc::set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e)); return (int)e; }
    *stream = (void *)s;
    return 0;
}

// Diagnostic for planning CU masks: out[b] = the XCD (HW_REG_XCC_ID, 0-7) workgroup b of an nblocks-wide launch on `stream`
// ran on.  Which mask bits belong to which XCD is not documented; a masked stream answers it.
namespace amc {
__global__ void probe_xcc_kernel(int *__restrict__ out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15);
    // stay resident for a moment so that the dispatcher has to spread the workgroups over everything the mask allows
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) {}
}
}  // namespace amc

AMC_API int amc3d_probe_xcc_ids(int nblocks, int *out, void *stream)
{
    if (nblocks <= 0) return 0;
    if (!out) return amc::bad_arg("amc3d_probe_xcc_ids: null pointer");
    hipLaunchKernelGGL(amc::probe_xcc_kernel, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, out);
    return amc::launch_status("amc3d_probe_xcc_ids");
}

// ---- scratch (private segment) high-water mark ---------------------------------------------------------------------------
// The runtime sizes the device's scratch memory by the largest per-lane request it has seen and re-allocates it when a kernel
// asks for more.  Graph nodes instantiated before such a re-allocation keep pointing at the old one (ROCm 7.2, measured: a
// hipGraph with scratch-using kernels -- rocprim's radix sort, 80 bytes per lane -- faults at its next replay once another
// stream has run, for the first time, a library kernel with a larger private segment: MIOpen's find mode at a new shape does).
// amc3d_reserve_scratch raises the mark BEFORE graphs are captured: one launch of a kernel with `bytes_per_lane` of private
// array (256, 1024, 4096 or 16384).
namespace amc {
template <int WORDS>
__global__ void scratch_reserve_kernel(int *__restrict__ out, int salt)
{
    volatile int a[WORDS];  // dynamically indexed: stays in scratch
    for (int i = 0; i < WORDS; ++i) a[i] = i * salt;
    int s = 0;
    for (int i = 0; i < WORDS; i += 97) s += a[(i + salt) % WORDS];
    if (s == 0x7fffffff) out[0] = s;  // (never: keeps the array alive)
}
}  // namespace amc

AMC_API int amc3d_reserve_scratch(int bytes_per_lane, int *scratch_out, void *stream)
{
    if (!scratch_out) return amc::bad_arg("amc3d_reserve_scratch: needs a device int to (never) write");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(256 * 8), block(256);  // every CU, full occupancy: the runtime sizes scratch for the resident waves
    if (bytes_per_lane <= 256) hipLaunchKernelGGL(amc::scratch_reserve_kernel<64>, grid, block, 0, st, scratch_out, 3);
    else if (bytes_per_lane <= 1024) hipLaunchKernelGGL(amc::scratch_reserve_kernel<256>, grid, block, 0, st, scratch_out, 3);
    else if (bytes_per_lane <= 4096) hipLaunchKernelGGL(amc::scratch_reserve_kernel<1024>, grid, block, 0, st, scratch_out, 3);
    else hipLaunchKernelGGL(amc::scratch_reserve_kernel<4096>, grid, block, 0, st, scratch_out, 3);
    return amc::launch_status("amc3d_reserve_scratch");
}

AMC_API int amc3d_stream_destroy(void *stream)
{
    if (!stream) return 0;
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) { amc::set_error("hipStreamDestroy: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}
