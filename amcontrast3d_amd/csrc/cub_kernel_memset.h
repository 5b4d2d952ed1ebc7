// hipCUB / rocPRIM with their hipMemsetAsync calls turned into a fill KERNEL.
//
// rocPRIM's radix sort zero-fills its digit offsets and look-back states with hipMemsetAsync (device_radix_sort.hpp).  Inside a
// captured graph those become memset NODES, and on ROCm 7.2 a memset node is not reliably ordered before the kernel nodes that
// follow it at replay (the reason this library's own zero-fills are kernels).  Measured in round 3: a graph holding one
// DeviceRadixSort::SortPairs replays correctly for thousands of steps -- and faults ("write access to a read-only page") at the
// first replay after other streams have run kernels the process had not run before (a first validation pass between two
// training epochs): the sort scattered through offsets that were not zero yet.  rocPRIM is header-only, so the call is
// redirected here for the translation units that instantiate it.  Include this INSTEAD of <hipcub/hipcub.hpp>.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amc {
static __global__ void cub_fill_bytes_kernel(unsigned char *__restrict__ p, int value, size_t bytes)
{
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= bytes && (((uintptr_t)p) & 3) == 0) {
        const unsigned v = (unsigned)(value & 0xff) * 0x01010101u;
        *reinterpret_cast<unsigned *>(p + i) = v;
    } else {
        for (size_t j = i; j < bytes && j < i + 4; ++j) p[j] = (unsigned char)value;
    }
}

static inline hipError_t memset_as_kernel(void *dst, int value, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    const size_t threads = (bytes + 3) / 4;
    hipLaunchKernelGGL(cub_fill_bytes_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, (unsigned char *)dst, value,
                       bytes);
    return hipGetLastError();
}
}  // namespace amc

#define hipMemsetAsync(dst, value, bytes, stream) ::amc::memset_as_kernel((dst), (value), (bytes), (stream))
#include <hipcub/hipcub.hpp>
#undef hipMemsetAsync
