// Input pipeline of the S3DIS / ScanNet loaders on the device:
//   voxelize   openpoints/dataset/data_util.py:92-141  floor(coord / voxel) -> FNV-1a 64-bit hash of the three cell
//              coordinates -> argsort -> unique / counts (one random or every point per voxel)
//   crop_pc    :146-174  the voxel_max points nearest to a seed point: argsort of the squared distances
// The reference does this in numpy inside 6 loader workers; at > 18 M points/s per GPU that is the bottleneck.
// Here: one kernel for the keys, a stable LSD radix sort of (key, index) pairs (hipCUB / rocPRIM device primitive --
// a plain library sort, as rocBLAS is for plain GEMMs), run-length kernels for voxel ids / starts / counts, and the
// same for the crop with the distance bits as keys.  numpy's argsort is NOT stable: which order the points of one
// voxel (or two equidistant points) come in is unspecified by the reference; a stable order is one valid choice.
// Arithmetic as numpy does it under numpy 2 (the container's): coord (float32) / np.array(voxel_size) is a float64
// division (a 0-d array is not a weak scalar), floor in float64, cast to uint64; squared distances in float32 without
// contraction, ((dx^2 + dy^2) + dz^2).
#include "cub_kernel_memset.h"  // hipCUB with its memsets as kernels (graph-safe)

#include "common.h"

namespace amc {

__global__ void voxel_key_kernel(int n, const float *__restrict__ coord, double voxel, unsigned long long *__restrict__ key,
                                 int *__restrict__ iota)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long h = 14695981039346656037ULL;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double c = floor((double)coord[(size_t)i * 3 + j] / voxel);
        h *= 1099511628211ULL;
        h ^= (unsigned long long)(long long)c;  // non-negative by contract (the cloud is shifted to its min corner first)
    }
    key[i] = h;
    iota[i] = i;
}

// head[i] = 1 where a new voxel starts in the sorted key sequence
__global__ void voxel_head_kernel(int n, const unsigned long long *__restrict__ ks, int *__restrict__ head)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || ks[i] != ks[i - 1]) ? 1 : 0;
}

// incl = inclusive scan of head: voxel id of sorted position i is incl[i] - 1
__global__ void voxel_start_kernel(int n, const int *__restrict__ head, const int *__restrict__ incl, int *__restrict__ voxel_idx,
                                   int *__restrict__ start, int *__restrict__ nvox)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = incl[i] - 1;
    voxel_idx[i] = v;
    if (head[i]) start[v] = i;
    if (i == n - 1) { *nvox = v + 1; start[v + 1] = n; }
}

__global__ void voxel_count_kernel(int n, const int *__restrict__ nvox, const int *__restrict__ start, int *__restrict__ count)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) count[v] = v < *nvox ? start[v + 1] - start[v] : 0;
}

__global__ void voxel_select_kernel(int nvox, const int *__restrict__ start, const int *__restrict__ count,
                                    const int *__restrict__ idx_sort, const int *__restrict__ rnd, int *__restrict__ out)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nvox) out[v] = idx_sort[start[v] + rnd[v] % count[v]];
}

__global__ void crop_d2_kernel(int n, const float *__restrict__ coord, int init, float *__restrict__ d2, unsigned *__restrict__ bits,
                               int *__restrict__ iota)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dx = __fsub_rn(coord[(size_t)i * 3], coord[(size_t)init * 3]);
    const float dy = __fsub_rn(coord[(size_t)i * 3 + 1], coord[(size_t)init * 3 + 1]);
    const float dz = __fsub_rn(coord[(size_t)i * 3 + 2], coord[(size_t)init * 3 + 2]);
    const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    d2[i] = d;
    bits[i] = __float_as_uint(d);  // non-negative floats order like their bit patterns
    iota[i] = i;
}

__global__ void copy_i32_kernel(int n, const int *__restrict__ src, int *__restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t sort64_temp(int n)
{
    size_t t = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                       (const int *)nullptr, (int *)nullptr, n);
    size_t s = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, s, (const int *)nullptr, (int *)nullptr, n);
    return align256(t > s ? t : s);
}

static size_t sort32_temp(int n)
{
    size_t t = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t, (const unsigned *)nullptr, (unsigned *)nullptr, (const int *)nullptr,
                                       (int *)nullptr, n);
    return align256(t);
}

}  // namespace amc

using namespace amc;

AMC_API size_t amc3d_voxelize_workspace_bytes(int n)
{
    if (n <= 0) return 0;
    // sorted keys | iota | head | incl | sort / scan temp
    return align256((size_t)n * 8) + 3 * align256((size_t)n * 4) + sort64_temp(n) + 256;
}

// coord (n,3) fp32, shifted to its min corner.  key (n): the FNV-1a hash of every point's cell (as fnv_hash_vec);
// idx_sort (n): point indices ordered by key (stable); voxel_idx (n): voxel id of sorted position i; start (n+1):
// first sorted position of every voxel (start[nvox] = n); count (n): points per voxel (0 beyond nvox); nvox (1).
AMC_API int amc3d_voxelize(int n, const float *coord, double voxel_size, unsigned long long *key, int *idx_sort, int *voxel_idx,
                           int *start, int *count, int *nvox, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (n <= 0) return 0;
    if (!coord || !(voxel_size > 0.0) || !key || !idx_sort || !voxel_idx || !start || !count || !nvox || !workspace ||
        workspace_bytes < amc3d_voxelize_workspace_bytes(n))
        return bad_arg("amc3d_voxelize: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    char *w = (char *)workspace;
    unsigned long long *ks = (unsigned long long *)w; w += align256((size_t)n * 8);
    int *iota = (int *)w; w += align256((size_t)n * 4);
    int *head = (int *)w; w += align256((size_t)n * 4);
    int *incl = (int *)w; w += align256((size_t)n * 4);
    size_t temp = sort64_temp(n);
    const int blocks = div_up(n, 256);
    hipLaunchKernelGGL(voxel_key_kernel, dim3(blocks), dim3(256), 0, stream, n, coord, voxel_size, key, iota);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(w, temp, (const unsigned long long *)key, ks, (const int *)iota, idx_sort, n, 0, 64, stream);
    if (e != hipSuccess) { set_error("amc3d_voxelize: radix sort: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(voxel_head_kernel, dim3(blocks), dim3(256), 0, stream, n, (const unsigned long long *)ks, head);
    temp = sort64_temp(n);
    e = hipcub::DeviceScan::InclusiveSum(w, temp, (const int *)head, incl, n, stream);
    if (e != hipSuccess) { set_error("amc3d_voxelize: scan: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(voxel_start_kernel, dim3(blocks), dim3(256), 0, stream, n, (const int *)head, (const int *)incl, voxel_idx, start,
                       nvox);
    hipLaunchKernelGGL(voxel_count_kernel, dim3(blocks), dim3(256), 0, stream, n, (const int *)nvox, (const int *)start, count);
    return launch_status("amc3d_voxelize");
}

// train mode of voxelize (data_util.py:136-140): one point per voxel, idx_unique[v] = idx_sort[start[v] + rnd[v] % count[v]]
AMC_API int amc3d_voxel_select(int nvox, const int *start, const int *count, const int *idx_sort, const int *rnd, int *idx_unique,
                               void *stream)
{
    if (nvox <= 0) return 0;
    if (!start || !count || !idx_sort || !rnd || !idx_unique) return bad_arg("amc3d_voxel_select: null pointer");
    hipLaunchKernelGGL(voxel_select_kernel, dim3(div_up(nvox, 256)), dim3(256), 0, (hipStream_t)stream, nvox, start, count, idx_sort,
                       rnd, idx_unique);
    return launch_status("amc3d_voxel_select");
}

AMC_API size_t amc3d_crop_nearest_workspace_bytes(int n)
{
    if (n <= 0) return 0;
    return 4 * align256((size_t)n * 4) + sort32_temp(n) + 256;  // bits | sorted bits | iota | sorted idx | temp
}

// crop_pc's nearest-N crop (data_util.py:157-160): d2 (n) = squared distance of every point to coord[init_idx];
// crop_idx (keep) = the keep nearest points in ascending distance (stable among equal distances)
AMC_API int amc3d_crop_nearest(int n, const float *coord, int init_idx, int keep, float *d2, int *crop_idx, void *workspace,
                               size_t workspace_bytes, void *stream_)
{
    if (n <= 0 || keep <= 0) return 0;
    if (!coord || init_idx < 0 || init_idx >= n || keep > n || !d2 || !crop_idx || !workspace ||
        workspace_bytes < amc3d_crop_nearest_workspace_bytes(n))
        return bad_arg("amc3d_crop_nearest: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    char *w = (char *)workspace;
    unsigned *bits = (unsigned *)w; w += align256((size_t)n * 4);
    unsigned *sbits = (unsigned *)w; w += align256((size_t)n * 4);
    int *iota = (int *)w; w += align256((size_t)n * 4);
    int *sidx = (int *)w; w += align256((size_t)n * 4);
    size_t temp = sort32_temp(n);
    hipLaunchKernelGGL(crop_d2_kernel, dim3(div_up(n, 256)), dim3(256), 0, stream, n, coord, init_idx, d2, bits, iota);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(w, temp, (const unsigned *)bits, sbits, (const int *)iota, sidx, n, 0, 32, stream);
    if (e != hipSuccess) { set_error("amc3d_crop_nearest: radix sort: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(copy_i32_kernel, dim3(div_up(keep, 256)), dim3(256), 0, stream, keep, (const int *)sidx, crop_idx);
    return launch_status("amc3d_crop_nearest");
}
