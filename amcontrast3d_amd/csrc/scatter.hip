// Scatter-add gradients of grouping / three_interpolate for gfx950.
//
// Reference: group_points_grad_kernel_fast (pointnet2_batch/src/group_points_gpu.cu:14-31) and
// three_interpolate_grad_kernel_fast (interpolate_gpu.cu:127-149): one fp32 atomicAdd per
// (b, c, entry) into the channel-major (B,C,N) gradient -- 64 lanes of a wave hit 64 unrelated
// addresses of one channel plane, the slowest shape for global float atomics on MI355X
// (~0.08 TB/s, MI355X_MICROARCH.md "Global float atomics").
//
// Here the adds go to a POINT-major scratch (B,N,C): a tile of 64 entries x 64 channels of grad_out
// is read coalesced (along the entry axis) into LDS, transposed there, and every atomic
// wave-instruction then adds 64 contiguous floats (or 2 x 32) of ONE destination row -- the shape
// that runs at the full atomic rate.  A second pass transposes the scratch into the (B,C,N)
// layout the reference returns.  Summation order is unspecified, as in the reference.
#include "common.h"

namespace amc {

constexpr int SC_TILE = 64;  // entries per workgroup tile

// entries are the flattened (p, s) positions of one batch; every entry has `fan` destinations
// (1 for grouping, 3 for interpolation) with optional weights
template <int FAN>
__global__ __launch_bounds__(256) void scatter_pm_kernel(int c, int n, int entries, const float *__restrict__ grad_out,
                                                         const int *__restrict__ idx, const float *__restrict__ weight,
                                                         float *__restrict__ scratch)
{
    __shared__ float tile[64][SC_TILE + 1];
    __shared__ int s_idx[SC_TILE * FAN];
    __shared__ float s_w[SC_TILE * FAN];
    const int bs = blockIdx.y;
    const int t0 = blockIdx.x * SC_TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int te = min(SC_TILE, entries - t0);
    for (int i = threadIdx.x; i < te * FAN; i += 256) {
        s_idx[i] = idx[((size_t)bs * entries + t0) * FAN + i];
        s_w[i] = weight ? weight[((size_t)bs * entries + t0) * FAN + i] : 1.f;
    }
    float *dst = scratch + (size_t)bs * n * c;
    for (int c0 = 0; c0 < c; c0 += 64) {
        const int cc = min(64, c - c0);
        __syncthreads();
        // coalesced read: one wave reads 64 consecutive entries of one channel plane
        for (int ch = wave; ch < cc; ch += 4)
            tile[ch][lane] = lane < te ? grad_out[((size_t)bs * c + c0 + ch) * entries + t0 + lane] : 0.f;
        __syncthreads();
        if (cc > 32) {
            // one entry per wave-instruction: lanes = channels
            for (int e = wave; e < te; e += 4) {
                if (lane < cc) {
                    const float g = tile[lane][e];
#pragma unroll
                    for (int f = 0; f < FAN; ++f)
                        atomicAdd(dst + (size_t)s_idx[e * FAN + f] * c + c0 + lane, FAN == 1 ? g : __fmul_rn(g, s_w[e * FAN + f]));
                }
            }
        } else {
            // two entries per wave-instruction: 2 x 32 contiguous floats
            const int half = lane >> 5, ch = lane & 31;
            for (int e = wave * 2 + half; e < te; e += 8) {
                if (ch < cc) {
                    const float g = tile[ch][e];
#pragma unroll
                    for (int f = 0; f < FAN; ++f)
                        atomicAdd(dst + (size_t)s_idx[e * FAN + f] * c + c0 + ch, FAN == 1 ? g : __fmul_rn(g, s_w[e * FAN + f]));
                }
            }
        }
    }
}

// (B,N,C) -> (B,C,N), accumulating into grad_points (which the caller zero-initialised)
__global__ __launch_bounds__(256) void transpose_add_kernel(int c, int n, const float *__restrict__ scratch,
                                                            float *__restrict__ grad_points)
{
    __shared__ float tile[64][65];
    const int bs = blockIdx.z;
    const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < 64; r += 4) {  // r: point within tile, lane: channel
        const int nn = n0 + r, ch = c0 + lane;
        tile[r][lane] = (nn < n && ch < c) ? scratch[((size_t)bs * n + nn) * c + ch] : 0.f;
    }
    __syncthreads();
    for (int r = wave; r < 64; r += 4) {  // r: channel within tile, lane: point
        const int ch = c0 + r, nn = n0 + lane;
        if (ch < c && nn < n) grad_points[((size_t)bs * c + ch) * n + nn] += tile[lane][r];
    }
}

int scatter_add_pm(int fan, int b, int c, int n, long entries, const float *grad_out, const int *idx,
                   const float *weight, float *grad_points, float *scratch, hipStream_t stream, const char *what)
{
    if (int st = fill_i32((int *)scratch, 0, (size_t)b * n * c, stream)) return st;
    dim3 grid(div_up(entries, SC_TILE), b);
    if (fan == 1)
        hipLaunchKernelGGL(scatter_pm_kernel<1>, grid, dim3(256), 0, stream, c, n, (int)entries, grad_out, idx, weight, scratch);
    else
        hipLaunchKernelGGL(scatter_pm_kernel<3>, grid, dim3(256), 0, stream, c, n, (int)entries, grad_out, idx, weight, scratch);
    hipLaunchKernelGGL(transpose_add_kernel, dim3(div_up(n, 64), div_up(c, 64), b), dim3(256), 0, stream, c, n, scratch,
                       grad_points);
    return launch_status(what);
}

}  // namespace amc
