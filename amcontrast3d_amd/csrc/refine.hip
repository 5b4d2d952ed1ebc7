// Masked refinement of AMContrast3D++ (the DualMasks rule with fusion 'MIN') as two launches per direction.
//
// Reference: openpoints/AMContrast3D/MaskedRefine.py:55-131 (RefinementMethod.DualMasks / cross_mask / self_mask), called per
// decoder level by openpoints/models/backbone/pointnext_MM.py:541-560:
//     neighbor_ambiguity = a_rows[neighbor_idx]                                 # (m, K-1, 1)
//     cross  = rows of f at the neighbour with the SMALLEST predicted ambiguity # one-hot x gather x sum over K-1 there
//     mask   = (a <= threshold_max) * (a >= threshold);  rate = count_nonzero(mask) / numel * 100
//     f_new  = f * ~mask + cross * mask;   f = gamma * f_new + (1 - gamma) * f
// with two reinterpretations that are part of the reference's behaviour and kept: the (B, D, n) feature tensor is VIEWED as
// (B*n, D) rows for the gather (row r = the D floats at flat offset r*D) and the gathered rows are viewed back as (B, D, n),
// while the mask (B, 1, n) is broadcast over D in the (B, D, n) indexing.  As tensor operations that is ~25 launches per level
// (index_select x 2, min, gather, comparisons, casts, five elementwise passes over the feature tensor, count_nonzero; backward:
// the same passes and an index_add); here:
//   refine_select_kernel    best[r] = the neighbour of row r with the smallest a (first on ties), mask[r], per-block counts
//   refine_combine_kernel   out = gamma * (f * ~mask + f_rows[best] * mask) + (1 - gamma) * f, elementwise in flat order (the
//                           same fp32 operations in the same order as the tensor expression); block 0 sums the counts
//   refine_backward_*       df = (gamma * dout) * ~mask + (1 - gamma) * dout, then the masked elements' gamma * dout added to
//                           row best[r] (float atomics, as torch's index_add)
#include "common.h"

namespace amc {

__global__ __launch_bounds__(256) void refine_select_kernel(int m, int k, int nbr_stride, const float *__restrict__ a,
                                                            const int *__restrict__ nbr, float thr, float thr_max,
                                                            int *__restrict__ best, unsigned char *__restrict__ mask,
                                                            int *__restrict__ block_count)
{
    __shared__ int s_cnt[4];
    const int r = blockIdx.x * 256 + threadIdx.x;
    bool msk = false;
    if (r < m) {
        int nb[16];
        float av[16];
        int bi = nbr[(size_t)r * nbr_stride];
        float bv = a[bi];
        for (int j0 = 1; j0 < k; j0 += 16) {  // up to 16 neighbours' loads in flight
#pragma unroll
            for (int u = 0; u < 16; ++u) nb[u] = nbr[(size_t)r * nbr_stride + min(j0 + u, k - 1)];
#pragma unroll
            for (int u = 0; u < 16; ++u) av[u] = a[nb[u]];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (j0 + u < k && av[u] < bv) { bv = av[u]; bi = nb[u]; }  // strict <: the first minimum wins (torch.min)
        }
        best[r] = bi;
        const float ar = a[r];
        msk = ar <= thr_max && ar >= thr;
        mask[r] = msk ? 1 : 0;
    }
    const int c = (int)__popcll(__ballot(msk));
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
}

__device__ __forceinline__ float refine_one(float f, float cross, bool msk, float gamma, float one_minus_gamma)
{
    // f * ~mask + cross * mask, then gamma * f_new + (1 - gamma) * f: the tensor expression's operations, unfused
    const float fnew = __fadd_rn(__fmul_rn(f, msk ? 0.f : 1.f), __fmul_rn(cross, msk ? 1.f : 0.f));
    return __fadd_rn(__fmul_rn(gamma, fnew), __fmul_rn(one_minus_gamma, f));
}

// one thread per 4 consecutive floats when D % 4 == 0 and n % 4 == 0 (same row of the (B*n, D) view, same cloud), else per float
__global__ __launch_bounds__(256) void refine_combine_kernel(long total, int D, int n, int vec, float gamma, float one_minus_gamma,
                                                             const float *__restrict__ f, const int *__restrict__ best,
                                                             const unsigned char *__restrict__ mask, float *__restrict__ out,
                                                             const int *__restrict__ block_count, int nblocks,
                                                             int *__restrict__ count)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && count) {
        int s = 0;
        for (int i = 0; i < nblocks; ++i) s += block_count[i];
        count[0] = s;
    }
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        const long i = t * 4;
        if (i >= total) return;
        const long row = i / D;                // row of the (B*n, D) view
        const int d = (int)(i - row * D);
        const long bd = i / n;                 // b * D + d' of the (B, D, n) indexing
        const int j = (int)(i - bd * n);
        const long bn = (bd / D) * n + j;      // b * n + j: the mask element
        const float4 fv = *(const float4 *)(f + i);
        const float4 cv = *(const float4 *)(f + (size_t)best[row] * D + d);
        const uchar4 mv = *(const uchar4 *)(mask + bn);
        float4 o;
        o.x = refine_one(fv.x, cv.x, mv.x != 0, gamma, one_minus_gamma);
        o.y = refine_one(fv.y, cv.y, mv.y != 0, gamma, one_minus_gamma);
        o.z = refine_one(fv.z, cv.z, mv.z != 0, gamma, one_minus_gamma);
        o.w = refine_one(fv.w, cv.w, mv.w != 0, gamma, one_minus_gamma);
        *(float4 *)(out + i) = o;
        return;
    }
    if (t >= total) return;
    const long row = t / D;
    const int d = (int)(t - row * D);
    const long bd = t / n;
    const long bn = (bd / D) * n + (t - bd * n);
    out[t] = refine_one(f[t], f[(size_t)best[row] * D + d], mask[bn] != 0, gamma, one_minus_gamma);
}

// df = (gamma * dout) * ~mask + (1 - gamma) * dout
__global__ __launch_bounds__(256) void refine_backward_direct_kernel(long total, int D, int n, float gamma, float one_minus_gamma,
                                                                     const float *__restrict__ dout,
                                                                     const unsigned char *__restrict__ mask, float *__restrict__ df)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long bd = t / n;
    const long bn = (bd / D) * n + (t - bd * n);
    const float g = dout[t];
    df[t] = __fadd_rn(__fmul_rn(__fmul_rn(gamma, g), mask[bn] ? 0.f : 1.f), __fmul_rn(one_minus_gamma, g));
}

// the masked elements' share goes to the row their value came from: df_rows[best[row]][d] += gamma * dout
__global__ __launch_bounds__(256) void refine_backward_scatter_kernel(long total, int D, int n, float gamma,
                                                                      const float *__restrict__ dout, const int *__restrict__ best,
                                                                      const unsigned char *__restrict__ mask, float *__restrict__ df)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long bd = t / n;
    const long bn = (bd / D) * n + (t - bd * n);
    if (!mask[bn]) return;
    const long row = t / D;
    const int d = (int)(t - row * D);
    atomicAdd(df + (size_t)best[row] * D + d, __fmul_rn(gamma, dout[t]));
}

}  // namespace amc

using namespace amc;

AMC_API size_t amc3d_masked_refine_workspace_ints(int m) { return m > 0 ? (size_t)div_up(m, 256) : 0; }

// f, out (B, D, n) fp32; a (B*n) fp32 predicted ambiguity; nbr (B*n, k) int32 with row stride nbr_stride (the k-NN lists with
// the self match dropped); best (B*n) int32 and mask (B*n) bytes are outputs kept for the backward; count (1) int32 = the
// number of refined points; workspace: amc3d_masked_refine_workspace_ints(B*n) ints
AMC_API int amc3d_masked_refine_forward(int B, int D, int n, int k, int nbr_stride, const float *f, const float *a, const int *nbr,
                                        float threshold, float threshold_max, float gamma, float *out, int *best,
                                        unsigned char *mask, int *count, int *workspace, void *stream_)
{
    if (B <= 0 || D <= 0 || n <= 0) return 0;
    const long m = (long)B * n, total = m * D;
    if (k <= 0 || nbr_stride < k || m >= (1L << 31) || !f || !a || !nbr || !out || !best || !mask || !count || !workspace)
        return bad_arg("amc3d_masked_refine_forward: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const int nblocks = div_up(m, 256);
    hipLaunchKernelGGL(refine_select_kernel, dim3(nblocks), dim3(256), 0, stream, (int)m, k, nbr_stride, a, nbr, threshold,
                       threshold_max, best, mask, workspace);
    const int vec = D % 4 == 0 && n % 4 == 0 && ((((uintptr_t)f) | ((uintptr_t)out)) & 15) == 0 && (((uintptr_t)mask) & 3) == 0;
    const long threads = vec ? total / 4 : total;
    hipLaunchKernelGGL(refine_combine_kernel, dim3(div_up(threads, 256)), dim3(256), 0, stream, total, D, n, vec, gamma,
                       (float)(1.0 - (double)gamma), f, (const int *)best, (const unsigned char *)mask, out,
                       (const int *)workspace, nblocks, count);
    return launch_status("amc3d_masked_refine_forward");
}

AMC_API int amc3d_masked_refine_backward(int B, int D, int n, float gamma, const float *dout, const int *best,
                                         const unsigned char *mask, float *df, void *stream_)
{
    if (B <= 0 || D <= 0 || n <= 0) return 0;
    const long total = (long)B * n * D;
    if (!dout || !best || !mask || !df) return bad_arg("amc3d_masked_refine_backward: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(refine_backward_direct_kernel, dim3(div_up(total, 256)), dim3(256), 0, stream, total, D, n, gamma,
                       (float)(1.0 - (double)gamma), dout, mask, df);
    hipLaunchKernelGGL(refine_backward_scatter_kernel, dim3(div_up(total, 256)), dim3(256), 0, stream, total, D, n, gamma, dout,
                       best, mask, df);
    return launch_status("amc3d_masked_refine_backward");
}
