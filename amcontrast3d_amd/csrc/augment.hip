// Training-time augmentation of a batch of cropped clouds on the device (gfx950).
//
// The reference applies its transform chain per cloud in the loader's worker processes, ~10 tensor ops on a
// 24000-point cloud each (cfgs/s3dis/default.yaml:33-43: ChromaticAutoContrast, PointsToTensor, PointCloudScaling,
// PointCloudXYZAlign, PointCloudRotation, PointCloudJitter, ChromaticDropGPU, ChromaticNormalize --
// openpoints/transforms/point_transform_cpu.py:192-209, point_transformer_gpu.py:70-89,135-164,216-229,267-311,373-409).
// Here the whole batch is two launches: per-cloud statistics (colour extrema, the mean of the scaled coordinates, the
// extrema of the gravity coordinate), then one elementwise pass.  Random draws are the caller's (per-cloud table + the
// jitter noise): the library has no generator of its own.
//
// Per-cloud parameter record (AugmentCloud, 24 floats):
//   contrast (0/1), blend, scale[3], rot[9] (row-major R: pos' = pos @ R^T), drop (0/1), pad
#include "common.h"

namespace amc {

struct AugmentCloud { float contrast, blend, scale[3], rot[9], drop, pad[9]; };  // 24 floats
struct AugmentStats { float lo[3], hi[3], mean[3], zmin, cmax, pad[5]; };          // 16 floats

// one workgroup per cloud: fixed-order reductions (deterministic)
__global__ __launch_bounds__(1024) void augment_stats_kernel(int n, int g, const float *__restrict__ pos,
                                                             const float *__restrict__ color,
                                                             const AugmentCloud *__restrict__ par, AugmentStats *__restrict__ st)
{
    __shared__ float s_lo[16][3], s_hi[16][3], s_zlo[16], s_zhi[16];
    __shared__ double s_sum[16][3];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const AugmentCloud p = par[b];
    const float *P = pos + (size_t)b * n * 3, *C = color + (size_t)b * n * 3;
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f}, zlo = 3.4e38f, zhi = -3.4e38f;
    double sum[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < n; i += 1024) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = C[(size_t)i * 3 + c];
            lo[c] = fminf(lo[c], v);
            hi[c] = fmaxf(hi[c], v);
            sum[c] += (double)__fmul_rn(P[(size_t)i * 3 + c], p.scale[c]);
        }
        const float z = P[(size_t)i * 3 + g];
        zlo = fminf(zlo, z);
        zhi = fmaxf(zhi, z);
    }
    for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], d, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], d, 64));
            sum[c] += __shfl_xor(sum[c], d, 64);
        }
        zlo = fminf(zlo, __shfl_xor(zlo, d, 64));
        zhi = fmaxf(zhi, __shfl_xor(zhi, d, 64));
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { s_lo[wave][c] = lo[c]; s_hi[wave][c] = hi[c]; s_sum[wave][c] = sum[c]; }
        s_zlo[wave] = zlo; s_zhi[wave] = zhi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        AugmentStats o;
        double tot[3] = {0.0, 0.0, 0.0};
        float l[3] = {3.4e38f, 3.4e38f, 3.4e38f}, h[3] = {-3.4e38f, -3.4e38f, -3.4e38f}, zl = 3.4e38f, zh = -3.4e38f;
        for (int w = 0; w < 16; ++w) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { l[c] = fminf(l[c], s_lo[w][c]); h[c] = fmaxf(h[c], s_hi[w][c]); tot[c] += s_sum[w][c]; }
            zl = fminf(zl, s_zlo[w]); zh = fmaxf(zh, s_zhi[w]);
        }
        float cmax = -3.4e38f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o.lo[c] = l[c]; o.hi[c] = h[c];
            o.mean[c] = (float)(tot[c] / (double)n);
            // the largest colour AFTER auto-contrast (every step is monotone in the colour under rounding): at x = hi
            float v = h[c];
            if (p.contrast != 0.f) {
                const float sc = __fdiv_rn(255.f, __fsub_rn(h[c], l[c]));
                v = __fadd_rn(__fmul_rn(__fsub_rn(1.f, p.blend), h[c]), __fmul_rn(p.blend, __fmul_rn(__fsub_rn(h[c], l[c]), sc)));
            }
            cmax = fmaxf(cmax, v);
        }
        o.cmax = p.drop != 0.f ? 0.f : cmax;
        // min over the points of fl(fl(z * s) - mean): monotone in z, so it sits at an extremum of z
        const float sg = p.scale[g];
        const float a = __fsub_rn(__fmul_rn(zl, sg), o.mean[g]), bb = __fsub_rn(__fmul_rn(zh, sg), o.mean[g]);
        o.zmin = fminf(a, bb);
        for (int c = 0; c < 5; ++c) o.pad[c] = 0.f;
        st[b] = o;
    }
}

__global__ __launch_bounds__(256) void augment_apply_kernel(int n, int g, float sigma, float clip, const float *__restrict__ pos,
                                                            const float *__restrict__ color, const float *__restrict__ noise,
                                                            const AugmentCloud *__restrict__ par,
                                                            const AugmentStats *__restrict__ st, const float *__restrict__ cmean,
                                                            const float *__restrict__ cstd, float *__restrict__ pos_out,
                                                            float *__restrict__ x_out, float *__restrict__ heights)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const AugmentCloud p = par[b];
    const AugmentStats s = st[b];
    const size_t o = ((size_t)b * n + i) * 3;
    float q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = __fsub_rn(__fmul_rn(pos[o + c], p.scale[c]), s.mean[c]);  // scaling, centring
    heights[(size_t)b * n + i] = pos[o + g];  // the loader's `heights`: the untransformed gravity coordinate (s3dis.py:141-142)
    q[g] = __fsub_rn(q[g], s.zmin);
#pragma unroll
    for (int r = 0; r < 3; ++r) {  // pos @ R^T, accumulated in the order of a length-3 dot product
        float v = __fmul_rn(q[0], p.rot[r * 3]);
        v = fmaf(q[1], p.rot[r * 3 + 1], v);
        v = fmaf(q[2], p.rot[r * 3 + 2], v);
        const float nz = fminf(fmaxf(__fmul_rn(noise[o + r], sigma), -clip), clip);
        pos_out[o + r] = __fadd_rn(v, nz);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float x = color[o + c];
        if (p.contrast != 0.f) {
            const float sc = __fdiv_rn(255.f, __fsub_rn(s.hi[c], s.lo[c]));
            x = __fadd_rn(__fmul_rn(__fsub_rn(1.f, p.blend), x), __fmul_rn(p.blend, __fmul_rn(__fsub_rn(x, s.lo[c]), sc)));
        }
        if (p.drop != 0.f) x = 0.f;
        if (s.cmax > 1.f) x = __fdiv_rn(x, 255.f);
        x_out[o + c] = __fdiv_rn(__fsub_rn(x, cmean[c]), cstd[c]);
    }
}

}  // namespace amc

using namespace amc;

AMC_API size_t amc3d_augment_workspace_bytes(int b) { return (size_t)(b > 0 ? b : 0) * sizeof(AugmentStats) + 64; }

// pos (b,n,3), color (b,n,3) [0..255 or 0..1], noise (b,n,3) standard normal draws, params (b,24) floats (layout above),
// color_mean / color_std (3) -> pos_out (b,n,3), x_out (b,n,3), heights (b,n).  gravity_dim in 0..2.
AMC_API int amc3d_augment_clouds(int b, int n, int gravity_dim, float jitter_sigma, float jitter_clip, const float *pos,
                                 const float *color, const float *noise, const float *params, const float *color_mean,
                                 const float *color_std, float *pos_out, float *x_out, float *heights, void *workspace,
                                 size_t workspace_bytes, void *stream_)
{
    if (b <= 0 || n <= 0) return 0;
    if (gravity_dim < 0 || gravity_dim > 2 || !pos || !color || !noise || !params || !color_mean || !color_std || !pos_out || !x_out ||
        !heights || !workspace || workspace_bytes < amc3d_augment_workspace_bytes(b))
        return bad_arg("amc3d_augment_clouds: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    AugmentStats *st = (AugmentStats *)workspace;
    hipLaunchKernelGGL(augment_stats_kernel, dim3(b), dim3(1024), 0, stream, n, gravity_dim, pos, color, (const AugmentCloud *)params, st);
    hipLaunchKernelGGL(augment_apply_kernel, dim3(div_up(n, 256), b), dim3(256), 0, stream, n, gravity_dim, jitter_sigma, jitter_clip,
                       pos, color, noise, (const AugmentCloud *)params, (const AugmentStats *)st, color_mean, color_std, pos_out,
                       x_out, heights);
    return launch_status("amc3d_augment_clouds");
}
