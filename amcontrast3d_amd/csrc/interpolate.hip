// three_nn / three_interpolate (+grad) for gfx950.
// Reference: pointnet2_batch/src/interpolate_gpu.cu:16-59, 84-104, 127-149.
#include "common.h"

namespace amc {

constexpr int NN_TILE = 512;  // known points per LDS tile
constexpr int NN_THREADS = 256;

// One thread per unknown point keeps its 3 best (strict '<' cascade, so the
// earlier index wins equal distances, exactly as the reference).  The reference
// holds the best distances in double; a double that only ever takes fp32 values
// (or the initial 1e40, which no fp32 exceeds and which casts to +inf on
// output) compares identically to fp32 with +inf as the initial value, so the
// fp64 pipe is not needed.  Known points are staged per workgroup in LDS as
// float4 and read as wave-wide broadcasts (one ds_read_b128 per candidate).
__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(int n, int m,
                                                              const float *__restrict__ unknown,
                                                              const float *__restrict__ known,
                                                              float *__restrict__ dist2, int *__restrict__ idx)
{
    __shared__ float4 tile[NN_TILE];
    const int bs = blockIdx.y;
    const int pt = blockIdx.x * NN_THREADS + threadIdx.x;
    const bool live = pt < n;
    const float *u = unknown + ((size_t)bs * n + (live ? pt : 0)) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const float *K = known + (size_t)bs * m * 3;

    const float inf = __builtin_inff();
    float best1 = inf, best2 = inf, best3 = inf;
    int besti1 = 0, besti2 = 0, besti3 = 0;

    for (int t0 = 0; t0 < m; t0 += NN_TILE) {
        const int tn = min(NN_TILE, m - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < tn; i += NN_THREADS) {
            const float *p = K + (size_t)(t0 + i) * 3;
            tile[i] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
        for (int k = 0; k < tn; ++k) {
            const float4 p = tile[k];
            const float d = dist2_ref(ux, uy, uz, p.x, p.y, p.z);
            if (d < best3) {  // rare after warm-up; the cascade below is the reference's
                const int kk = t0 + k;
                if (d < best1) {
                    best3 = best2; besti3 = besti2;
                    best2 = best1; besti2 = besti1;
                    best1 = d; besti1 = kk;
                } else if (d < best2) {
                    best3 = best2; besti3 = besti2;
                    best2 = d; besti2 = kk;
                } else {
                    best3 = d; besti3 = kk;
                }
            }
        }
    }
    if (live) {
        float *od = dist2 + ((size_t)bs * n + pt) * 3;
        int *oi = idx + ((size_t)bs * n + pt) * 3;
        od[0] = best1; od[1] = best2; od[2] = best3;
        oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
    }
}

// out[b,c,p] = w0*points[b,c,i0] + w1*points[b,c,i1] + w2*points[b,c,i2], evaluated
// left to right without contraction (interpolate_gpu.cu:103).  One thread per
// (b,p) keeps idx/weight in registers and walks the channels: coalesced stores.
// `base` (optional, (b,c,n)): out = base + interpolation -- the skip branch of a FeaturePropagation conv applied before
// the interpolation (W . [f1 ; up(f2)] = W1 . f1 + up(W2 . f2): the interpolation is linear and commutes with the 1x1 conv)
__global__ void three_interpolate_kernel(int c, int m, int n, int ch_per, const float *__restrict__ points,
                                         const int *__restrict__ idx, const float *__restrict__ weight,
                                         const float *__restrict__ base, float *__restrict__ out)
{
    const int bs = blockIdx.y;
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n) return;
    const int *ii = idx + ((size_t)bs * n + pt) * 3;
    const float *ww = weight + ((size_t)bs * n + pt) * 3;
    const int i0 = ii[0], i1 = ii[1], i2 = ii[2];
    const float w0 = ww[0], w1 = ww[1], w2 = ww[2];
    const float *src = points + (size_t)bs * c * m;
    float *dst = out + (size_t)bs * c * n + pt;
    // blockIdx.z splits the channels so that coarse levels (few points, many channels) still fill the chip
    const int ch0 = blockIdx.z * ch_per, ch1 = min(c, ch0 + ch_per);
    for (int ch = ch0; ch < ch1; ++ch) {
        const float *row = src + (size_t)ch * m;
        const float v = __fadd_rn(__fadd_rn(__fmul_rn(w0, row[i0]), __fmul_rn(w1, row[i1])), __fmul_rn(w2, row[i2]));
        dst[(size_t)ch * n] = base ? __fadd_rn(base[(size_t)bs * c * n + pt + (size_t)ch * n], v) : v;
    }
}

// grad_points[b,c,idx_j] += grad_out[b,c,p] * w_j  (interpolate_gpu.cu:146-148)
__global__ void three_interpolate_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                              const int *__restrict__ idx, const float *__restrict__ weight,
                                              float *__restrict__ grad_points)
{
    const int bs = blockIdx.y;
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n) return;
    const int *ii = idx + ((size_t)bs * n + pt) * 3;
    const float *ww = weight + ((size_t)bs * n + pt) * 3;
    const int i0 = ii[0], i1 = ii[1], i2 = ii[2];
    const float w0 = ww[0], w1 = ww[1], w2 = ww[2];
    const float *src = grad_out + (size_t)bs * c * n + pt;
    float *dst = grad_points + (size_t)bs * c * m;
    for (int ch = 0; ch < c; ++ch) {
        const float g = src[(size_t)ch * n];
        float *row = dst + (size_t)ch * m;
        atomicAdd(row + i0, __fmul_rn(g, w0));
        atomicAdd(row + i1, __fmul_rn(g, w1));
        atomicAdd(row + i2, __fmul_rn(g, w2));
    }
}

}  // namespace amc

using namespace amc;

namespace amc {
bool grid_search_pays(int b, int n, int m);
size_t grid_search_workspace_bytes(int b, int n, int m);
int three_nn_grid(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx, void *workspace,
                  hipStream_t stream);
}

AMC_API int amc3d_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                           int *idx, void *workspace, size_t workspace_bytes, void *stream)
{
    if (b <= 0 || n <= 0) return 0;
    if (m < 0 || !unknown || !known || !dist2 || !idx) return bad_arg("amc3d_three_nn: bad argument");
    if (workspace && m >= 3 && grid_search_pays(b, m, n) && workspace_bytes >= grid_search_workspace_bytes(b, m, n))
        return three_nn_grid(b, n, m, unknown, known, dist2, idx, workspace, (hipStream_t)stream);
    hipLaunchKernelGGL(three_nn_kernel, dim3(div_up(n, NN_THREADS), b), dim3(NN_THREADS), 0, (hipStream_t)stream, n,
                       m, unknown, known, dist2, idx);
    return launch_status("amc3d_three_nn");
}

static int three_interpolate_launch(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                                    const float *base, float *out, void *stream);

AMC_API int amc3d_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                    const float *weight, float *out, void *stream)
{
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (!points || !idx || !weight || !out) return bad_arg("amc3d_three_interpolate: null pointer");
    return three_interpolate_launch(b, c, m, n, points, idx, weight, nullptr, out, stream);
}

// out (b,c,n) = base (b,c,n) + three_interpolate(points (b,c,m)); out may alias base
AMC_API int amc3d_three_interpolate_add(int b, int c, int m, int n, const float *points, const int *idx,
                                        const float *weight, const float *base, float *out, void *stream)
{
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (!points || !idx || !weight || !base || !out) return bad_arg("amc3d_three_interpolate_add: null pointer");
    return three_interpolate_launch(b, c, m, n, points, idx, weight, base, out, stream);
}

static int three_interpolate_launch(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                                    const float *base, float *out, void *stream)
{
    long chunks = 262144 / ((long)b * n);
    if (chunks < 1) chunks = 1;
    if (chunks > c) chunks = c;
    const int ch_per = div_up(c, chunks);
    hipLaunchKernelGGL(three_interpolate_kernel, dim3(div_up(n, 256), b, div_up(c, ch_per)), dim3(256), 0,
                       (hipStream_t)stream, c, m, n, ch_per, points, idx, weight, base, out);
    return launch_status("amc3d_three_interpolate");
}

namespace amc {
int scatter_add_pm(int fan, int b, int c, int n, long entries, const float *grad_out, const int *idx,
                   const float *weight, float *grad_points, float *scratch, hipStream_t stream, const char *what);
}

AMC_API int amc3d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                         const float *weight, float *grad_points, void *workspace,
                                         size_t workspace_bytes, void *stream)
{
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (!grad_out || !idx || !weight || !grad_points) return bad_arg("amc3d_three_interpolate_grad: null pointer");
    if (workspace && workspace_bytes >= (size_t)b * c * m * sizeof(float) && c >= 8)
        return scatter_add_pm(3, b, c, m, n, grad_out, idx, weight, grad_points, (float *)workspace,
                              (hipStream_t)stream, "amc3d_three_interpolate_grad");
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(div_up(n, 256), b), dim3(256), 0, (hipStream_t)stream, c, n,
                       m, grad_out, idx, weight, grad_points);
    return launch_status("amc3d_three_interpolate_grad");
}
