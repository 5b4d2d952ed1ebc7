// The residual branch of a strided SetAbstraction block as two kernel families instead of ten library launches.
//
// Reference: openpoints/models/backbone/pointnext_AA.py:157-168 (SetAbstraction.forward with use_res, PointNeXt-S):
//     fi       = torch.gather(f, -1, idx.unsqueeze(1).expand(-1, f.shape[1], -1))      # features at the FPS picks
//     identity = self.skipconv(fi)                                                     # Conv1d k=1 WITH bias, no norm/act
//     ...
//     f        = self.act(f + identity)                                                # ReLU
// As torch operators that is gather -> convolution library (+ layout transposes around its weight gradient) -> add -> ReLU
// forward and threshold -> bias sum -> three conv-backward launches -> zero-fill -> scatter-add backward, 0.2 GFLOP per
// stage and ~45 launches per step for the four stages.  Here:
//
//  * sa_res_fwd_kernel       out = relu(y + W . f[:, idx] + bias): the columns of f are gathered while the tile is staged
//                            (and written once as fi for the weight gradient), the residual, bias and ReLU sit in the epilogue.
//  * sa_res_mask_kernel      g = dout * (out > 0) (the gradient of BOTH branches), per-cloud bias-gradient sums, and the
//                            zero-fill of df (only the sampled columns receive a gradient).
//  * sa_res_bwd_data_kernel  df[:, idx] += W^T . g: the scatter is the epilogue's add into the zero-filled df (repeated picks
//                            sum, as torch.gather's backward does; one add per address -- deterministic -- where the picks
//                            are distinct); its first workgroup finishes the bias gradient in a fixed order.
//  * the weight gradient dW = sum g fi^T is amc3d_pointwise_conv_backward on (g, fi) (deterministic partial sums).
//
// The products are tiny (K = 32..256, 744..48000 positions): a 64 x 64 register-blocked fp32 tile per workgroup keeps
// every launch at a few hundred workgroups and a few microseconds; the MFMA would not be visible next to the launch itself.
#include "common.h"

namespace amc {

constexpr int SR_T = 64;        // tile edge: 64 rows (channels) x 64 columns (positions)
constexpr int SR_KC = 32;       // reduction chunk staged in LDS
constexpr int SR_LDA = SR_T + 4;  // padded row of the A image when it is stored transposed (4-way instead of 32-way conflicts)

// acc[i][j] += sum_kk as[kk][ty*4 + i] * bs[kk][tx*4 + j]
__device__ __forceinline__ int div_up_dev(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ void sr_multiply(const float *as, int lda, const float *bs, int tx, int ty, float (&acc)[4][4])
{
#pragma unroll 8
    for (int kk = 0; kk < SR_KC; ++kk) {
        const float4 a = *(const float4 *)(as + kk * lda + ty * 4);
        const float4 v = *(const float4 *)(bs + kk * SR_T + tx * 4);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(av[i], bv[j], acc[i][j]);
    }
}

// sum the KG k-groups' accumulators through LDS in group order (group 0 ends up with the total); all threads call it
template <int KG>
__device__ __forceinline__ void sr_reduce_groups(float (*smem)[SR_KC * SR_LDA + SR_KC * SR_T], int q, int tx, int ty, float (&acc)[4][4])
{
    if (KG == 1) return;
    __syncthreads();  // staging images consumed
    if (q > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(float4 *)(smem[q] + (ty * 4 + i) * SR_T + tx * 4) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int g = 1; g < KG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 v = *(const float4 *)(smem[g] + (ty * 4 + i) * SR_T + tx * 4);
                acc[i][0] += v.x; acc[i][1] += v.y; acc[i][2] += v.z; acc[i][3] += v.w;
            }
    }
}

// out[b][co][p] = relu(y[b][co][p] + (bias[co] + sum_ci w[co][ci] f[b][ci][idx[b][p]]));  fi[b][ci][p] = f[b][ci][idx[b][p]]
// KG groups of 256 threads share a tile and split its K chunks (the coarse stages have 64-128 tiles with K = 128-512: one
// group per CU walks the K axis at one wave per SIMD, 4-5 us per chunk, measured)
template <int KG>
__global__ __launch_bounds__(256 * KG) void sa_res_fwd_kernel(int nb, int cin, int cout, int n, int m, const float *__restrict__ f,
                                                              const int *__restrict__ idx, const float *__restrict__ w,
                                                              const float *__restrict__ bias, const float *__restrict__ y,
                                                              float *__restrict__ out, float *__restrict__ fi)
{
    __shared__ __attribute__((aligned(16))) float smem[KG][SR_KC * SR_LDA + SR_KC * SR_T];
    __shared__ int cols[SR_T];
    const int t = threadIdx.x & 255, q = threadIdx.x >> 8;
    float *as = smem[q], *bs = smem[q] + SR_KC * SR_LDA;  // [k][co], [k][p]
    const int tx = t & 15, ty = t >> 4;
    // blockIdx.x = tile * nb + cloud: workgroups are dealt to the 8 XCDs round-robin, so with 8 clouds (or 2, 4) every cloud's
    // tiles meet in ONE XCD, whose L2 then holds that cloud's features (3 MB at the first stage) for the column gathers
    const int b = blockIdx.x % nb, p0 = (blockIdx.x / nb) * SR_T, c0 = blockIdx.y * SR_T;
    if (threadIdx.x < SR_T) {
        const int p = p0 + threadIdx.x;
        int col = p < m ? idx[(size_t)b * m + p] : -1;
        if (col >= n) col = -1;  // never read outside the cloud
        cols[threadIdx.x] = col;
    }
    const float *F = f + (size_t)b * cin * n;
    float acc[4][4] = {};
    // a chunk = 8 + 8 values per thread, loaded into registers as one batch (16 loads in flight) one chunk ahead of the
    // multiply
    constexpr int LPT = SR_T * SR_KC / 256;
    float ra[LPT], rb[LPT];
    __syncthreads();  // cols visible
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            const int i = t + u * 256;
            const int r = i / SR_KC, kk = i - r * SR_KC;
            const int co = c0 + r, k = k0 + kk;
            ra[u] = (co < cout && k < cin) ? w[(size_t)co * cin + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            const int i = t + u * 256;
            const int kk = i / SR_T, c = i - kk * SR_T;
            const int k = k0 + kk, col = cols[c];
            rb[u] = (k < cin && col >= 0) ? F[(size_t)k * n + col] : 0.f;
        }
    };
    const int iters = div_up_dev(div_up_dev(cin, SR_KC), KG);
    if (q * SR_KC < cin) fetch(q * SR_KC);
    for (int it = 0; it < iters; ++it) {
        const int k0 = (it * KG + q) * SR_KC;  // wave-uniform
        __syncthreads();  // previous chunk consumed
        if (k0 < cin) {
#pragma unroll
            for (int u = 0; u < LPT; ++u) {
                const int i = t + u * 256;
                const int r = i / SR_KC, kk = i - r * SR_KC;
                as[kk * SR_LDA + r] = ra[u];
                const int kb = i / SR_T, c = i - kb * SR_T;
                bs[i] = rb[u];
                if (fi && blockIdx.y == 0 && k0 + kb < cin && p0 + c < m) fi[((size_t)b * cin + k0 + kb) * m + p0 + c] = rb[u];
            }
        }
        __syncthreads();
        if (k0 + KG * SR_KC < cin) fetch(k0 + KG * SR_KC);
        if (k0 < cin) sr_multiply(as, SR_LDA, bs, tx, ty, acc);
    }
    sr_reduce_groups<KG>(smem, q, tx, ty, acc);
    if (q != 0) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = c0 + ty * 4 + i;
        if (co >= cout) continue;
        const float bv = bias ? bias[co] : 0.f;
        const size_t row = ((size_t)b * cout + co) * m;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = p0 + tx * 4 + j;
            if (p < m) out[row + p] = fmaxf(y[row + p] + (acc[i][j] + bv), 0.f);
        }
    }
}

// g = dout * (out > 0); dbp[b][co] = sum_p g[b][co][p]; df (total_df floats) zero-filled, a slice per workgroup
__global__ __launch_bounds__(256) void sa_res_mask_kernel(int cout, int m, const float *__restrict__ dout,
                                                          const float *__restrict__ out, float *__restrict__ g,
                                                          float *__restrict__ dbp, float *__restrict__ df, size_t total_df)
{
    __shared__ float red[4];
    const int co = blockIdx.x, b = blockIdx.y;
    const size_t row = ((size_t)b * cout + co) * m;
    float s = 0.f;
    for (int p = threadIdx.x; p < m; p += 256) {
        const float v = out[row + p] > 0.f ? dout[row + p] : 0.f;
        g[row + p] = v;
        s += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dbp[(size_t)b * cout + co] = (red[0] + red[1]) + (red[2] + red[3]);
    if (df) {
        const size_t nwg = (size_t)gridDim.x * gridDim.y, wg = (size_t)b * gridDim.x + co;
        const size_t per = (total_df + nwg - 1) / nwg;
        const size_t lo = wg * per, hi = lo + per < total_df ? lo + per : total_df;
        for (size_t i = lo + threadIdx.x; i < hi; i += 256) df[i] = 0.f;
    }
}

// df[b][ci][idx[b][p]] += sum_co w[co][ci] g[b][co][p] (df zero-filled by sa_res_mask_kernel);  workgroup (0,0,0) also sums the bias gradient over the clouds
template <int KG>
__global__ __launch_bounds__(256 * KG) void sa_res_bwd_data_kernel(int nb, int cin, int cout, int n, int m,
                                                                   const float *__restrict__ g, const int *__restrict__ idx,
                                                                   const float *__restrict__ w, float *__restrict__ df,
                                                                   const float *__restrict__ dbp, float *__restrict__ db,
                                                                   const int *__restrict__ dup_flag)
{
    __shared__ __attribute__((aligned(16))) float smem[KG][SR_KC * SR_LDA + SR_KC * SR_T];
    __shared__ int cols[SR_T];
    const bool dup = !dup_flag || *dup_flag != 0;  // repeated picks (or unknown): adds
    const int t = threadIdx.x & 255, q = threadIdx.x >> 8;
    float *as = smem[q], *bs = smem[q] + SR_KC * SR_LDA;  // [k = co][ci], [k = co][p]
    const int tx = t & 15, ty = t >> 4;
    // blockIdx.x = tile * nb + cloud: workgroups are dealt to the 8 XCDs round-robin, so with 8 clouds (or 2, 4) every cloud's
    // tiles meet in ONE XCD, whose L2 then holds that cloud's features (3 MB at the first stage) for the column gathers
    const int b = blockIdx.x % nb, p0 = (blockIdx.x / nb) * SR_T, c0 = blockIdx.y * SR_T;
    if (db && blockIdx.x == 0 && blockIdx.y == 0) {
        for (int co = threadIdx.x; co < cout; co += 256 * KG) {
            float s = 0.f;
            for (int bb = 0; bb < nb; ++bb) s += dbp[(size_t)bb * cout + co];
            db[co] = s;
        }
    }
    if (!df) return;
    if (threadIdx.x < SR_T) {
        const int p = p0 + threadIdx.x;
        int col = p < m ? idx[(size_t)b * m + p] : -1;
        if (col >= n) col = -1;
        cols[threadIdx.x] = col;
    }
    const float *G = g + (size_t)b * cout * m;
    float acc[4][4] = {};
    constexpr int LPT = SR_T * SR_KC / 256;
    float ra[LPT], rb[LPT];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            const int i = t + u * 256;
            const int kk = i / SR_T, r = i - kk * SR_T;
            const int k = k0 + kk, ci = c0 + r, p = p0 + r;
            ra[u] = (k < cout && ci < cin) ? w[(size_t)k * cin + ci] : 0.f;
            rb[u] = (k < cout && p < m) ? G[(size_t)k * m + p] : 0.f;
        }
    };
    const int iters = div_up_dev(div_up_dev(cout, SR_KC), KG);
    if (q * SR_KC < cout) fetch(q * SR_KC);
    for (int it = 0; it < iters; ++it) {
        const int k0 = (it * KG + q) * SR_KC;  // wave-uniform
        __syncthreads();  // previous chunk consumed (first pass: cols visible)
        if (k0 < cout) {
#pragma unroll
            for (int u = 0; u < LPT; ++u) {
                const int i = t + u * 256;
                as[i] = ra[u];
                bs[i] = rb[u];
            }
        }
        __syncthreads();
        if (k0 + KG * SR_KC < cout) fetch(k0 + KG * SR_KC);
        if (k0 < cout) sr_multiply(as, SR_T, bs, tx, ty, acc);
    }
    sr_reduce_groups<KG>(smem, q, tx, ty, acc);
    if (q != 0) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = c0 + ty * 4 + i;
        if (ci >= cin) continue;
        float *row = df + ((size_t)b * cin + ci) * n;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = cols[tx * 4 + j];
            // FPS re-picks a point when a cloud holds fewer distinct points than picks (crop_pc pads small rooms by
            // repetition, data_util.py:161-167): torch.gather's backward sums over repeated indices, so does this -- an add
            // into the zero-filled df (one add per address, hence still deterministic, wherever the picks are distinct)
            // (a plain store where the plan found the picks of this batch distinct -- the usual case: scattered 4-byte float
            //  atomics are the slowest access there is, 83 us for the 1.5 M adds of stage 1 against 25 us of stores)
            if (col >= 0) {
                if (dup) atomicAdd(row + col, acc[i][j]);
                else row[col] = acc[i][j];
            }
        }
    }
}

// k-groups per tile: none where the K axis is one or two chunks or the tiles already fill the chip
static int sr_kgroups(int k, long tiles) { return (k <= 2 * SR_KC || tiles >= 1024) ? 1 : (k <= 4 * SR_KC || tiles >= 512) ? 2 : 4; }

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace amc

using namespace amc;

AMC_API int amc3d_sa_residual_forward(int b, int cin, int cout, int n, int m, const float *f, const int *fps_idx,
                                      const float *weight, const float *bias, const float *y, float *out, float *fi,
                                      void *stream)
{
    if (b <= 0 || m <= 0 || cout <= 0) return 0;
    if (cin <= 0 || n <= 0 || !f || !fps_idx || !weight || !y || !out) return bad_arg("amc3d_sa_residual_forward: bad argument");
    const dim3 grid(div_up(m, SR_T) * b, div_up(cout, SR_T), 1);
    const int kg = sr_kgroups(cin, (long)grid.x * grid.y * grid.z);
#define AMC_SRF(KG) hipLaunchKernelGGL((sa_res_fwd_kernel<KG>), grid, dim3(256 * KG), 0, (hipStream_t)stream, b, cin, cout, n, m, f, \
                                       fps_idx, weight, bias, y, out, fi)
    if (kg == 1) AMC_SRF(1); else if (kg == 2) AMC_SRF(2); else AMC_SRF(4);
#undef AMC_SRF
    return launch_status("amc3d_sa_residual_forward");
}

AMC_API size_t amc3d_sa_residual_workspace_bytes(int b, int cin, int cout, int m)
{
    if (b <= 0 || cin <= 0 || cout <= 0 || m <= 0) return 0;
    return align256((size_t)b * cout * sizeof(float)) + amc3d_pointwise_conv_workspace_bytes(b, cin, cout, m);
}

namespace amc {
__global__ void index_mark_kernel(int n, int m, const int *__restrict__ idx, int *__restrict__ mark, int *__restrict__ flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= m) return;
    const int v = idx[(size_t)b * m + i];
    if (v < 0 || v >= n || atomicExch(mark + (size_t)b * n + v, 1) != 0) *flag = 1;  // (out of range counts as "not distinct")
}
}  // namespace amc

// flag[0] = 1 if some cloud's picks idx (b, m) into n points repeat an index (or leave [0, n)), else 0.  workspace: b * n ints.
// Coordinates only: part of the sampling plan; amc3d_sa_residual_backward scatters with plain stores where the flag is 0.
AMC_API size_t amc3d_index_duplicates_workspace_bytes(int b, int n) { return (size_t)(b > 0 ? b : 0) * (size_t)(n > 0 ? n : 0) * sizeof(int) + 16; }
AMC_API int amc3d_index_duplicates(int b, int n, int m, const int *idx, int *flag, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (!flag) return bad_arg("amc3d_index_duplicates: null flag");
    hipStream_t stream = (hipStream_t)stream_;
    if (int st = fill_i32(flag, 0, 1, stream)) return st;
    if (b <= 0 || m <= 0) return 0;
    if (n <= 0 || !idx || !workspace || workspace_bytes < amc3d_index_duplicates_workspace_bytes(b, n))
        return bad_arg("amc3d_index_duplicates: bad argument or workspace too small");
    if (int st = fill_i32((int *)workspace, 0, (size_t)b * n, stream)) return st;
    hipLaunchKernelGGL(index_mark_kernel, dim3(div_up(m, 256), b), dim3(256), 0, stream, n, m, idx, (int *)workspace, flag);
    return launch_status("amc3d_index_duplicates");
}

AMC_API int amc3d_sa_residual_backward(int b, int cin, int cout, int n, int m, const float *dout, const float *out,
                                       const float *fi, const int *fps_idx, const int *dup_flag, const float *weight, float *g,
                                       float *df, float *dweight, float *dbias, void *workspace, size_t workspace_bytes,
                                       void *stream_)
{
    if (b <= 0 || m <= 0 || cout <= 0) return 0;
    if (cin <= 0 || n <= 0 || !dout || !out || !fps_idx || !weight || !g || !workspace ||
        workspace_bytes < amc3d_sa_residual_workspace_bytes(b, cin, cout, m) || (dweight && !fi))
        return bad_arg("amc3d_sa_residual_backward: bad argument or workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    float *dbp = (float *)workspace;
    const size_t head = align256((size_t)b * cout * sizeof(float));
    hipLaunchKernelGGL(sa_res_mask_kernel, dim3(cout, b), dim3(256), 0, stream, cout, m, dout, out, g, dbp, df,
                       (size_t)b * cin * n);
    if (df || dbias) {
        const dim3 grid = df ? dim3(div_up(m, SR_T) * b, div_up(cin, SR_T), 1) : dim3(1, 1, 1);
        const int kg = df ? sr_kgroups(cout, (long)grid.x * grid.y * grid.z) : 1;
#define AMC_SRB(KG) hipLaunchKernelGGL((sa_res_bwd_data_kernel<KG>), grid, dim3(256 * KG), 0, stream, b, cin, cout, n, m, \
                                       (const float *)g, fps_idx, weight, df, (const float *)dbp, dbias, dup_flag)
        if (kg == 1) AMC_SRB(1); else if (kg == 2) AMC_SRB(2); else AMC_SRB(4);
#undef AMC_SRB
    }
    if (int st = launch_status("amc3d_sa_residual_backward")) return st;
    if (dweight)
        return amc3d_pointwise_conv_backward(b, cin, cout, m, fi, weight, g, nullptr, dweight, (char *)workspace + head,
                                             workspace_bytes - head, stream_);
    return 0;
}
