// Neighbourhood aggregation with ONE grouped convolution -- every LocalAggregation of the InvResMLP blocks and every
// single-layer SetAbstraction of PointNeXt-B/L/XL -- as "convolve first, gather after" (gfx950).
//
// Reference (pointnext_AA.py:57-63, 139-170; group.py:244-255, 323-325): ball query -> grouping_operation gathers the
// neighbours' features into (B,C,M,32) -> cat with the relative positions dp (B,3,M,32) -> Conv2d 1x1 (C+3 -> C') ->
// BatchNorm2d (batch statistics) -> ReLU -> max over the 32 neighbours.  Every one of those tensors is 32x the size
// of the layer's input and output; the conv runs on M*32 positions.
//
// The conv is linear and the gather only selects columns, so  W . [dp ; f[idx]] = (W_f . f)[idx] + W_dp . dp :
//   G (B,C',N) = W_f . f                       one pointwise conv on the N source points (32x fewer positions)
//   y[b,c,m,k] = G[b,c,idx[b,m,k]] + W_dp[c] . dp[b,:,m,k]
// and the BatchNorm statistics of y over all B*M*32 positions follow from N-sized sums and three geometry moments
//   cnt[n] = #{(m,k): idx[m,k] = n},   D[n] = sum of dp over those positions,   S1 = sum_p dp_p,  S2 = sum_p dp_p dp_p^T :
//   sum_p y_p   = sum_n cnt[n] G[n] + W_dp . S1
//   sum_p y_p^2 = sum_n cnt[n] G[n]^2 + 2 sum_n G[n] (W_dp . D[n]) + W_dp^T S2 W_dp
// (accumulated in fp64).  The only pass over M*32 positions is the max-pool itself: gather 32 rows of G per centroid,
// add the 3-term dp product, normalise, keep the first maximum (torch.max's rule).  Nothing of size (B,C,M,32) exists.
//
// Backward.  The gradient of the pooled output reaches ONE neighbour per (b,c,m); BatchNorm's backward then makes
// it dense again, dz_p = g*is*(dq_p - mean(dq) - xhat_p*mean(dq*xhat)), but what the layer needs is only
//   dG[n] = sum_{p -> n} dz_p = g*is*( Q[n] - cnt[n]*ma - mb*is*( cnt[n]*(G[n] - mu) + W_dp . D[n] ) ),
// with Q[n] the scattered sparse gradients (M*C' atomics instead of M*32*C'), and dW_dp, dgamma, dbeta from the same
// M-sized and N-sized sums.  df = W_f^T dG and dW_f = dG f^T are the pointwise conv's backward on N points.
// The geometry moments depend on coordinates only and are part of the geometry plan (computed ahead, shared by all
// blocks of a stage).
#include <stdlib.h>

#include "common.h"

namespace amc {

constexpr double LAGG_FX_D = 68719476736.0;   // 2^36: fixed point of the per-point dp sums (|dp| <= radius or 1)
constexpr double LAGG_FX_M = 1073741824.0;    // 2^30: fixed point of the global moments (block partials in double)
constexpr int LAGG_MT = 32;                   // centroids per workgroup tile (8 per wave)
constexpr int LAGG_CT = 128;                  // channels per workgroup (pool / scatter kernels)
constexpr int LAGG_NT = 64;                   // points per tile (statistics / apply kernels)
constexpr int LAGG_TILES = 4;                 // tiles per workgroup in the partial-sum kernels

struct LaggMoments {  // views into the opaque moments buffer of amc3d_group_moments
    const long long *mom;   // [9 (+7 pad)] S1[3], S2[(0,0),(0,1),(0,2),(1,1),(1,2),(2,2)]
    const int *cnt;         // (b, n)
    const long long *dfx;   // (b, n, 3)
};

static size_t lagg_moments_bytes(int b, int n) { return 16 * 8 + (size_t)b * n * 4 + (((size_t)b * n) & 1) * 4 + (size_t)b * n * 24; }

static LaggMoments lagg_views(const void *buf, int b, int n)
{
    LaggMoments v;
    const char *p = (const char *)buf;
    v.mom = (const long long *)p;
    v.cnt = (const int *)(p + 16 * 8);
    v.dfx = (const long long *)(p + 16 * 8 + (size_t)b * n * 4 + (((size_t)b * n) & 1) * 4);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// geometry moments: integer / fixed-point atomics, so the result does not depend on the order of arrival
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lagg_geom_kernel(int n, long P, const int *__restrict__ idx, const float *__restrict__ dp,
                                                        int *__restrict__ cnt, unsigned long long *__restrict__ dfx,
                                                        unsigned long long *__restrict__ mom)
{
    __shared__ double red[4][9];
    const int b = blockIdx.y;
    const int *ib = idx + (size_t)b * P;
    const float *d0 = dp + (size_t)b * 3 * P, *d1 = d0 + P, *d2 = d1 + P;
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        const int id = ib[p];
        const float x = d0[p], y = d1[p], z = d2[p];
        if (id >= 0 && id < n) {
            atomicAdd(cnt + (size_t)b * n + id, 1);
            unsigned long long *t = dfx + ((size_t)b * n + id) * 3;
            atomicAdd(t + 0, (unsigned long long)__double2ll_rn((double)x * LAGG_FX_D));
            atomicAdd(t + 1, (unsigned long long)__double2ll_rn((double)y * LAGG_FX_D));
            atomicAdd(t + 2, (unsigned long long)__double2ll_rn((double)z * LAGG_FX_D));
        }
        const double dx = x, dy = y, dz = z;
        s[0] += dx; s[1] += dy; s[2] += dz;
        s[3] += dx * dx; s[4] += dx * dy; s[5] += dx * dz; s[6] += dy * dy; s[7] += dy * dz; s[8] += dz * dz;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double v = s[j];
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const double v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        atomicAdd(mom + threadIdx.x, (unsigned long long)__double2ll_rn(v * LAGG_FX_M));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward statistics: per channel  sum_n cnt G,  sum_n cnt G^2,  sum_n G D_j  (j = 0..2)  + the point-major copy of G
// grid (n-tile groups, channel chunks of 64, b)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lagg_stats_kernel(int C, int n, const float *__restrict__ g_cm, float *__restrict__ g_pm,
                                                         LaggMoments gm, double *__restrict__ partial, int nparts_per_b,
                                                         int tiles_per_wg)
{
    __shared__ float tile[64][LAGG_NT + 1];
    __shared__ double red[4][64][5];
    const int b = blockIdx.z, c0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = c0 + lane;
    double a[5] = {0, 0, 0, 0, 0};
    for (int tt = 0; tt < tiles_per_wg; ++tt) {
        const int n0 = (blockIdx.x * tiles_per_wg + tt) * LAGG_NT;
        if (n0 >= n) break;
        __syncthreads();
        for (int r = wave; r < 64; r += 4) {  // r: channel of the chunk, lane: point
            const int ch = c0 + r, nn = n0 + lane;
            tile[r][lane] = (ch < C && nn < n) ? g_cm[((size_t)b * C + ch) * n + nn] : 0.f;
        }
        __syncthreads();
        for (int i = 0; i < 16; ++i) {
            const int nl = wave * 16 + i, nn = n0 + nl;
            if (nn >= n) break;  // wave-uniform
            const float g = tile[lane][nl];
            if (c < C) g_pm[((size_t)b * n + nn) * C + c] = g;
            const double cn = (double)gm.cnt[(size_t)b * n + nn];
            const long long *df = gm.dfx + ((size_t)b * n + nn) * 3;
            const double dx = (double)df[0] * (1.0 / LAGG_FX_D), dy = (double)df[1] * (1.0 / LAGG_FX_D),
                         dz = (double)df[2] * (1.0 / LAGG_FX_D);
            const double gd = g;
            a[0] += cn * gd; a[1] += cn * gd * gd; a[2] += gd * dx; a[3] += gd * dy; a[4] += gd * dz;
        }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) red[wave][lane][j] = a[j];
    __syncthreads();
    if (wave == 0 && c < C) {
        double *out = partial + (((size_t)b * nparts_per_b + blockIdx.x) * C + c) * 5;
#pragma unroll
        for (int j = 0; j < 5; ++j) out[j] = (red[0][lane][j] + red[1][lane][j]) + (red[2][lane][j] + red[3][lane][j]);
    }
}

// mean / invstd / unbiased variance of y from the partials and the geometry moments; gd[c][3] = sum_n G D (kept for backward)
// mode 0: everything from this rank's sums.  Statistics over several ranks (torch.nn.SyncBatchNorm semantics,
// main_AA.py:146-148): mode 1 writes this rank's {sum y, sum y^2} per channel and its position count to sums (2C + 1
// doubles) for the caller to all-reduce, mode 2 derives mean / invstd from the reduced sums.
__global__ __launch_bounds__(256) void lagg_stats_finalize_kernel(int C, int nparts, double count, float eps, float momentum,
                                                                  const double *__restrict__ partial, const long long *__restrict__ mom,
                                                                  const float *__restrict__ w_dp, float *__restrict__ mean,
                                                                  float *__restrict__ invstd, float *__restrict__ var_unbiased,
                                                                  double *__restrict__ gd, float *__restrict__ running_mean,
                                                                  float *__restrict__ running_var, long long *__restrict__ tracked,
                                                                  int mode, double *__restrict__ sums)
{
    __shared__ double red[4][5];
    const int c = blockIdx.x;
    double a[5] = {0, 0, 0, 0, 0};
    if (mode == 2) nparts = 0;
    for (int k = threadIdx.x; k < nparts; k += 256) {
        const double *p = partial + ((size_t)k * C + c) * 5;
#pragma unroll
        for (int j = 0; j < 5; ++j) a[j] += p[j];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        double v = a[j];
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
#pragma unroll
    for (int j = 0; j < 5; ++j) a[j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    double m[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) m[j] = (double)mom[j] * (1.0 / LAGG_FX_M);
    const double w0 = w_dp[c * 3 + 0], w1 = w_dp[c * 3 + 1], w2 = w_dp[c * 3 + 2];
    const double sy = a[0] + w0 * m[0] + w1 * m[1] + w2 * m[2];
    const double quad = w0 * w0 * m[3] + w1 * w1 * m[6] + w2 * w2 * m[8] + 2.0 * (w0 * w1 * m[4] + w0 * w2 * m[5] + w1 * w2 * m[7]);
    double sy2 = a[1] + 2.0 * (w0 * a[2] + w1 * a[3] + w2 * a[4]) + quad, sy1 = sy;
    if (mode != 2) { gd[c * 3 + 0] = a[2]; gd[c * 3 + 1] = a[3]; gd[c * 3 + 2] = a[4]; }
    if (mode == 1) {
        sums[2 * c] = sy1; sums[2 * c + 1] = sy2;
        if (c == 0) sums[2 * C] = count;
        return;
    }
    if (mode == 2) { sy1 = sums[2 * c]; sy2 = sums[2 * c + 1]; count = sums[2 * C]; }
    const double mu = sy1 / count;
    double var = sy2 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float mf = (float)mu, vu = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
    mean[c] = mf;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    var_unbiased[c] = vu;
    if (running_mean && momentum >= 0.f) {
        running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * mf;
        running_var[c] = running_var[c] * (1.f - momentum) + momentum * vu;
        if (c == 0 && tracked) *tracked += 1;
    }
}

__device__ __forceinline__ float lagg_bn(float x, float mean, float invstd, float gamma, float beta)
{
    return __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(x, mean), invstd), gamma), beta);
}

// ---------------------------------------------------------------------------------------------------------------
// forward max-pool: pooled[b,c,m] = max_k [relu](bn(G[b, idx[b,m,k], c] + W_dp[c] . dp[b,:,m,k])), arg = first k, ystar = raw y there
// grid (m tiles of 32, channel chunks of <= 128, b); a wave owns 8 centroids; LPR lanes share one gathered row
// (16 bytes each), 64/LPR rows per load instruction
// ---------------------------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void lagg_pool_kernel(int C, int n, int M, int K, int cpw, int relu,
                                                        const float *__restrict__ g_pm, const int *__restrict__ idx,
                                                        const float *__restrict__ dp, const float *__restrict__ w_dp,
                                                        const float *__restrict__ mean, const float *__restrict__ invstd,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        float *__restrict__ pooled, unsigned char *__restrict__ arg,
                                                        float *__restrict__ ystar)
{
    extern __shared__ float lagg_smem[];
    constexpr int ct = 4 * LPR, rpi = 64 / LPR;   // channels of this chunk; gathered rows per load instruction
    const int MT = 4 * cpw;                        // centroids per workgroup (cpw per wave)
    float *sp = lagg_smem;                         // [ct][MT + 1] pooled
    float *sy = sp + ct * (MT + 1);                // ystar
    float *sa = sy + ct * (MT + 1);                // arg (as float bits of an int)
    const int b = blockIdx.z, c0 = blockIdx.y * ct, m0 = blockIdx.x * MT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % LPR, r = lane / LPR;
    const int cq = c0 + 4 * q;
    float w[4][3], mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = cq + j;
        w[j][0] = w_dp[c * 3 + 0]; w[j][1] = w_dp[c * 3 + 1]; w[j][2] = w_dp[c * 3 + 2];
        mu[j] = mean[c]; is[j] = invstd[c]; ga[j] = gamma[c]; be[j] = beta[c];
    }
    const long P = (long)M * K;
    // a centroid's indices and relative positions (lane = neighbour) are fetched while the centroid before it is pooled: the
    // gathers of a centroid wait for its indices, and a wave walks its centroids one after the other
    int id_n = 0;
    float n0 = 0.f, n1 = 0.f, n2 = 0.f;
    auto fetch = [&](int m_) {
        id_n = 0; n0 = n1 = n2 = 0.f;
        if (lane < K && m_ < M) {
            const size_t p = (size_t)m_ * K + lane;
            id_n = idx[(size_t)b * P + p];
            n0 = dp[((size_t)b * 3 + 0) * P + p]; n1 = dp[((size_t)b * 3 + 1) * P + p]; n2 = dp[((size_t)b * 3 + 2) * P + p];
        }
    };
    fetch(m0 + wave * cpw);
    for (int i = 0; i < cpw; ++i) {
        const int ml = wave * cpw + i, m = m0 + ml;
        if (m >= M) break;  // wave-uniform
        const int id_l = id_n;
        const float d0 = n0, d1 = n1, d2 = n2;
        if (i + 1 < cpw) fetch(m + 1);
        float best[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        float by[4] = {0.f, 0.f, 0.f, 0.f};
        int bk[4] = {0, 0, 0, 0};
        auto visit = [&](int k0) {
            const int k = k0 + r;
            const int ks = k < K ? k : K - 1;
            const int id = __shfl(id_l, ks, 64);
            const float e0 = __shfl(d0, ks, 64), e1 = __shfl(d1, ks, 64), e2 = __shfl(d2, ks, 64);
            const float4 g = *reinterpret_cast<const float4 *>(g_pm + ((size_t)b * n + id) * C + cq);
            const float gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = __fmaf_rn(w[j][0], e0, __fmaf_rn(w[j][1], e1, __fmaf_rn(w[j][2], e2, gs[j])));
                float v = lagg_bn(y, mu[j], is[j], ga[j], be[j]);
                if (relu) v = fmaxf(v, 0.f);
                if (k < K && v > best[j]) { best[j] = v; bk[j] = k; by[j] = y; }
            }
        };
        if (K == 32) {  // the configured neighbourhood size: fully unrolled, all gathers of a centroid in flight together
#pragma unroll
            for (int k0 = 0; k0 < 32; k0 += rpi) visit(k0);
        } else {
            for (int k0 = 0; k0 < K; k0 += rpi) visit(k0);
        }
#pragma unroll
        for (int s = LPR; s < 64; s <<= 1) {  // combine the row slots, first index wins among equal values
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ov = __shfl_xor(best[j], s, 64), oy = __shfl_xor(by[j], s, 64);
                const int ok = __shfl_xor(bk[j], s, 64);
                if (ov > best[j] || (ov == best[j] && ok < bk[j])) { best[j] = ov; bk[j] = ok; by[j] = oy; }
            }
        }
        if (r == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cl = 4 * q + j;
                sp[cl * (MT + 1) + ml] = best[j];
                sy[cl * (MT + 1) + ml] = by[j];
                sa[cl * (MT + 1) + ml] = __int_as_float(bk[j]);
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ct * MT; t += 256) {
        const int cl = t / MT, ml = t - cl * MT, m = m0 + ml;
        if (m < M) {
            const size_t o = ((size_t)b * C + c0 + cl) * M + m;
            pooled[o] = sp[cl * (MT + 1) + ml];
            ystar[o] = sy[cl * (MT + 1) + ml];
            arg[o] = (unsigned char)__float_as_int(sa[cl * (MT + 1) + ml]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward 1: per (b,c,m): dq = dpooled * [relu: bn(ystar) > 0], scatter into Q[b, idx[b,m,arg], c]; per-channel partial sums
//   {sum dq, sum dq xhat, sum dq dp_j}.  grid (m-tile groups, channel chunks of 128, b)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lagg_bwd_scatter_kernel(int C, int n, int M, int K, int relu,
                                                               const float *__restrict__ dpooled, const float *__restrict__ ystar,
                                                               const unsigned char *__restrict__ arg, const int *__restrict__ idx,
                                                               const float *__restrict__ dp, const float *__restrict__ mean,
                                                               const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta, float *__restrict__ Q,
                                                               double *__restrict__ partial, int nparts_per_b, int tiles_per_wg)
{
    extern __shared__ float lagg_smem[];
    const int ct = min(LAGG_CT, C - (int)blockIdx.y * LAGG_CT);
    float *sd = lagg_smem;                        // [ct][LAGG_MT + 1] dpooled
    float *sy = sd + ct * (LAGG_MT + 1);          // ystar
    float *sa = sy + ct * (LAGG_MT + 1);          // arg
    int *sidx = reinterpret_cast<int *>(sa + ct * (LAGG_MT + 1));  // [LAGG_MT][K]     neighbour indices of the tile's centroids
    float *sdp = reinterpret_cast<float *>(sidx + LAGG_MT * K);    // [3][LAGG_MT][K]  their relative positions
    const int b = blockIdx.z, c0 = blockIdx.y * LAGG_CT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long P = (long)M * K;
    double acc[2][5];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[h][j] = 0.0;
    float mu[2], is[2], ga[2], be[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cl = lane + 64 * h, c = cl < ct ? c0 + cl : c0;
        mu[h] = mean[c]; is[h] = invstd[c]; ga[h] = gamma[c]; be[h] = beta[c];
    }
    for (int tt = 0; tt < tiles_per_wg; ++tt) {
        const int m0 = (blockIdx.x * tiles_per_wg + tt) * LAGG_MT;
        if (m0 >= M) break;
        __syncthreads();
        for (int t = threadIdx.x; t < ct * LAGG_MT; t += 256) {
            const int cl = t / LAGG_MT, ml = t - cl * LAGG_MT, m = m0 + ml;
            float d = 0.f, y = 0.f;
            int a = 0;
            if (m < M) {
                const size_t o = ((size_t)b * C + c0 + cl) * M + m;
                d = dpooled[o]; y = ystar[o]; a = arg[o];
            }
            sd[cl * (LAGG_MT + 1) + ml] = d; sy[cl * (LAGG_MT + 1) + ml] = y; sa[cl * (LAGG_MT + 1) + ml] = __int_as_float(a);
        }
        // the tile's rows of idx and dp (contiguous: LAGG_MT * K positions), read once for all channels -- every routed
        // gradient used to fetch its own index and, behind it, three floats of dp: two dependent round trips per element
        for (int t = threadIdx.x; t < LAGG_MT * K; t += 256) {
            const long pos = (long)m0 * K + t;
            const bool in = pos < P;
            sidx[t] = in ? idx[(size_t)b * P + pos] : 0;
            const size_t pd = (size_t)b * 3 * P + pos;
            sdp[t] = in ? dp[pd] : 0.f;
            sdp[LAGG_MT * K + t] = in ? dp[pd + P] : 0.f;
            sdp[2 * LAGG_MT * K + t] = in ? dp[pd + 2 * P] : 0.f;
        }
        __syncthreads();
        for (int i = 0; i < LAGG_MT / 4; ++i) {
            const int ml = wave * (LAGG_MT / 4) + i, m = m0 + ml;
            if (m >= M) break;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cl = lane + 64 * h;
                if (cl >= ct) continue;
                float dq = sd[cl * (LAGG_MT + 1) + ml];
                const float y = sy[cl * (LAGG_MT + 1) + ml];
                const int k = __float_as_int(sa[cl * (LAGG_MT + 1) + ml]);
                const float xh = __fmul_rn(__fsub_rn(y, mu[h]), is[h]);
                if (relu && !(__fadd_rn(__fmul_rn(xh, ga[h]), be[h]) > 0.f)) dq = 0.f;
                if (dq != 0.f) {
                    const int slot = ml * K + k;
                    const int id = sidx[slot];
                    const double dd = dq;
                    acc[h][0] += dd; acc[h][1] += dd * (double)xh;
                    acc[h][2] += dd * (double)sdp[slot]; acc[h][3] += dd * (double)sdp[LAGG_MT * K + slot];
                    acc[h][4] += dd * (double)sdp[2 * LAGG_MT * K + slot];
                    atomicAdd(Q + ((size_t)b * n + id) * C + c0 + cl, dq);
                }
            }
        }
    }
    __syncthreads();
    double *red = reinterpret_cast<double *>(lagg_smem);  // [4][ct][5] doubles
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (lane + 64 * h < ct) {
#pragma unroll
            for (int j = 0; j < 5; ++j) red[((size_t)wave * ct + lane + 64 * h) * 5 + j] = acc[h][j];
        }
    __syncthreads();
    for (int t = threadIdx.x; t < ct * 5; t += 256) {
        const int cl = t / 5, j = t - cl * 5;
        const double v = (red[((size_t)0 * ct + cl) * 5 + j] + red[((size_t)1 * ct + cl) * 5 + j]) +
                         (red[((size_t)2 * ct + cl) * 5 + j] + red[((size_t)3 * ct + cl) * 5 + j]);
        partial[(((size_t)b * nparts_per_b + blockIdx.x) * C + c0 + cl) * 5 + j] = v;
    }
}

// dgamma, dbeta, dW_dp and the per-channel coefficients of dG: coef[c] = {g*is, ma, mb*is, mu}
// mode 0: one rank.  mode 1: reduce this rank's partials, publish {sum dq, sum dq xhat} to dsums (2C doubles) for the
// all-reduce, keep the five local sums in partial slot 0; mode 2: coefficients from the reduced dsums and the global count
// (*count_dev), dW_dp from the local sums (parameter gradients stay rank-local, as torch's SyncBatchNorm leaves them)
__global__ __launch_bounds__(256) void lagg_bwd_finalize_kernel(int C, int nparts, double count, double *__restrict__ partial,
                                                                const long long *__restrict__ mom, const double *__restrict__ gd,
                                                                const float *__restrict__ w_dp, const float *__restrict__ mean,
                                                                const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                                float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                float *__restrict__ dw_dp, float *__restrict__ coef, int mode,
                                                                double *__restrict__ dsums, const double *__restrict__ count_dev)
{
    __shared__ double red[4][5];
    const int c = blockIdx.x;
    double a[5] = {0, 0, 0, 0, 0};
    if (mode == 2) nparts = 1;  // slot 0 holds this rank's reduced sums
    {   // four partials in flight per thread (the reverse-list collapse writes up to 8192 of them: a chain of 32 dependent
        // round trips per thread otherwise); four interleaved sub-sums added in a fixed order
        double a1[5] = {0, 0, 0, 0, 0}, a2[5] = {0, 0, 0, 0, 0}, a3[5] = {0, 0, 0, 0, 0};
        int k = threadIdx.x;
        for (; k + 768 < nparts; k += 1024) {
            const double *p0 = partial + ((size_t)k * C + c) * 5, *p1 = p0 + (size_t)256 * C * 5, *p2 = p1 + (size_t)256 * C * 5,
                         *p3 = p2 + (size_t)256 * C * 5;
#pragma unroll
            for (int j = 0; j < 5; ++j) { a[j] += p0[j]; a1[j] += p1[j]; a2[j] += p2[j]; a3[j] += p3[j]; }
        }
        for (; k < nparts; k += 256) {
            const double *p = partial + ((size_t)k * C + c) * 5;
#pragma unroll
            for (int j = 0; j < 5; ++j) a[j] += p[j];
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) a[j] = (a[j] + a1[j]) + (a2[j] + a3[j]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        double v = a[j];
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
#pragma unroll
    for (int j = 0; j < 5; ++j) a[j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    double m[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) m[j] = (double)mom[j] * (1.0 / LAGG_FX_M);
    const double S2[3][3] = {{m[3], m[4], m[5]}, {m[4], m[6], m[7]}, {m[5], m[7], m[8]}};
    const double mu = mean[c], is = invstd[c], g = gamma[c];
    if (mode != 2) { dbeta[c] = (float)a[0]; dgamma[c] = (float)a[1]; }
    if (mode == 1) {
#pragma unroll
        for (int j = 0; j < 5; ++j) partial[(size_t)c * 5 + j] = a[j];
        dsums[2 * c] = a[0]; dsums[2 * c + 1] = a[1];
        return;
    }
    double sa = a[0], sb = a[1];
    if (mode == 2) { sa = dsums[2 * c]; sb = dsums[2 * c + 1]; count = *count_dev; }
    const double ma = sa / count, mb = sb / count, gi = g * is;
    const double w[3] = {w_dp[c * 3 + 0], w_dp[c * 3 + 1], w_dp[c * 3 + 2]};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // sum_p xhat_p dp_p[j] = is * ( sum_n G D_j + sum_j' w_j' S2[j'][j] - mu S1[j] )
        const double sxd = is * (gd[c * 3 + j] + w[0] * S2[0][j] + w[1] * S2[1][j] + w[2] * S2[2][j] - mu * m[j]);
        dw_dp[c * 3 + j] = (float)(gi * (a[2 + j] - ma * m[j] - mb * sxd));
    }
    coef[c * 4 + 0] = (float)gi; coef[c * 4 + 1] = (float)ma; coef[c * 4 + 2] = (float)(mb * is); coef[c * 4 + 3] = (float)mu;
}

// ---------------------------------------------------------------------------------------------------------------
// backward 2: dG[b,c,n] = gi * ( Q - cnt*ma - mbis * ( cnt*(G - mu) + W_dp . D ) ), written channel-major
// grid (n tiles of 64, channel chunks of 64, b)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lagg_bwd_apply_kernel(int C, int n, const float *__restrict__ Q, const float *__restrict__ g_pm,
                                                             LaggMoments gm, const float *__restrict__ w_dp,
                                                             const float *__restrict__ coef, float *__restrict__ dg_cm)
{
    __shared__ float tile[64][LAGG_NT + 1];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, n0 = blockIdx.x * LAGG_NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = c0 + lane;
    float gi = 0.f, ma = 0.f, mbis = 0.f, mu = 0.f, w0 = 0.f, w1 = 0.f, w2 = 0.f;
    if (c < C) {
        gi = coef[c * 4 + 0]; ma = coef[c * 4 + 1]; mbis = coef[c * 4 + 2]; mu = coef[c * 4 + 3];
        w0 = w_dp[c * 3 + 0]; w1 = w_dp[c * 3 + 1]; w2 = w_dp[c * 3 + 2];
    }
    for (int i = 0; i < 16; ++i) {
        const int nl = wave * 16 + i, nn = n0 + nl;
        float v = 0.f;
        if (nn < n && c < C) {
            const size_t o = ((size_t)b * n + nn) * C + c;
            const float cn = (float)gm.cnt[(size_t)b * n + nn];
            const long long *df = gm.dfx + ((size_t)b * n + nn) * 3;
            const float dx = (float)((double)df[0] * (1.0 / LAGG_FX_D)), dy = (float)((double)df[1] * (1.0 / LAGG_FX_D)),
                        dz = (float)((double)df[2] * (1.0 / LAGG_FX_D));
            const float wd = w0 * dx + w1 * dy + w2 * dz;
            v = gi * (Q[o] - cn * ma - mbis * (cn * (g_pm[o] - mu) + wd));
        }
        tile[lane][nl] = v;
    }
    __syncthreads();
    for (int r = wave; r < 64; r += 4) {
        const int ch = c0 + r, nn = n0 + lane;
        if (ch < C && nn < n) dg_cm[((size_t)b * C + ch) * n + nn] = tile[r][lane];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// First layer of a multi-layer SetAbstraction MLP (PointNeXt-S: sa_layers = 2): the same convolve-before-gather
// conv + BatchNorm + ReLU, but the activation x1 (B,C,M,K) is materialised for the layers that follow
//   expand    x1[b,c,m,k] = [relu](bn(G[b, idx[b,m,k], c] + W_dp[c] . dp[b,:,m,k]))       one gather pass, one write
//   collapse  from dx1 (B,C,M,K): d = dx1 * [relu mask], scattered into Q[b, idx, c] (row-contiguous float atomics, zeros
//             skipped) + the per-channel sums {sum d, sum d xhat, sum d dp_j}: BatchNorm's backward then needs nothing
//             else of size M*K (lagg_bwd_finalize / lagg_bwd_apply turn Q into dG on the N source points).
// One pass over dx1 replaces BN-backward statistics, BN-backward apply, the conv's backward-data product + scatter and
// its backward-weight product.  grid (position-tile groups, channel chunks of 64, b); a tile = 128 positions.
// ---------------------------------------------------------------------------------------------------------------
constexpr int LAGG_PT = 128;  // positions per tile
constexpr int LAGG_XC = 64;   // channels per workgroup

template <int LPR>
__global__ __launch_bounds__(256) void lagg_expand_kernel(int C, int n, long P, int relu, const float *__restrict__ g_pm,
                                                          const int *__restrict__ idx, const float *__restrict__ dp,
                                                          const float *__restrict__ w_dp, const float *__restrict__ mean,
                                                          const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ x1)
{
    __shared__ float tile[LAGG_XC][LAGG_PT + 1];  // odd stride: the (channel quad, position) writes are at most 2-way
    const int b = blockIdx.z, c0 = blockIdx.y * LAGG_XC;
    const int ct = min(LAGG_XC, C - c0);
    const long p0 = (long)blockIdx.x * LAGG_PT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int rpi = 64 / LPR;
    const int q = lane % LPR, r = lane / LPR;
    const int cq = c0 + 4 * q;
    float w[4][3], mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = cq + j;
        w[j][0] = w_dp[c * 3 + 0]; w[j][1] = w_dp[c * 3 + 1]; w[j][2] = w_dp[c * 3 + 2];
        mu[j] = mean[c]; is[j] = invstd[c]; ga[j] = gamma[c]; be[j] = beta[c];
    }
    // this wave's 32 positions: lane l < 32 fetches index and dp of position wave*32 + l
    const long pw = p0 + wave * 32;
    int id_l = 0;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (lane < 32 && pw + lane < P) {
        id_l = idx[(size_t)b * P + pw + lane];
        d0 = dp[((size_t)b * 3 + 0) * P + pw + lane]; d1 = dp[((size_t)b * 3 + 1) * P + pw + lane]; d2 = dp[((size_t)b * 3 + 2) * P + pw + lane];
    }
#pragma unroll
    for (int k0 = 0; k0 < 32; k0 += rpi) {
        const int k = k0 + r;
        const int id = __shfl(id_l, k, 64);
        const float e0 = __shfl(d0, k, 64), e1 = __shfl(d1, k, 64), e2 = __shfl(d2, k, 64);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pw + k < P) g = *reinterpret_cast<const float4 *>(g_pm + ((size_t)b * n + id) * C + cq);
        const float gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float y = __fmaf_rn(w[j][0], e0, __fmaf_rn(w[j][1], e1, __fmaf_rn(w[j][2], e2, gs[j])));
            float v = lagg_bn(y, mu[j], is[j], ga[j], be[j]);
            if (relu) v = fmaxf(v, 0.f);
            tile[4 * q + j][wave * 32 + k] = v;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ct * LAGG_PT; t += 256) {  // a wave stores 256 contiguous bytes of one channel row
        const int cl = t / LAGG_PT, pp = t - cl * LAGG_PT;
        if (p0 + pp < P) x1[((size_t)b * C + c0 + cl) * P + p0 + pp] = tile[cl][pp];
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void lagg_collapse_kernel(int C, int n, long P, int relu, const float *__restrict__ dx1,
                                                            const float *__restrict__ g_pm, const int *__restrict__ idx,
                                                            const float *__restrict__ dp, const float *__restrict__ w_dp,
                                                            const float *__restrict__ mean, const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            float *__restrict__ Q, double *__restrict__ partial, int nparts_per_b,
                                                            int tiles_per_wg)
{
    __shared__ float tile[LAGG_XC][LAGG_PT + 1];
    double(*red)[LAGG_XC][5] = reinterpret_cast<double(*)[LAGG_XC][5]>(&tile[0][0]);  // reused after the last tile (10 KiB of 33)
    constexpr int rpi = 64 / LPR, CT = 4 * LPR;  // channels of this workgroup's chunk
    const int b = blockIdx.z, c0 = blockIdx.y * LAGG_XC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % LPR, r = lane / LPR;
    const int cq = c0 + 4 * q;
    float w[4][3], mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = cq + j;
        w[j][0] = w_dp[c * 3 + 0]; w[j][1] = w_dp[c * 3 + 1]; w[j][2] = w_dp[c * 3 + 2];
        mu[j] = mean[c]; is[j] = invstd[c]; ga[j] = gamma[c]; be[j] = beta[c];
    }
    double acc[4][5];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int v = 0; v < 5; ++v) acc[j][v] = 0.0;
    for (int tt = 0; tt < tiles_per_wg; ++tt) {
        const long p0 = ((long)blockIdx.x * tiles_per_wg + tt) * LAGG_PT;
        if (p0 >= P) break;
        __syncthreads();
        for (int t = threadIdx.x; t < CT * LAGG_PT; t += 256) {
            const int cl = t / LAGG_PT, pp = t - cl * LAGG_PT;
            tile[cl][pp] = (p0 + pp < P) ? dx1[((size_t)b * C + c0 + cl) * P + p0 + pp] : 0.f;
        }
        __syncthreads();
        const long pw = p0 + wave * 32;
        int id_l = 0;
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
        if (lane < 32 && pw + lane < P) {
            id_l = idx[(size_t)b * P + pw + lane];
            d0 = dp[((size_t)b * 3 + 0) * P + pw + lane]; d1 = dp[((size_t)b * 3 + 1) * P + pw + lane]; d2 = dp[((size_t)b * 3 + 2) * P + pw + lane];
        }
        // phase A: 16-byte gathers of the G rows; mask the ReLU, accumulate the sums, leave the masked gradient in the tile
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += rpi) {
            const int k = k0 + r;
            const int id = __shfl(id_l, k, 64);
            const float e0 = __shfl(d0, k, 64), e1 = __shfl(d1, k, 64), e2 = __shfl(d2, k, 64);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pw + k < P) g = *reinterpret_cast<const float4 *>(g_pm + ((size_t)b * n + id) * C + cq);
            const float gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = __fmaf_rn(w[j][0], e0, __fmaf_rn(w[j][1], e1, __fmaf_rn(w[j][2], e2, gs[j])));
                const float xh = __fmul_rn(__fsub_rn(y, mu[j]), is[j]);
                float d = tile[4 * q + j][wave * 32 + k];
                if (relu && !(__fadd_rn(__fmul_rn(xh, ga[j]), be[j]) > 0.f)) d = 0.f;
                tile[4 * q + j][wave * 32 + k] = d;  // (this lane is the only reader and writer of the element)
                const double dd = d;
                acc[j][0] += dd; acc[j][1] += dd * (double)xh;
                acc[j][2] += dd * (double)e0; acc[j][3] += dd * (double)e1; acc[j][4] += dd * (double)e2;
            }
        }
        // phase B: scatter with one lane per channel: a wave-instruction adds 64 / CT whole rows of CT contiguous floats
        // (each wave reads back only what it wrote itself: its own 32 positions)
        constexpr int ppi = 64 / CT > 0 ? 64 / CT : 1;  // positions per instruction
        const int cl = lane % CT, sub = lane / CT;
        for (int k0 = 0; k0 < 32; k0 += ppi) {
            const int k = k0 + sub;
            const int id = __shfl(id_l, k & 31, 64);
            if (sub < ppi && pw + k < P) {
                const float d = tile[cl][wave * 32 + k];
                if (d != 0.f) atomicAdd(Q + ((size_t)b * n + id) * C + c0 + cl, d);
            }
        }
    }
    // combine the row slots (lanes q, q + LPR, ...), then the four waves
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            double x = acc[j][v];
#pragma unroll
            for (int s = LPR; s < 64; s <<= 1) x += __shfl_xor(x, s, 64);
            acc[j][v] = x;
        }
    __syncthreads();  // the tile is no longer read: its memory holds the cross-wave reduction
    if (r == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 5; ++v) red[wave][4 * q + j][v] = acc[j][v];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < CT * 5; t += 256) {
        const int cl2 = t / 5, v = t - cl2 * 5;
        partial[(((size_t)b * nparts_per_b + blockIdx.x) * C + c0 + cl2) * 5 + v] =
            (red[0][cl2][v] + red[1][cl2][v]) + (red[2][cl2][v] + red[3][cl2][v]);
    }
}

static bool lagg_supported(int C, int K)
{
    if (C < 8 || C % 4 || K < 1 || K > 64) return false;
    const int ct = C < LAGG_CT ? C : LAGG_CT;
    const int lpr = ct / 4;
    return (lpr & (lpr - 1)) == 0 && C % ct == 0;  // 8,16,32,64,128 and multiples of 128
}

// tiles per workgroup of the statistics kernel: one while that still leaves the grid small (latency-bound kernel)
static int lagg_stats_tiles(int b, int C, int n)
{
    const long wgs1 = (long)b * div_up(n, LAGG_NT) * div_up(C, 64);
    return wgs1 <= 8192 ? 1 : LAGG_TILES;
}

// ... and of the backward scatter kernel (its workgroups walk their centroids serially: a few dozen of them took 50-100 us
// on the coarse stages of PointNeXt-L)
static int lagg_scatter_tiles(int b, int C, int M)
{
    static const bool off = getenv("AMC3D_LAGG_SCATTER_TILES4") != nullptr;
    const long wgs1 = (long)b * div_up(M, LAGG_MT) * div_up(C, LAGG_CT);
    return (wgs1 <= 8192 && !off) ? 1 : LAGG_TILES;
}

static size_t lagg_partial_bytes(int b, int C, int n, int M)
{
    const size_t pf = (size_t)b * div_up(div_up(n, LAGG_NT), lagg_stats_tiles(b, C, n)),
                 pb = (size_t)b * div_up(div_up(M, LAGG_MT), lagg_scatter_tiles(b, C, M));
    const size_t pc = (size_t)b * div_up((long)M * 32, LAGG_PT);  // collapse kernel (K = 32), at most one partial per tile
    size_t m = pf > pb ? pf : pb;
    if (pc > m) m = pc;
    return m * C * 5 * sizeof(double);
}

// the materialising first layer: K = 32, channel chunks of 64 (or the whole of C in {8, 16, 32})
static bool lagg_expand_supported(int C, int K)
{
    if (K != 32 || C < 8 || C % 4) return false;
    const int ct = C < LAGG_XC ? C : LAGG_XC;
    const int lpr = ct / 4;
    return (lpr & (lpr - 1)) == 0 && C % ct == 0;
}

}  // namespace amc

using namespace amc;

AMC_API int amc3d_local_aggregation_supported(int cout, int nsample) { return lagg_supported(cout, nsample) ? 1 : 0; }

AMC_API size_t amc3d_group_moments_bytes(int b, int n) { return (b <= 0 || n <= 0) ? 0 : lagg_moments_bytes(b, n); }

// moments (opaque, amc3d_group_moments_bytes): in-degree and dp sum of every support point, global dp moments
AMC_API int amc3d_group_moments(int b, int n, int npoints, int nsample, const int *idx, const float *dp, void *moments,
                                size_t moments_bytes, void *stream_)
{
    if (b <= 0 || n <= 0) return 0;
    if (!idx || !dp || !moments || moments_bytes < lagg_moments_bytes(b, n) || npoints <= 0 || nsample <= 0)
        return bad_arg("amc3d_group_moments: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    if (int st = fill_i32((int *)moments, 0, lagg_moments_bytes(b, n) / 4, stream)) return st;
    const LaggMoments v = lagg_views(moments, b, n);
    const long P = (long)npoints * nsample;
    const int blocks = (int)(div_up(P, 256 * 4) < 1024 ? div_up(P, 256 * 4) : 1024);
    hipLaunchKernelGGL(lagg_geom_kernel, dim3(blocks, b), dim3(256), 0, stream, n, P, idx, dp, (int *)v.cnt,
                       (unsigned long long *)v.dfx, (unsigned long long *)v.mom);
    return launch_status("amc3d_group_moments");
}

AMC_API size_t amc3d_local_aggregation_workspace_bytes(int b, int cout, int n, int npoints)
{
    if (b <= 0 || cout <= 0 || n <= 0 || npoints <= 0) return 0;
    // partial sums | Q (b, n, cout) | coef (cout, 4)
    return lagg_partial_bytes(b, cout, n, npoints) + (size_t)b * n * cout * sizeof(float) + (size_t)cout * 4 * sizeof(float) + 64;
}

// g_cm (b,cout,n) = W_f . f, computed by the caller (amc3d_pointwise_conv_forward).  Outputs: g_pm (b,n,cout) the
// point-major copy (kept for backward), pooled / ystar (b,cout,npoints) fp32, arg (b,cout,npoints) bytes, mean / invstd /
// var_unbiased (cout), gd (cout,3) doubles.  training == 0: mean / invstd are INPUTS (running statistics), no statistics pass.
AMC_API int amc3d_local_aggregation_forward(int b, int cout, int n, int npoints, int nsample, int training, int relu, float eps,
                                            float momentum, const float *g_cm, const int *idx, const float *dp,
                                            const float *w_dp, const void *moments, const float *gamma, const float *beta,
                                            float *g_pm, float *pooled, unsigned char *arg, float *ystar, float *mean,
                                            float *invstd, float *var_unbiased, double *gd, float *running_mean,
                                            float *running_var, long long *num_batches_tracked, int phase, double *sums,
                                            void *workspace, size_t workspace_bytes, void *stream_)
{
    if (training && phase != 0 && !sums) return bad_arg("amc3d_local_aggregation_forward: phase 1 / 2 need the sums buffer");
    if (b <= 0 || npoints <= 0) return 0;
    if (!lagg_supported(cout, nsample) || n <= 0 || !g_cm || !idx || !dp || !w_dp || !gamma || !beta || !g_pm || !pooled || !arg ||
        !ystar || !mean || !invstd || (training && (!moments || !var_unbiased || !gd || !workspace ||
        workspace_bytes < amc3d_local_aggregation_workspace_bytes(b, cout, n, npoints))) ||
        (running_mean && (!running_var || !num_batches_tracked)))
        return bad_arg("amc3d_local_aggregation_forward: unsupported shape, null pointer or workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    LaggMoments gm{};
    if (moments) gm = lagg_views(moments, b, n);
    const int stiles = lagg_stats_tiles(b, cout, n);
    const int nparts_b = div_up(div_up(n, LAGG_NT), stiles);
    double *partial = (double *)workspace;
    if (training) {
        if (phase != 2)
            hipLaunchKernelGGL(lagg_stats_kernel, dim3(nparts_b, div_up(cout, 64), b), dim3(256), 0, stream, cout, n, g_cm, g_pm, gm,
                               partial, nparts_b, stiles);
        hipLaunchKernelGGL(lagg_stats_finalize_kernel, dim3(cout), dim3(256), 0, stream, cout, nparts_b * b,
                           (double)b * (double)npoints * (double)nsample, eps, momentum, (const double *)partial, gm.mom, w_dp,
                           mean, invstd, var_unbiased, gd, running_mean, running_var, num_batches_tracked, phase, sums);
        if (phase == 1) return launch_status("amc3d_local_aggregation_forward");
    } else {
        // eval mode: only the point-major copy is needed
        if (int st = amc3d_transpose_cn(b, cout, n, g_cm, g_pm, stream_)) return st;
    }
    const int ct = cout < LAGG_CT ? cout : LAGG_CT;
    // centroids per wave: 8, fewer where that leaves the chip with under ~2 workgroups per CU (the coarse stages)
    int cpw = 8;
    while (cpw > 1 && (long)div_up(npoints, 4 * cpw) * (cout / ct) * b < 1024) cpw >>= 1;
    const size_t lds = (size_t)3 * ct * (4 * cpw + 1) * sizeof(float);
#define AMC_POOL(L)                                                                                                            \
    hipLaunchKernelGGL(lagg_pool_kernel<L>, dim3(div_up(npoints, 4 * cpw), cout / ct, b), dim3(256), lds, stream, cout, n, npoints, \
                       nsample, cpw, relu, (const float *)g_pm, idx, dp, w_dp, (const float *)mean, (const float *)invstd, gamma,  \
                       beta, pooled, arg, ystar)
    switch (ct / 4) { case 2: AMC_POOL(2); break; case 4: AMC_POOL(4); break; case 8: AMC_POOL(8); break; case 16: AMC_POOL(16); break;
                      default: AMC_POOL(32); }
#undef AMC_POOL
    return launch_status("amc3d_local_aggregation_forward");
}

// dg_cm (b,cout,n): gradient w.r.t. G = W_f . f (the caller runs the pointwise conv's backward on it);
// dw_dp (cout,3), dgamma, dbeta (cout)
AMC_API int amc3d_local_aggregation_backward(int b, int cout, int n, int npoints, int nsample, int relu, const float *dpooled,
                                             const float *ystar, const unsigned char *arg, const float *g_pm, const int *idx,
                                             const float *dp, const float *w_dp, const void *moments, const double *gd,
                                             const float *mean, const float *invstd, const float *gamma, const float *beta,
                                             float *dg_cm, float *dw_dp, float *dgamma, float *dbeta, int phase, double *dsums,
                                             const double *count_dev, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (b <= 0 || npoints <= 0) return 0;
    if (phase != 0 && (!dsums || (phase == 2 && !count_dev))) return bad_arg("amc3d_local_aggregation_backward: phase 1 / 2 need dsums (and the count)");
    if (!lagg_supported(cout, nsample) || n <= 0 || !dpooled || !ystar || !arg || !g_pm || !idx || !dp || !w_dp || !moments || !gd ||
        !mean || !invstd || !gamma || !beta || !dg_cm || !dw_dp || !dgamma || !dbeta || !workspace ||
        workspace_bytes < amc3d_local_aggregation_workspace_bytes(b, cout, n, npoints))
        return bad_arg("amc3d_local_aggregation_backward: unsupported shape, null pointer or workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const LaggMoments gm = lagg_views(moments, b, n);
    double *partial = (double *)workspace;
    float *Q = (float *)((char *)workspace + lagg_partial_bytes(b, cout, n, npoints));
    float *coef = Q + (size_t)b * n * cout;
    const int ct = cout < LAGG_CT ? cout : LAGG_CT;
    const int stiles = lagg_scatter_tiles(b, cout, npoints);
    const int nparts_b = div_up(div_up(npoints, LAGG_MT), stiles);
    if (phase != 2) {
        if (int st = fill_i32((int *)Q, 0, (size_t)b * n * cout, stream)) return st;
        size_t lds = ((size_t)3 * ct * (LAGG_MT + 1) + (size_t)4 * LAGG_MT * nsample) * sizeof(float);
        const size_t red = (size_t)4 * ct * 5 * sizeof(double);
        if (lds < red) lds = red;
        if (lds > 65536)  // (128 channels x 33 x 3 images + the idx / dp rows: 67 KB; gfx950 has 160 KB per workgroup)
            (void)hipFuncSetAttribute((const void *)lagg_bwd_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lagg_bwd_scatter_kernel, dim3(nparts_b, cout / ct, b), dim3(256), lds, stream, cout, n, npoints, nsample,
                           relu, dpooled, ystar, arg, idx, dp, mean, invstd, gamma, beta, Q, partial, nparts_b, stiles);
    }
    hipLaunchKernelGGL(lagg_bwd_finalize_kernel, dim3(cout), dim3(256), 0, stream, cout, nparts_b * b,
                       (double)b * (double)npoints * (double)nsample, partial, gm.mom, gd, w_dp, mean, invstd,
                       gamma, dgamma, dbeta, dw_dp, coef, phase, dsums, count_dev);
    if (phase == 1) return launch_status("amc3d_local_aggregation_backward");
    hipLaunchKernelGGL(lagg_bwd_apply_kernel, dim3(div_up(n, LAGG_NT), div_up(cout, 64), b), dim3(256), 0, stream, cout, n,
                       (const float *)Q, g_pm, gm, w_dp, (const float *)coef, dg_cm);
    return launch_status("amc3d_local_aggregation_backward");
}

// ---- first layer of a multi-layer SetAbstraction MLP: conv (before the gather) + BatchNorm + ReLU, x1 materialised ----
AMC_API int amc3d_grouped_conv_bn_supported(int cout, int nsample) { return lagg_expand_supported(cout, nsample) ? 1 : 0; }

// x1 (b,cout,npoints,32) = [relu](bn(G[idx] + W_dp . dp)); other arguments as amc3d_local_aggregation_forward
AMC_API int amc3d_grouped_conv_bn_forward(int b, int cout, int n, int npoints, int nsample, int training, int relu, float eps,
                                          float momentum, const float *g_cm, const int *idx, const float *dp, const float *w_dp,
                                          const void *moments, const float *gamma, const float *beta, float *g_pm, float *x1,
                                          float *mean, float *invstd, float *var_unbiased, double *gd, float *running_mean,
                                          float *running_var, long long *num_batches_tracked, int phase, double *sums,
                                          void *workspace, size_t workspace_bytes, void *stream_)
{
    if (training && phase != 0 && !sums) return bad_arg("amc3d_grouped_conv_bn_forward: phase 1 / 2 need the sums buffer");
    if (b <= 0 || npoints <= 0) return 0;
    if (!lagg_expand_supported(cout, nsample) || n <= 0 || !g_cm || !idx || !dp || !w_dp || !gamma || !beta || !g_pm || !x1 ||
        !mean || !invstd || (training && (!moments || !var_unbiased || !gd || !workspace ||
        workspace_bytes < amc3d_local_aggregation_workspace_bytes(b, cout, n, npoints))) ||
        (running_mean && (!running_var || !num_batches_tracked)))
        return bad_arg("amc3d_grouped_conv_bn_forward: unsupported shape, null pointer or workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    LaggMoments gm{};
    if (moments) gm = lagg_views(moments, b, n);
    const int stiles = lagg_stats_tiles(b, cout, n);
    const int nparts_b = div_up(div_up(n, LAGG_NT), stiles);
    if (training) {
        if (phase != 2)
            hipLaunchKernelGGL(lagg_stats_kernel, dim3(nparts_b, div_up(cout, 64), b), dim3(256), 0, stream, cout, n, g_cm, g_pm, gm,
                               (double *)workspace, nparts_b, stiles);
        hipLaunchKernelGGL(lagg_stats_finalize_kernel, dim3(cout), dim3(256), 0, stream, cout, nparts_b * b,
                           (double)b * (double)npoints * (double)nsample, eps, momentum, (const double *)workspace, gm.mom, w_dp,
                           mean, invstd, var_unbiased, gd, running_mean, running_var, num_batches_tracked, phase, sums);
        if (phase == 1) return launch_status("amc3d_grouped_conv_bn_forward");
    } else if (int st = amc3d_transpose_cn(b, cout, n, g_cm, g_pm, stream_)) {
        return st;
    }
    const long P = (long)npoints * nsample;
    const int ct = cout < LAGG_XC ? cout : LAGG_XC;
#define AMC_EXPAND(L)                                                                                                          \
    hipLaunchKernelGGL(lagg_expand_kernel<L>, dim3(div_up(P, LAGG_PT), cout / ct, b), dim3(256), 0, stream, cout, n, P, relu, \
                       (const float *)g_pm, idx, dp, w_dp, (const float *)mean, (const float *)invstd, gamma, beta, x1)
    switch (ct / 4) { case 2: AMC_EXPAND(2); break; case 4: AMC_EXPAND(4); break; case 8: AMC_EXPAND(8); break; default: AMC_EXPAND(16); }
#undef AMC_EXPAND
    return launch_status("amc3d_grouped_conv_bn_forward");
}

// from dx1 (b,cout,npoints,32): dg_cm (b,cout,n), dw_dp (cout,3), dgamma, dbeta
AMC_API int amc3d_grouped_conv_bn_backward(int b, int cout, int n, int npoints, int nsample, int relu, const float *dx1,
                                           const float *g_pm, const int *idx, const float *dp, const float *w_dp,
                                           const void *moments, const double *gd, const float *mean, const float *invstd,
                                           const float *gamma, const float *beta, float *dg_cm, float *dw_dp, float *dgamma,
                                           float *dbeta, int phase, double *dsums, const double *count_dev, void *workspace,
                                           size_t workspace_bytes, void *stream_)
{
    if (b <= 0 || npoints <= 0) return 0;
    if (phase != 0 && (!dsums || (phase == 2 && !count_dev))) return bad_arg("amc3d_grouped_conv_bn_backward: phase 1 / 2 need dsums (and the count)");
    if (!lagg_expand_supported(cout, nsample) || n <= 0 || !dx1 || !g_pm || !idx || !dp || !w_dp || !moments || !gd || !mean ||
        !invstd || !gamma || !beta || !dg_cm || !dw_dp || !dgamma || !dbeta || !workspace ||
        workspace_bytes < amc3d_local_aggregation_workspace_bytes(b, cout, n, npoints))
        return bad_arg("amc3d_grouped_conv_bn_backward: unsupported shape, null pointer or workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const LaggMoments gm = lagg_views(moments, b, n);
    double *partial = (double *)workspace;
    float *Q = (float *)((char *)workspace + lagg_partial_bytes(b, cout, n, npoints));
    float *coef = Q + (size_t)b * n * cout;
    const long P = (long)npoints * nsample;
    const int ct = cout < LAGG_XC ? cout : LAGG_XC;
    const int ctiles = 2;  // tiles per workgroup (1, 2, 4 measured alike: the kernel runs at the float-atomic rate)
    const int nparts_b = div_up(div_up(P, LAGG_PT), ctiles);
    if (phase != 2) {
    if (int st = fill_i32((int *)Q, 0, (size_t)b * n * cout, stream)) return st;
#define AMC_COLLAPSE(L)                                                                                                        \
    hipLaunchKernelGGL(lagg_collapse_kernel<L>, dim3(nparts_b, cout / ct, b), dim3(256), 0, stream, cout, n, P, relu, dx1, g_pm,  \
                       idx, dp, w_dp, mean, invstd, gamma, beta, Q, partial, nparts_b, ctiles)
    switch (ct / 4) { case 2: AMC_COLLAPSE(2); break; case 4: AMC_COLLAPSE(4); break; case 8: AMC_COLLAPSE(8); break; default: AMC_COLLAPSE(16); }
#undef AMC_COLLAPSE
    }
    hipLaunchKernelGGL(lagg_bwd_finalize_kernel, dim3(cout), dim3(256), 0, stream, cout, nparts_b * b,
                       (double)b * (double)npoints * (double)nsample, partial, gm.mom, gd, w_dp, mean, invstd,
                       gamma, dgamma, dbeta, dw_dp, coef, phase, dsums, count_dev);
    if (phase == 1) return launch_status("amc3d_grouped_conv_bn_backward");
    hipLaunchKernelGGL(lagg_bwd_apply_kernel, dim3(div_up(n, LAGG_NT), div_up(cout, 64), b), dim3(256), 0, stream, cout, n,
                       (const float *)Q, g_pm, gm, w_dp, (const float *)coef, dg_cm);
    return launch_status("amc3d_grouped_conv_bn_backward");
}

// the same backward without float atomics: rev_start / rev_edge are the reverse lists of amc3d_group_csr; dx1 is first
// transposed to position-major (workspace), then every source point sums its incoming positions in list order
// (csrc/csr.hip): deterministic, and ~2.5x faster than the atomic scatter.  workspace: amc3d_grouped_conv_bn_csr_workspace_bytes
namespace amc {
int csr_collapse(int b, int cout, int n, int npoints, int nsample, int relu, const float *dx1_pm, const float *g_pm,
                 const int *rev_start, const int *rev_edge, const float *rev_dp, const float *dp, const float *w_dp,
                 const float *mean, const float *invstd, const float *gamma, const float *beta, float *Q, double *partial,
                 int *nparts, hipStream_t stream);
size_t csr_partials(int b, int cout, int n);
}

AMC_API size_t amc3d_grouped_conv_bn_csr_workspace_bytes(int b, int cout, int n, int npoints, int nsample)
{
    if (b <= 0 || cout <= 0 || n <= 0 || npoints <= 0) return 0;
    const size_t parts = csr_partials(b, cout, n);
    // partial sums | Q (b, n, cout) | coef (cout, 4) | dx1 position-major (b, P, cout)
    return parts * cout * 5 * sizeof(double) + (size_t)b * n * cout * sizeof(float) + (size_t)cout * 4 * sizeof(float) + 256 +
           (size_t)b * npoints * nsample * cout * sizeof(float);
}

AMC_API int amc3d_grouped_conv_bn_backward_csr(int b, int cout, int n, int npoints, int nsample, int relu, const float *dx1,
                                               int dx1_position_major, const float *g_pm, const int *rev_start, const int *rev_edge,
                                               const float *rev_dp, const float *dp,
                                               const float *w_dp, const void *moments, const double *gd, const float *mean,
                                               const float *invstd, const float *gamma, const float *beta, float *dg_cm,
                                               float *dw_dp, float *dgamma, float *dbeta, int phase, double *dsums,
                                               const double *count_dev, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (b <= 0 || npoints <= 0) return 0;
    if (phase != 0 && (!dsums || (phase == 2 && !count_dev))) return bad_arg("amc3d_grouped_conv_bn_backward_csr: phase 1 / 2 need dsums (and the count)");
    if (!lagg_expand_supported(cout, nsample) || n <= 0 || !dx1 || !g_pm || !rev_start || !rev_edge || !dp || !w_dp || !moments ||
        !gd || !mean || !invstd || !gamma || !beta || !dg_cm || !dw_dp || !dgamma || !dbeta || !workspace ||
        workspace_bytes < amc3d_grouped_conv_bn_csr_workspace_bytes(b, cout, n, npoints, nsample))
        return bad_arg("amc3d_grouped_conv_bn_backward_csr: unsupported shape, null pointer or workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const LaggMoments gm = lagg_views(moments, b, n);
    const size_t parts = csr_partials(b, cout, n);
    char *w = (char *)workspace;
    double *partial = (double *)w; w += parts * cout * 5 * sizeof(double);
    float *Q = (float *)w; w += (size_t)b * n * cout * sizeof(float);
    float *coef = (float *)w; w += (size_t)cout * 4 * sizeof(float);
    float *dx1_pm = (float *)(((uintptr_t)w + 255) & ~(uintptr_t)255);
    const long P = (long)npoints * nsample;
    // (b, cout, P) -> (b, P, cout)
    if (P >= (1L << 31)) return bad_arg("amc3d_grouped_conv_bn_backward_csr: too many positions");
    int nparts = (int)parts;
    if (phase != 2) {
        if (dx1_position_major) {
            if ((uintptr_t)dx1 & 15) return bad_arg("amc3d_grouped_conv_bn_backward_csr: position-major dx1 must be 16-byte aligned");
            dx1_pm = const_cast<float *>(dx1);  // read only
        } else if (int st = amc3d_transpose_cn(b, cout, (int)P, dx1, dx1_pm, stream_)) return st;
        if (rev_dp && ((uintptr_t)rev_dp & 15)) return bad_arg("amc3d_grouped_conv_bn_backward_csr: rev_dp must be 16-byte aligned");
        if (int st = csr_collapse(b, cout, n, npoints, nsample, relu, dx1_pm, g_pm, rev_start, rev_edge, rev_dp, dp, w_dp, mean, invstd,
                                  gamma, beta, Q, partial, &nparts, stream))
            return st;
    }
    hipLaunchKernelGGL(lagg_bwd_finalize_kernel, dim3(cout), dim3(256), 0, stream, cout, nparts,
                       (double)b * (double)npoints * (double)nsample, partial, gm.mom, gd, w_dp, mean, invstd,
                       gamma, dgamma, dbeta, dw_dp, coef, phase, dsums, count_dev);
    if (phase == 1) return launch_status("amc3d_grouped_conv_bn_backward_csr");
    hipLaunchKernelGGL(lagg_bwd_apply_kernel, dim3(div_up(n, LAGG_NT), div_up(cout, 64), b), dim3(256), 0, stream, cout, n,
                       (const float *)Q, g_pm, gm, w_dp, (const float *)coef, dg_cm);
    return launch_status("amc3d_grouped_conv_bn_backward_csr");
}
