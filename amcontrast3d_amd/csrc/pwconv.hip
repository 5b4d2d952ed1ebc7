// Pointwise (1x1) convolution on fp32 MFMA for gfx950: forward, backward-data, backward-weight.
//
// Reference: every Conv1d / Conv2d of the path has kernel size 1 (models/layers/conv.py:8-21, used by
// pointnext_AA.py:104-127 SetAbstraction, :196-208 FeaturePropogation, base_seg.py:236-252 SegHead), i.e.
//     Y[b][co][p] = sum_ci W[co][ci] * X[b][ci][p] (+ bias[co])
// on channel-major (B, C, P) tensors, P = points (Conv1d) or points x neighbours (Conv2d).  The reference
// hands these to cuDNN; MIOpen's answer on gfx950 is NCHW<->NHWC transposes around igemm kernels (the
// weight gradient of a 32-channel layer costs three launches and ~100 us).  The three products are small-K
// GEMMs against a tall operand, HBM-bound for the wide-P layers and MFMA-bound for the deep ones:
//
//  * pw_gemm_kernel      Y = A . X per cloud with A = W (forward) or A = W^T (backward-data, strides
//                        swapped): 128-position tiles, the K chunk of A and X staged in LDS, one
//                        v_mfma_f32_32x32x2_f32 per (32 channels x 32 positions x 2 k).
//  * pw_wgrad_kernel     dW = sum_{b,p} dY X^T: the position axis is the reduction, so both operands go
//                        through a padded LDS transpose; a workgroup owns a (<=128 x <=64) block of dW over a
//                        contiguous run of position tiles and writes a partial, which are summed in a fixed
//                        order (deterministic, no float atomics).
#include "common.h"

namespace amc {

typedef float pw_f32x16 __attribute__((ext_vector_type(16)));

constexpr int PW_TP = 128;          // positions per workgroup tile (4 waves x 32)
constexpr int PW_KC = 32;           // reduction chunk staged in LDS
constexpr int PW_LDA = PW_KC + 1;   // odd row stride: the A operand is read down a column
constexpr int PWW_TP = 64;          // positions per tile in the weight-gradient kernel
constexpr int PWW_LD = PWW_TP + 1;

int reduce_partials(int total, int nparts, const float *partial, float *out, hipStream_t stream);  // gcc.hip

// Y[b][m][p] = sum_k A(m,k) X[b][k][p] (+ bias[m]),  A(m,k) = a[m*sam + k*sak]
template <int NCT>
__global__ __launch_bounds__(256) void pw_gemm_kernel(int M, int K, long P, const float *__restrict__ a, long sam,
                                                      long sak, const float *__restrict__ bias,
                                                      const float *__restrict__ x, float *__restrict__ y, int vec)
{
    __shared__ float as[NCT * 32 * PW_LDA];
    __shared__ __attribute__((aligned(16))) float xs[PW_KC * PW_TP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pl = lane & 31, kh = lane >> 5;
    const long p0 = (long)blockIdx.x * PW_TP;
    const int m0 = blockIdx.y * NCT * 32;
    const float *X = x + (size_t)blockIdx.z * K * P;
    float *Y = y + (size_t)blockIdx.z * M * P;
    pw_f32x16 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[c] = pw_f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // vec path: the X chunk of the NEXT k-step is loaded into registers before the current one is multiplied (layers with more
    // than 32 input channels walk 2-4 chunks per tile: load -> barrier -> multiply left every load exposed)
    constexpr int LPT = PW_KC * PW_TP / 4 / 256;  // 4 float4 per thread and chunk
    float4 pre[LPT];
    auto prefetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            const int i = threadIdx.x + u * 256;
            const int kk = i / (PW_TP / 4), c4 = i - kk * (PW_TP / 4);
            const int k = k0 + kk;
            const long p = p0 + c4 * 4;
            pre[u] = (k < K && p < P) ? *(const float4 *)(X + (size_t)k * P + p) : make_float4(0.f, 0.f, 0.f, 0.f);  // P % 4 == 0
        }
    };
    if (vec) prefetch(0);
    for (int k0 = 0; k0 < K; k0 += PW_KC) {
        __syncthreads();  // previous chunk consumed
        // A chunk (NCT*32 rows x 32 k); consecutive threads walk the contiguous axis of a
        for (int i = threadIdx.x; i < NCT * 32 * PW_KC; i += 256) {
            int r, kk;
            if (sak == 1) { r = i / PW_KC; kk = i - r * PW_KC; } else { kk = i / (NCT * 32); r = i - kk * (NCT * 32); }
            const int m = m0 + r, k = k0 + kk;
            as[r * PW_LDA + kk] = (m < M && k < K) ? a[(long)m * sam + (long)k * sak] : 0.f;
        }
        // X chunk (32 k x 128 positions)
        if (vec) {
#pragma unroll
            for (int u = 0; u < LPT; ++u) {
                const int i = threadIdx.x + u * 256;
                const int kk = i / (PW_TP / 4), c4 = i - kk * (PW_TP / 4);
                *(float4 *)(xs + kk * PW_TP + c4 * 4) = pre[u];
            }
        } else {
            for (int i = threadIdx.x; i < PW_KC * PW_TP; i += 256) {
                const int kk = i / PW_TP, c = i - kk * PW_TP;
                const int k = k0 + kk;
                const long p = p0 + c;
                xs[i] = (k < K && p < P) ? X[(size_t)k * P + p] : 0.f;
            }
        }
        __syncthreads();
        if (vec && k0 + PW_KC < K) prefetch(k0 + PW_KC);  // in flight during the MFMAs
        // MFMA operands: A[i = channel][k] -> lane (pl, kh) holds A(pl, kh); B[k][j = position] -> X(kh, pl)
#pragma unroll 4
        for (int s = 0; s < PW_KC; s += 2) {
            const float bv = xs[(s + kh) * PW_TP + wave * 32 + pl];
#pragma unroll
            for (int c = 0; c < NCT; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[(c * 32 + pl) * PW_LDA + s + kh], bv, acc[c], 0, 0, 0);
        }
    }
    // accumulator layout: column = lane & 31 (position), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const long p = p0 + wave * 32 + pl;
    if (p < P) {
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + c * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < M) Y[(size_t)m * P + p] = bias ? acc[c][r] + bias[m] : acc[c][r];
            }
    }
}

// partial[g][m][k] = sum over workgroup g's position tiles of dY[b][m][p] X[b][k][p]
template <int NCT, int NIT>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(int M, int K, long P, int tiles_per_cloud, int ntiles,
                                                       int tiles_per_wg, const float *__restrict__ dy,
                                                       const float *__restrict__ x, float *__restrict__ partial, int vec)
{
    extern __shared__ float smem[];
    float *dys = smem;                        // [NCT*32][PWW_LD]
    float *xs = smem + NCT * 32 * PWW_LD;     // [NIT*32][PWW_LD]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pl = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * NCT * 32, k0 = blockIdx.z * NIT * 32;
    pw_f32x16 acc[NCT][NIT];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int d = 0; d < NIT; ++d) acc[c][d] = pw_f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    const int t_end = min(ntiles, (int)(blockIdx.x + 1) * tiles_per_wg);
    // vec path: the next tile's rows are loaded into registers before the current tile is multiplied (in flight during the
    // MFMAs) and stored to LDS after the barrier that retires it
    constexpr int LPT = (NCT + NIT) * 32 * (PWW_TP / 4) / 256;  // float4 per thread and tile
    float4 pre[LPT];
    auto prefetch = [&](int t) {
        const int b = t / tiles_per_cloud;
        const long p0 = (long)(t - b * tiles_per_cloud) * PWW_TP;
        const float *DY = dy + (size_t)b * M * P, *X = x + (size_t)b * K * P;
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
            const int i = threadIdx.x + u * 256;
            const int row = i / (PWW_TP / 4), c4 = i - row * (PWW_TP / 4);
            const long p = p0 + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < NCT * 32) {
                const int m = m0 + row;
                if (m < M && p < P) v = *(const float4 *)(DY + (size_t)m * P + p);
            } else {
                const int k = k0 + row - NCT * 32;
                if (k < K && p < P) v = *(const float4 *)(X + (size_t)k * P + p);
            }
            pre[u] = v;
        }
    };
    if (vec && (int)(blockIdx.x * tiles_per_wg) < t_end) prefetch(blockIdx.x * tiles_per_wg);
    for (int t = blockIdx.x * tiles_per_wg; t < t_end; ++t) {
        const int b = t / tiles_per_cloud;
        const long p0 = (long)(t - b * tiles_per_cloud) * PWW_TP;
        const float *DY = dy + (size_t)b * M * P, *X = x + (size_t)b * K * P;
        __syncthreads();  // previous tile consumed
        if (vec) {
            // rows 0..NCT*32-1 are dY channels, the rest X channels; 16 float4 per row
#pragma unroll
            for (int u = 0; u < LPT; ++u) {
                const int i = threadIdx.x + u * 256;
                const int row = i / (PWW_TP / 4), c4 = i - row * (PWW_TP / 4);
                float *d = smem + row * PWW_LD + c4 * 4;
                d[0] = pre[u].x; d[1] = pre[u].y; d[2] = pre[u].z; d[3] = pre[u].w;
            }
        } else {
            for (int i = threadIdx.x; i < (NCT + NIT) * 32 * PWW_TP; i += 256) {
                const int row = i / PWW_TP, c = i - row * PWW_TP;
                const long p = p0 + c;
                float v = 0.f;
                if (row < NCT * 32) {
                    const int m = m0 + row;
                    if (m < M && p < P) v = DY[(size_t)m * P + p];
                } else {
                    const int k = k0 + row - NCT * 32;
                    if (k < K && p < P) v = X[(size_t)k * P + p];
                }
                smem[row * PWW_LD + c] = v;
            }
        }
        __syncthreads();
        if (vec && t + 1 < t_end) prefetch(t + 1);
        // A[i = co][k = position] = dY[co][position],  B[k = position][j = ci] = X[ci][position]
#pragma unroll
        for (int s = 0; s < PWW_TP / 4; s += 2) {
            const int pp = wave * (PWW_TP / 4) + s + kh;
            float av[NCT], bv[NIT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) av[c] = dys[(c * 32 + pl) * PWW_LD + pp];
#pragma unroll
            for (int d = 0; d < NIT; ++d) bv[d] = xs[(d * 32 + pl) * PWW_LD + pp];
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int d = 0; d < NIT; ++d) acc[c][d] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[d], acc[c][d], 0, 0, 0);
        }
    }
    // sum the four waves' accumulators through LDS (fixed order), write this workgroup's block of its partial
    __syncthreads();
    float *red = smem;  // [NCT*32][NIT*32]
    for (int i = threadIdx.x; i < NCT * 32 * NIT * 32; i += 256) red[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int d = 0; d < NIT; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        red[row * (NIT * 32) + d * 32 + pl] += acc[c][d][r];
                    }
        }
        __syncthreads();
    }
    float *out = partial + (size_t)blockIdx.x * M * K;
    for (int i = threadIdx.x; i < NCT * 32 * NIT * 32; i += 256) {
        const int row = i / (NIT * 32), col = i - row * (NIT * 32);
        const int m = m0 + row, k = k0 + col;
        if (m < M && k < K) out[(size_t)m * K + k] = red[i];
    }
}

// db[c] = sum_{b,p} dy[b][c][p]: one 1024-thread workgroup per channel, 16-byte loads, fixed summation order (the bias
// gradient of the stem and head convs is a (8,32,24000) / (8,13,24000) reduction: 56 + 23 us as library reductions)
__global__ __launch_bounds__(1024) void pw_bias_grad_kernel(int nb, int C, long P, const float *__restrict__ dy,
                                                            float *__restrict__ db, int vec)
{
    __shared__ float red[16];
    const int c = blockIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int b = 0; b < nb; ++b) {
        const float *row = dy + ((size_t)b * C + c) * P;
        if (vec) {
            const long n4 = P / 4;
            long i = threadIdx.x;
            for (; i + 1024 < n4; i += 2048) {  // two loads in flight
                const float4 u = *(const float4 *)(row + i * 4), v = *(const float4 *)(row + (i + 1024) * 4);
                s0 += u.x + v.x; s1 += u.y + v.y; s2 += u.z + v.z; s3 += u.w + v.w;
            }
            for (; i < n4; i += 1024) {
                const float4 u = *(const float4 *)(row + i * 4);
                s0 += u.x; s1 += u.y; s2 += u.z; s3 += u.w;
            }
        } else {
            for (long i = threadIdx.x; i < P; i += 1024) s0 += row[i];
        }
    }
    float s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k];
        db[c] = t;
    }
}

struct PwSplit {
    int tiles_per_cloud, ntiles, tiles_per_wg, groups, mchunks, kchunks, nct, nit;
};

static PwSplit pw_split(int b, int cin, int cout, long P)
{
    PwSplit s;
    s.nct = cout <= 32 ? 1 : cout <= 64 ? 2 : cout <= 96 ? 3 : 4;
    s.nit = cin <= 32 ? 1 : 2;
    s.mchunks = div_up(cout, s.nct * 32);
    s.kchunks = div_up(cin, s.nit * 32);
    s.tiles_per_cloud = div_up(P, PWW_TP);
    s.ntiles = s.tiles_per_cloud * b;
    const int max_groups = max(1, 2048 / (s.mchunks * s.kchunks));
    // eight tiles per workgroup on long layers; short ones (a few dozen tiles: the coarse stages) two, or they run on a handful
    // of workgroups (6 workgroups took 37-94 us for the masking module's coarse convs)
    int groups = min(max(1, s.ntiles >= 512 ? s.ntiles / 8 : (s.ntiles + 1) / 2), max_groups);
    s.tiles_per_wg = div_up(s.ntiles, groups);
    s.groups = div_up(s.ntiles, s.tiles_per_wg);
    return s;
}

template <int NCT>
static void launch_pw_gemm(int b, int M, int K, long P, const float *a, long sam, long sak, const float *bias,
                           const float *x, float *y, hipStream_t stream)
{
    const int vec = (P % 4 == 0) && (((uintptr_t)x & 15) == 0);
    hipLaunchKernelGGL((pw_gemm_kernel<NCT>), dim3(div_up(P, PW_TP), div_up(M, NCT * 32), b), dim3(256), 0, stream, M, K, P,
                       a, sam, sak, bias, x, y, vec);
}

static void pw_gemm(int b, int M, int K, long P, const float *a, long sam, long sak, const float *bias, const float *x,
                    float *y, hipStream_t stream)
{
    if (M <= 32) launch_pw_gemm<1>(b, M, K, P, a, sam, sak, bias, x, y, stream);
    else if (M <= 64) launch_pw_gemm<2>(b, M, K, P, a, sam, sak, bias, x, y, stream);
    else if (M <= 96) launch_pw_gemm<3>(b, M, K, P, a, sam, sak, bias, x, y, stream);
    else launch_pw_gemm<4>(b, M, K, P, a, sam, sak, bias, x, y, stream);
}

template <int NCT, int NIT>
static void launch_pw_wgrad(const PwSplit &s, int M, int K, long P, const float *dy, const float *x, float *partial,
                            hipStream_t stream)
{
    const size_t lds = (size_t)(NCT + NIT) * 32 * PWW_LD * sizeof(float);
    (void)hipFuncSetAttribute((const void *)pw_wgrad_kernel<NCT, NIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int vec = (P % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)dy & 15) == 0);
    hipLaunchKernelGGL((pw_wgrad_kernel<NCT, NIT>), dim3(s.groups, s.mchunks, s.kchunks), dim3(256), lds, stream, M, K, P,
                       s.tiles_per_cloud, s.ntiles, s.tiles_per_wg, dy, x, partial, vec);
}

// gemm.hip: the MFMA-bound variant for layers with >= 64 channels on both sides
bool gemm_conv_pays(int cin, int cout);
int gemm_conv_forward(int b, int cin, int cout, long P, const float *x, const float *w, const float *bias, float *y, float *partial,
                      hipStream_t stream);
bool gemm_conv_short(int b, int rows, long P);
// the 128 x 128-tile GEMM (gemm.hip) instead of the streaming kernel above: deep layers without a bias, and -- with or without
// one -- deep layers so short that the streaming kernel's one-workgroup-per-tile walk of the whole K axis leaves the chip empty
static bool pw_forward_deep(int b, int cin, int cout, long P, bool has_bias)
{
    if (!gemm_conv_pays(cin, cout) || P >= (1L << 31)) return false;
    return !has_bias || gemm_conv_short(b, cout, P);
}
size_t gemm_conv_forward_workspace_bytes(int b, int cin, int cout, long P);
size_t gemm_conv_backward_data_workspace_bytes(int b, int cin, int cout, long P);
int gemm_conv_backward_data(int b, int cin, int cout, long P, const float *dy, const float *w, float *dx, float *partial, hipStream_t stream);
size_t gemm_conv_wgrad_workspace_bytes(int b, int cin, int cout, long P);
int gemm_conv_backward_weight(int b, int cin, int cout, long P, const float *x, const float *dy, float *dw, float *partial,
                              hipStream_t stream);

}  // namespace amc

using namespace amc;

// y (b,cout,P) = weight (cout,cin) . x (b,cin,P) (+ bias)
AMC_API int amc3d_pointwise_conv_forward(int b, int cin, int cout, long P, const float *x, const float *weight,
                                         const float *bias, float *y, void *stream)
{
    if (b <= 0 || P <= 0 || cout <= 0) return 0;
    if (cin <= 0 || !x || !weight || !y) return bad_arg("amc3d_pointwise_conv_forward: bad argument");
    if (pw_forward_deep(b, cin, cout, P, bias != nullptr))
        return gemm_conv_forward(b, cin, cout, P, x, weight, bias, y, nullptr, (hipStream_t)stream);
    pw_gemm(b, cout, cin, P, weight, cin, 1, bias, x, y, (hipStream_t)stream);
    return launch_status("amc3d_pointwise_conv_forward");
}

// the same with scratch for the short deep layers (a few hundred positions per cloud), whose K axis is then split over
// workgroups: amc3d_pointwise_conv_forward_workspace_bytes() bytes, 0 for every other shape
AMC_API size_t amc3d_pointwise_conv_forward_workspace_bytes(int b, int cin, int cout, long P, int has_bias)
{
    if (b <= 0 || P <= 0 || cin <= 0 || cout <= 0 || !pw_forward_deep(b, cin, cout, P, has_bias != 0)) return 0;
    return gemm_conv_forward_workspace_bytes(b, cin, cout, P);
}

AMC_API int amc3d_pointwise_conv_forward_ws(int b, int cin, int cout, long P, const float *x, const float *weight,
                                            const float *bias, float *y, void *workspace, size_t workspace_bytes, void *stream)
{
    if (b <= 0 || P <= 0 || cout <= 0) return 0;
    if (cin <= 0 || !x || !weight || !y) return bad_arg("amc3d_pointwise_conv_forward_ws: bad argument");
    const size_t need = amc3d_pointwise_conv_forward_workspace_bytes(b, cin, cout, P, bias != nullptr);
    if (need && (!workspace || workspace_bytes < need)) return bad_arg("amc3d_pointwise_conv_forward_ws: workspace too small");
    if (pw_forward_deep(b, cin, cout, P, bias != nullptr))
        return gemm_conv_forward(b, cin, cout, P, x, weight, bias, y, need ? (float *)workspace : nullptr, (hipStream_t)stream);
    pw_gemm(b, cout, cin, P, weight, cin, 1, bias, x, y, (hipStream_t)stream);
    return launch_status("amc3d_pointwise_conv_forward_ws");
}

AMC_API size_t amc3d_pointwise_conv_workspace_bytes(int b, int cin, int cout, long P)
{
    if (b <= 0 || P <= 0 || cin <= 0 || cout <= 0) return 0;
    if (gemm_conv_pays(cin, cout) && P < (1L << 31)) {  // weight-gradient partials, then (same stream) the split-K partials of dx
        const size_t wg = gemm_conv_wgrad_workspace_bytes(b, cin, cout, P);
        const size_t bd = (cout > 128 || gemm_conv_short(b, cin, P)) ? gemm_conv_backward_data_workspace_bytes(b, cin, cout, P) : 0;
        return wg > bd ? wg : bd;
    }
    const PwSplit s = pw_split(b, cin, cout, P);
    return (size_t)s.groups * cout * cin * sizeof(float);
}

// dx (b,cin,P) = weight^T . dy   (when dx != NULL);  dweight (cout,cin) = sum_{b,p} dy x^T   (when dweight != NULL)
AMC_API int amc3d_pointwise_conv_backward(int b, int cin, int cout, long P, const float *x, const float *weight,
                                          const float *dy, float *dx, float *dweight, void *workspace,
                                          size_t workspace_bytes, void *stream_)
{
    if (b <= 0 || P <= 0) return 0;
    if (cin <= 0 || cout <= 0 || !dy || (dx && !weight)) return bad_arg("amc3d_pointwise_conv_backward: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const bool deep = gemm_conv_pays(cin, cout) && P < (1L << 31);
    if (dx) {
        if (deep && (cout > 128 || gemm_conv_short(b, cin, P))) {
            const size_t need = gemm_conv_backward_data_workspace_bytes(b, cin, cout, P);
            float *part = (need && workspace && workspace_bytes >= need) ? (float *)workspace : nullptr;
            if (int st = gemm_conv_backward_data(b, cin, cout, P, dy, weight, dx, part, stream)) return st;
        }
        else pw_gemm(b, cin, cout, P, weight, 1, cin, nullptr, dy, dx, stream);
    }
    if (dweight) {
        if (!x || !workspace || workspace_bytes < amc3d_pointwise_conv_workspace_bytes(b, cin, cout, P))
            return bad_arg("amc3d_pointwise_conv_backward: null pointer or workspace too small");
        if (deep) return gemm_conv_backward_weight(b, cin, cout, P, x, dy, dweight, (float *)workspace, stream);
        const PwSplit s = pw_split(b, cin, cout, P);
        float *partial = (float *)workspace;
#define AMC_PWW(A, B) launch_pw_wgrad<A, B>(s, cout, cin, P, dy, x, partial, stream)
        if (s.nit == 1) {
            if (s.nct == 1) AMC_PWW(1, 1); else if (s.nct == 2) AMC_PWW(2, 1); else if (s.nct == 3) AMC_PWW(3, 1); else AMC_PWW(4, 1);
        } else {
            if (s.nct == 1) AMC_PWW(1, 2); else if (s.nct == 2) AMC_PWW(2, 2); else if (s.nct == 3) AMC_PWW(3, 2); else AMC_PWW(4, 2);
        }
#undef AMC_PWW
        const int st = reduce_partials(cout * cin, s.groups, partial, dweight, stream);
        if (st) return st;
    }
    return launch_status("amc3d_pointwise_conv_backward");
}

// dbias (c) = sum_{b,p} dy (b,c,P): the bias gradient of a 1x1 convolution (fixed order -> deterministic)
AMC_API int amc3d_bias_grad(int b, int c, long P, const float *dy, float *dbias, void *stream)
{
    if (c <= 0) return 0;
    if (b < 0 || P < 0 || !dbias || (!dy && b > 0 && P > 0)) return bad_arg("amc3d_bias_grad: bad argument");
    const int vec = (P % 4 == 0) && (((uintptr_t)dy & 15) == 0);
    hipLaunchKernelGGL(pw_bias_grad_kernel, dim3(c), dim3(1024), 0, (hipStream_t)stream, b, c, P, dy, dbias, vec);
    return launch_status("amc3d_bias_grad");
}

// ---- a 1x1-conv weight cut into two column blocks, and the two gradient blocks joined again, one launch each ------------
// (the neighbourhood layers' W = [W_dp | W_f], pointnext_AA.py:57-63 after get_aggregation_feautres' concatenation, and the
// FeaturePropagation conv's [W_skip | W_up], :210-226: as tensor slices that is two copies forward and a concatenation
// backward per layer, 36 + 18 launches per PointNeXt-L step)
namespace amc {
__global__ void split_columns_kernel(long total, int c1, int c2, const float *__restrict__ w, float *__restrict__ a,
                                     float *__restrict__ b)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long r = i / (c1 + c2);
    const int c = (int)(i - r * (c1 + c2));
    if (c < c1) a[r * c1 + c] = w[i]; else b[r * c2 + (c - c1)] = w[i];
}

__global__ void join_columns_kernel(long total, int c1, int c2, const float *__restrict__ a, const float *__restrict__ b,
                                    float *__restrict__ w)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long r = i / (c1 + c2);
    const int c = (int)(i - r * (c1 + c2));
    w[i] = c < c1 ? a[r * c1 + c] : b[r * c2 + (c - c1)];
}
}  // namespace amc

// w (rows, c1 + c2) -> a (rows, c1), b (rows, c2)
AMC_API int amc3d_split_columns(int rows, int c1, int c2, const float *w, float *a, float *b, void *stream)
{
    if (rows <= 0 || c1 < 0 || c2 < 0 || c1 + c2 == 0) return 0;
    if (!w || (c1 && !a) || (c2 && !b)) return bad_arg("amc3d_split_columns: null pointer");
    const long total = (long)rows * (c1 + c2);
    hipLaunchKernelGGL(split_columns_kernel, dim3(div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, total, c1, c2, w, a, b);
    return launch_status("amc3d_split_columns");
}

// a (rows, c1), b (rows, c2) -> w (rows, c1 + c2)
AMC_API int amc3d_join_columns(int rows, int c1, int c2, const float *a, const float *b, float *w, void *stream)
{
    if (rows <= 0 || c1 < 0 || c2 < 0 || c1 + c2 == 0) return 0;
    if (!w || (c1 && !a) || (c2 && !b)) return bad_arg("amc3d_join_columns: null pointer");
    const long total = (long)rows * (c1 + c2);
    hipLaunchKernelGGL(join_columns_kernel, dim3(div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, total, c1, c2, a, b, w);
    return launch_status("amc3d_join_columns");
}
