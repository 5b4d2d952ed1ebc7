// Grouped 1x1 convolution fused with its gather: the first layer of every SetAbstraction /
// LocalAggregation MLP, on the fp32 matrix cores of gfx950.
//
// Reference (pointnext_AA.py:164-166, group.py:244-255,323-325): ball_query -> grouping_operation
// materialises fj (B,C,M,32), torch.cat([dp, fj]) copies it into X (B,C+3,M,32), nn.Conv2d 1x1 reads X
// and writes Y (B,C',M,32); the backward writes dX, slices it and scatters it with atomics.  X is the
// largest tensor of the model (215 MB at the first stage) and exists only to be multiplied by a
// (C+3) x C' matrix.
//
// Here Y = W . [dp ; f[idx]] is computed tile by tile without ever storing X:
//   * features are kept point-major (B,N,C) for this layer, so a neighbour is ONE contiguous row;
//   * a workgroup stages 128 positions: 128 gathered rows + their 3 relative coordinates into an LDS
//     tile [position][channel] (row stride odd -> conflict-free column reads), W next to it;
//   * each of the 4 waves multiplies its 32 positions with v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered
//     fma chain): output channels on the rows, positions on the columns, so the accumulator registers are
//     written as 128-byte rows of the channel-major Y the BatchNorm kernels read;
//   * backward-data is the transposed product dY^T . W with the channels on the columns, so each
//     accumulator register is a 128-byte contiguous piece of one neighbour row and goes out as a
//     full-rate float atomic into the point-major gradient image (no dX tensor);
//   * backward-weight contracts over positions (dY . X^T), accumulates 1024 positions per workgroup in
//     registers and writes one partial per workgroup; a second kernel sums the partials in a fixed order.
// Shapes covered: C <= 64 input channels, C' in {32, 64, 96, 128}; other layers use the generic path.
#include "common.h"

namespace amc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GCC_TP = 128;        // positions per tile (4 waves x 32)
constexpr int GCC_TILES_WRW = 8;   // tiles per workgroup in the weight-gradient kernel
constexpr int GCC_MAX_CIN = 64;
constexpr int GCC_MAX_COUT = 128;

__host__ __device__ inline int gcc_kp(int cin) { return (cin + 3 + 1) & ~1; }   // K padded to even
__host__ __device__ inline int gcc_xs(int cin) { return gcc_kp(cin) + 1; }      // odd LDS row stride

// (B,C,N) -> (B,N,C)
__global__ __launch_bounds__(256) void transpose_cn_kernel(int c, int n, const float *__restrict__ src,
                                                           float *__restrict__ dst)
{
    __shared__ float tile[64][65];
    const int bs = blockIdx.z;
    const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < 64; r += 4) {  // r: channel, lane: point
        const int ch = c0 + r, nn = n0 + lane;
        tile[r][lane] = (ch < c && nn < n) ? src[((size_t)bs * c + ch) * n + nn] : 0.f;
    }
    __syncthreads();
    for (int r = wave; r < 64; r += 4) {  // r: point, lane: channel
        const int nn = n0 + r, ch = c0 + lane;
        if (nn < n && ch < c) dst[((size_t)bs * n + nn) * c + ch] = tile[lane][r];
    }
}

// stage one tile of X = [dp ; f[idx]] as xs[position][channel] (stride XS), zero padded
__device__ __forceinline__ void gcc_stage_x(int cin, int n, long P, long p0, int bs, const float *__restrict__ f_pm,
                                            const float *__restrict__ dp, const int *__restrict__ idx, float *xs,
                                            int *sidx)
{
    const int XS = gcc_xs(cin), KP = gcc_kp(cin);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < GCC_TP; i += 256) sidx[i] = (p0 + i < P) ? idx[(size_t)bs * P + p0 + i] : -1;
    for (int i = threadIdx.x; i < 3 * GCC_TP; i += 256) {
        const int c = i / GCC_TP, p = i - c * GCC_TP;
        xs[p * XS + c] = (p0 + p < P) ? dp[((size_t)bs * 3 + c) * P + p0 + p] : 0.f;
    }
    for (int i = threadIdx.x; i < GCC_TP * (KP - cin - 3); i += 256) {  // K padding column(s)
        const int p = i / (KP - cin - 3), c = cin + 3 + i % (KP - cin - 3);
        xs[p * XS + c] = 0.f;
    }
    __syncthreads();  // sidx
    if (cin <= 32) {  // two rows per wave-instruction
        const int half = lane >> 5, c = lane & 31;
        for (int r = wave * 2 + half; r < GCC_TP; r += 8) {
            const int id = sidx[r];
            if (c < cin) xs[r * XS + 3 + c] = id >= 0 ? f_pm[((size_t)bs * n + id) * cin + c] : 0.f;
        }
    } else {
        for (int r = wave; r < GCC_TP; r += 4) {
            const int id = sidx[r];
            for (int c = lane; c < cin; c += 64) xs[r * XS + 3 + c] = id >= 0 ? f_pm[((size_t)bs * n + id) * cin + c] : 0.f;
        }
    }
}

__device__ __forceinline__ void gcc_stage_w(int cin, int cout, const float *__restrict__ W, float *ws)
{
    const int XS = gcc_xs(cin), KP = gcc_kp(cin), CP = cin + 3;
    for (int i = threadIdx.x; i < cout * KP; i += 256) {
        const int co = i / KP, ci = i - co * KP;
        ws[co * XS + ci] = ci < CP ? W[(size_t)co * CP + ci] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// forward: Y (B,Cout,P) = W (Cout, Cin+3) . X
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gcc_fwd_kernel(int cin, int cout, int n, long P, const float *__restrict__ f_pm,
                                                      const float *__restrict__ dp, const int *__restrict__ idx,
                                                      const float *__restrict__ W, float *__restrict__ Y)
{
    extern __shared__ float smem[];
    const int XS = gcc_xs(cin), KP = gcc_kp(cin);
    float *xs = smem;                       // [GCC_TP][XS]
    float *ws = xs + GCC_TP * XS;           // [cout][XS]
    int *sidx = (int *)(ws + cout * XS);    // [GCC_TP]
    const int bs = blockIdx.y;
    const long p0 = (long)blockIdx.x * GCC_TP;
    gcc_stage_w(cin, cout, W, ws);
    gcc_stage_x(cin, n, P, p0, bs, f_pm, dp, idx, xs, sidx);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pl = lane & 31, kh = lane >> 5;
    const float *xrow = xs + (wave * 32 + pl) * XS + kh;
    const long pos = p0 + wave * 32 + pl;
    for (int ct = 0; ct < cout / 32; ++ct) {
        const float *wrow = ws + (ct * 32 + pl) * XS + kh;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < KP; s += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[s], xrow[s], acc, 0, 0, 0);
        if (pos < P) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                Y[((size_t)bs * cout + co) * P + pos] = acc[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward-data: dF_pm (B,N,Cin) += scatter( dY^T . W[:, 3:] )   (no gradient for the 3 dp channels)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gcc_bwd_data_kernel(int cin, int cout, int n, long P,
                                                           const float *__restrict__ dY, const int *__restrict__ idx,
                                                           const float *__restrict__ W, float *__restrict__ dF_pm)
{
    extern __shared__ float smem[];
    const int XS = gcc_xs(cin);
    constexpr int DS = GCC_TP + 1;
    float *dys = smem;                    // [cout][DS]
    float *ws = dys + cout * DS;          // [cout][XS]
    int *sidx = (int *)(ws + cout * XS);  // [GCC_TP]
    const int bs = blockIdx.y;
    const long p0 = (long)blockIdx.x * GCC_TP;
    gcc_stage_w(cin, cout, W, ws);
    for (int i = threadIdx.x; i < GCC_TP; i += 256) sidx[i] = (p0 + i < P) ? idx[(size_t)bs * P + p0 + i] : -1;
    for (int i = threadIdx.x; i < cout * GCC_TP; i += 256) {
        const int co = i / GCC_TP, p = i - co * GCC_TP;
        dys[co * DS + p] = (p0 + p < P) ? dY[((size_t)bs * cout + co) * P + p0 + p] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pl = lane & 31, kh = lane >> 5;
    for (int cit = 0; cit < (cin + 31) / 32; ++cit) {
        const int ci = cit * 32 + pl;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // A[i = position][k = co] = dY[co][position],  B[k = co][j = ci] = W[co][3 + ci]
        for (int s = 0; s < cout; s += 2) {
            const float a = dys[(s + kh) * DS + wave * 32 + pl];
            const float b = ci < cin ? ws[(s + kh) * XS + 3 + ci] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (ci < cin) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int prow = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int id = sidx[prow];
                if (id >= 0) atomicAdd(dF_pm + ((size_t)bs * n + id) * cin + ci, acc[r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward-weight: partial[wg] (Cout, Cin+3) = sum over the workgroup's positions of dY . X^T
// ---------------------------------------------------------------------------------------------
template <int NCT, int NIT>
__global__ __launch_bounds__(256) void gcc_bwd_weight_kernel(int cin, int cout, int n, long P,
                                                             const float *__restrict__ f_pm,
                                                             const float *__restrict__ dp, const int *__restrict__ idx,
                                                             const float *__restrict__ dY, float *__restrict__ partial)
{
    extern __shared__ float smem[];
    const int XS = gcc_xs(cin), CP = cin + 3;
    constexpr int DS = GCC_TP + 1;
    float *xs = smem;                       // [GCC_TP][XS]
    float *dys = xs + GCC_TP * XS;          // [cout][DS]
    int *sidx = (int *)(dys + cout * DS);   // [GCC_TP]
    const int bs = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pl = lane & 31, kh = lane >> 5;
    f32x16 acc[NCT][NIT];
#pragma unroll
    for (int a = 0; a < NCT; ++a)
#pragma unroll
        for (int b = 0; b < NIT; ++b) acc[a][b] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float dpacc[NCT][3];  // relative-position channels: lane (pl, kh) accumulates dW[co = a*32+pl][c] over its positions
#pragma unroll
    for (int a = 0; a < NCT; ++a) dpacc[a][0] = dpacc[a][1] = dpacc[a][2] = 0.f;

    for (int t = 0; t < GCC_TILES_WRW; ++t) {
        const long p0 = ((long)blockIdx.x * GCC_TILES_WRW + t) * GCC_TP;
        if (p0 >= P) break;  // workgroup-uniform
        __syncthreads();     // previous tile consumed
        gcc_stage_x(cin, n, P, p0, bs, f_pm, dp, idx, xs, sidx);
        for (int i = threadIdx.x; i < cout * GCC_TP; i += 256) {
            const int co = i / GCC_TP, p = i - co * GCC_TP;
            dys[co * DS + p] = (p0 + p < P) ? dY[((size_t)bs * cout + co) * P + p0 + p] : 0.f;
        }
        __syncthreads();
        // A[i = co][k = position] = dY[co][position],  B[k = position][j = ci] = X[position][3 + ci]
        for (int s = 0; s < 32; s += 2) {
            const int prow = wave * 32 + s + kh;
            const float d0 = xs[prow * XS + 0], d1 = xs[prow * XS + 1], d2 = xs[prow * XS + 2];  // broadcast reads
            float av[NCT];
#pragma unroll
            for (int a = 0; a < NCT; ++a) {
                av[a] = dys[(a * 32 + pl) * DS + prow];
                // the same dY value feeds the three dp channels as plain FMAs (co-issued under the MFMAs)
                dpacc[a][0] += av[a] * d0;
                dpacc[a][1] += av[a] * d1;
                dpacc[a][2] += av[a] * d2;
            }
#pragma unroll
            for (int b = 0; b < NIT; ++b) {
                const int ci = b * 32 + pl;
                const float xb = ci < cin ? xs[prow * XS + 3 + ci] : 0.f;
#pragma unroll
                for (int a = 0; a < NCT; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], xb, acc[a][b], 0, 0, 0);
            }
        }
    }
    // sum the four waves' accumulators through LDS, write this workgroup's partial
    __syncthreads();
    float *red = smem;  // [cout][cin] floats, reuses the tile memory
    for (int i = threadIdx.x; i < cout * cin; i += 256) red[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < NCT; ++a)
#pragma unroll
                for (int b = 0; b < NIT; ++b) {
                    const int ci = b * 32 + pl;
                    if (ci < cin) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int co = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                            red[co * cin + ci] += acc[a][b][r];
                        }
                    }
                }
        }
        __syncthreads();
    }
    float *out = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * cout * CP;
    for (int i = threadIdx.x; i < cout * cin; i += 256) {
        const int co = i / cin, ci = i - co * cin;
        out[co * CP + 3 + ci] = red[i];
    }
    // dp channels: sum the two position halves (lanes l, l+32) and the four waves
    __syncthreads();
    float *redp = smem;  // [cout][3]
    for (int i = threadIdx.x; i < cout * 3; i += 256) redp[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NCT; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) dpacc[a][c] += __shfl_xor(dpacc[a][c], 32, 64);
    for (int w = 0; w < 4; ++w) {
        if (wave == w && kh == 0) {
#pragma unroll
            for (int a = 0; a < NCT; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) redp[(a * 32 + pl) * 3 + c] += dpacc[a][c];
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < cout * 3; i += 256) out[(i / 3) * CP + i % 3] = redp[i];
}

// dW (Cout, Cin+3) = sum of the partials in a fixed order (deterministic): a workgroup owns 64
// consecutive elements, its 16 slices each sum every 16th partial, then the slices are added in order.
__global__ __launch_bounds__(1024) void gcc_reduce_partials_kernel(int total, int nparts, const float *__restrict__ partial,
                                                                   float *__restrict__ dW)
{
    __shared__ float red[16][64];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e;
    float s = 0.f;
    if (i < total) {
        // four loads in flight per thread (a fixed order all the same: four interleaved sub-sums, added in order)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = sl;
        for (; k + 48 < nparts; k += 64) {
            s0 += partial[(size_t)k * total + i];
            s1 += partial[(size_t)(k + 16) * total + i];
            s2 += partial[(size_t)(k + 32) * total + i];
            s3 += partial[(size_t)(k + 48) * total + i];
        }
        for (; k < nparts; k += 16) s0 += partial[(size_t)k * total + i];
        s = (s0 + s1) + (s2 + s3);
    }
    red[sl][e] = s;
    __syncthreads();
    if (sl == 0 && i < total) {
        float t = red[0][e];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][e];
        dW[i] = t;
    }
}

// <= 16 partials (the wide, short layers of PointNeXt-XL: a 4096 x 1024 weight over 124 positions has two): a thread sums
// four outputs' partials in order -- the value the kernel above produces for nparts <= 16 (one partial per slice, slices added
// in order), without 1024 threads per 64 outputs (96 -> 15 us for 4 M outputs)
__global__ __launch_bounds__(256) void gcc_reduce_few_kernel(int total4, int nparts, const float4 *__restrict__ partial,
                                                             float4 *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 t = partial[i];
    for (int k = 1; k < nparts; ++k) {
        const float4 v = partial[(size_t)k * total4 + i];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    out[i] = t;
}

int reduce_partials(int total, int nparts, const float *partial, float *out, hipStream_t stream)
{
    if (nparts >= 1 && nparts <= 16 && total % 4 == 0 && total >= 65536 && (((uintptr_t)partial | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(gcc_reduce_few_kernel, dim3(div_up(total / 4, 256)), dim3(256), 0, stream, total / 4, nparts,
                           (const float4 *)partial, (float4 *)out);
        return launch_status("reduce_partials");
    }
    hipLaunchKernelGGL(gcc_reduce_partials_kernel, dim3(div_up(total, 64)), dim3(1024), 0, stream, total, nparts, partial, out);
    return launch_status("reduce_partials");
}

// opt in to large dynamic LDS allocations (gfx950: up to 160 KiB per workgroup)
template <typename K>
static void allow_lds(K kernel, size_t bytes)
{
    (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static bool gcc_supported(int cin, int cout) { return cin >= 1 && cin <= GCC_MAX_CIN && cout >= 32 && cout <= GCC_MAX_COUT && cout % 32 == 0; }

}  // namespace amc

using namespace amc;

AMC_API int amc3d_grouped_conv_supported(int cin, int cout) { return gcc_supported(cin, cout) ? 1 : 0; }

// (B,C,N) -> (B,N,C): the point-major copy of the features the fused layer gathers from
AMC_API int amc3d_transpose_cn(int b, int c, int n, const float *src, float *dst, void *stream)
{
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (!src || !dst) return bad_arg("amc3d_transpose_cn: null pointer");
    hipLaunchKernelGGL(transpose_cn_kernel, dim3(div_up(n, 64), div_up(c, 64), b), dim3(256), 0, (hipStream_t)stream, c, n,
                       src, dst);
    return launch_status("amc3d_transpose_cn");
}

// Y (b,cout,npoints,nsample) = W (cout, cin+3) . [dp (b,3,npoints,nsample) ; f_pm[idx] ], f_pm (b,n,cin) point-major
AMC_API int amc3d_grouped_conv_forward(int b, int cin, int cout, int n, int npoints, int nsample, const float *f_pm,
                                       const float *dp, const int *idx, const float *weight, float *y, void *stream)
{
    const long P = (long)npoints * nsample;
    if (b <= 0 || P <= 0) return 0;
    if (!gcc_supported(cin, cout) || !f_pm || !dp || !idx || !weight || !y)
        return bad_arg("amc3d_grouped_conv_forward: unsupported shape or null pointer");
    const size_t lds = ((size_t)GCC_TP * gcc_xs(cin) + (size_t)cout * gcc_xs(cin) + GCC_TP) * 4;
    allow_lds(gcc_fwd_kernel, lds);
    hipLaunchKernelGGL(gcc_fwd_kernel, dim3(div_up(P, GCC_TP), b), dim3(256), lds, (hipStream_t)stream, cin, cout, n, P, f_pm,
                       dp, idx, weight, y);
    return launch_status("amc3d_grouped_conv_forward");
}

AMC_API size_t amc3d_grouped_conv_workspace_bytes(int b, int cin, int cout, int npoints, int nsample)
{
    const long P = (long)npoints * nsample;
    const long groups = (P + (long)GCC_TP * GCC_TILES_WRW - 1) / ((long)GCC_TP * GCC_TILES_WRW);
    return (size_t)b * groups * cout * (cin + 3) * sizeof(float);
}

// df_pm (b,n,cin) += gather-transposed W^T dY (caller zero-initialises); dweight (cout, cin+3) = dY . X^T
AMC_API int amc3d_grouped_conv_backward(int b, int cin, int cout, int n, int npoints, int nsample, const float *f_pm,
                                        const float *dp, const int *idx, const float *weight, const float *dy,
                                        float *df_pm, float *dweight, void *workspace, size_t workspace_bytes,
                                        void *stream_)
{
    const long P = (long)npoints * nsample;
    if (b <= 0 || P <= 0) return 0;
    if (!gcc_supported(cin, cout) || !f_pm || !dp || !idx || !weight || !dy || !workspace ||
        workspace_bytes < amc3d_grouped_conv_workspace_bytes(b, cin, cout, npoints, nsample))
        return bad_arg("amc3d_grouped_conv_backward: unsupported shape, null pointer or workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    constexpr int DS = GCC_TP + 1;
    if (df_pm) {
        const size_t lds = ((size_t)cout * DS + (size_t)cout * gcc_xs(cin) + GCC_TP) * 4;
        allow_lds(gcc_bwd_data_kernel, lds);
        hipLaunchKernelGGL(gcc_bwd_data_kernel, dim3(div_up(P, GCC_TP), b), dim3(256), lds, stream, cin, cout, n, P, dy, idx,
                           weight, df_pm);
    }
    if (dweight) {
        const int groups = div_up(P, (long)GCC_TP * GCC_TILES_WRW);
        const size_t lds = ((size_t)GCC_TP * gcc_xs(cin) + (size_t)cout * DS + GCC_TP) * 4;
        float *partial = (float *)workspace;
        const int nct = cout / 32, nit = (cin + 31) / 32;
#define AMC_WRW(A, B)                                                                                                  \
    allow_lds(gcc_bwd_weight_kernel<A, B>, lds);                                                                       \
    hipLaunchKernelGGL((gcc_bwd_weight_kernel<A, B>), dim3(groups, b), dim3(256), lds, stream, cin, cout, n, P, f_pm, dp, \
                       idx, dy, partial)
        if (nit == 1) {
            if (nct == 1) { AMC_WRW(1, 1); } else if (nct == 2) { AMC_WRW(2, 1); } else if (nct == 3) { AMC_WRW(3, 1); } else { AMC_WRW(4, 1); }
        } else {
            if (nct == 1) { AMC_WRW(1, 2); } else if (nct == 2) { AMC_WRW(2, 2); } else if (nct == 3) { AMC_WRW(3, 2); } else { AMC_WRW(4, 2); }
        }
#undef AMC_WRW
        const int total = cout * (cin + 3);
        hipLaunchKernelGGL(gcc_reduce_partials_kernel, dim3(div_up(total, 64)), dim3(1024), 0, stream, total, groups * b,
                           (const float *)partial, dweight);
    }
    return launch_status("amc3d_grouped_conv_backward");
}
