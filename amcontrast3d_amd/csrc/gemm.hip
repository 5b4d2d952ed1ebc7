// fp32 MFMA GEMM for the deep pointwise convolutions (>= 64 channels on both sides) of gfx950.
//
// The three products of a 1x1 convolution on channel-major (B, C, P) tensors
//     forward        Y[b]  = W        . X[b]     M = Cout, N = P,   K = Cin
//     backward-data  dX[b] = W^T      . dY[b]    M = Cin,  N = P,   K = Cout
//     backward-weight dW   = sum_b dY[b] . X[b]^T  M = Cout, N = Cin, K = P (split over workgroups)
// are one kernel:  C[m][n] = sum_k A(m,k) B(k,n)  with strided operand views.  pwconv.hip streams these well when
// the layer is HBM-bound (few channels, many positions); with >= 64 channels the MFMA pipe is the limit and that
// kernel's single-buffered 32-position wave tiles leave it half idle.  Here:
//   * 128 x 128 output tile per workgroup, 2 x 2 waves of 64 x 64 (four 32x32 accumulators: every A and B
//     fragment feeds two v_mfma_f32_32x32x2_f32);
//   * K in chunks of 16, double-buffered in LDS; the next chunk's global loads are issued before the current
//     chunk's MFMAs and written to the other buffer after them: one barrier per chunk;
//   * an operand that is contiguous along k in global memory is kept [row][k] in LDS (row stride 20 floats:
//     8 consecutive rows x 16 bytes cover the 32 banks once) and a lane fetches its 8 k-values of a chunk with
//     two ds_read_b128; an operand contiguous along m / n is kept as it arrives, [k][row], and read with eight
//     conflict-free ds_read_b32 (a transposing store would hit 2 of the 32 banks).
// MFMA operand convention (v_mfma_f32_32x32x2_f32): lane l supplies A[i = l & 31][kk = l >> 5] and
// B[kk = l >> 5][j = l & 31]; step t of a chunk pairs k = t (lanes 0-31) with k = 8 + t (lanes 32-63) in both
// operands -- any pairing is a valid order of the K sum as long as A and B agree.
#include <stdlib.h>

#include "common.h"

namespace amc {

typedef float gm_f32x16 __attribute__((ext_vector_type(16)));

constexpr int GM_T = 128;    // tile edge (M and N)
constexpr int GM_KC = 16;    // K chunk
constexpr int GM_S = 20;     // LDS row stride in floats
constexpr int GM_STAGE = 2 * GM_T * GM_S;  // floats per stage (A then B)

int reduce_partials(int total, int nparts, const float *partial, float *out, hipStream_t stream);  // gcc.hip

struct GemmView {  // element (r, k) of an operand at base[r * sr + k * sk]; exactly one of sr, sk is 1
    const float *base;
    long sr, sk;
};

// registers of one chunk of one operand: 128 rows x 16 k = 512 float4, two per thread
struct GmRegs {
    float4 v[2];
};

// KCONT: the operand is contiguous along k in global memory (natural); otherwise along its row index
template <bool KCONT>
__device__ __forceinline__ void gm_load(GmRegs &r, const GemmView &g, int row0, int nrows, long k0, long kend, bool vec)
{
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KCONT) {
            const int row = t / 4 + 64 * j, c4 = t % 4;
            const long k = k0 + c4 * 4;
            if (row < nrows && k < kend) {
                const float *p = g.base + (long)(row0 + row) * g.sr + k;
                if (vec && k + 3 < kend) v = *reinterpret_cast<const float4 *>(p);
                else { v.x = p[0]; if (k + 1 < kend) v.y = p[1]; if (k + 2 < kend) v.z = p[2]; if (k + 3 < kend) v.w = p[3]; }
            }
        } else {
            const int krow = t / 32 + 8 * j, r4 = t % 32;
            const long k = k0 + krow;
            const int row = r4 * 4;
            if (k < kend && row < nrows) {
                const float *p = g.base + k * g.sk + (row0 + row);
                if (vec && row + 3 < nrows) v = *reinterpret_cast<const float4 *>(p);
                else { v.x = p[0]; if (row + 1 < nrows) v.y = p[1]; if (row + 2 < nrows) v.z = p[2]; if (row + 3 < nrows) v.w = p[3]; }
            }
        }
        r.v[j] = v;
    }
}

template <bool KCONT>
__device__ __forceinline__ void gm_store(const GmRegs &r, float *lds)
{
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (KCONT) {
            const int row = t / 4 + 64 * j, c4 = t % 4;
            *reinterpret_cast<float4 *>(lds + row * GM_S + c4 * 4) = r.v[j];
        } else {
            const int krow = t / 32 + 8 * j, row = (t % 32) * 4;
            *reinterpret_cast<float4 *>(lds + krow * GM_T + row) = r.v[j];
        }
    }
}

// grid: (N tiles, M tiles, batch * splits).  Per z: batch b = z / splits, split s = z % splits owns K range
// [s * kper, min(K, (s+1) * kper)).  Output C = cbase + z * czstride (forward/backward-data: splits = 1 and
// czstride = M * ldc; backward-weight: one partial (M, N) per z).
template <bool A_KCONT, bool B_KCONT>
__global__ __launch_bounds__(256) void gm_gemm_kernel(int M, int N, long K, int splits, long kper, GemmView A, long a_bstride,
                                                      GemmView B, long b_bstride, float *__restrict__ cbase, long czstride,
                                                      long ldc, int vec_a, int vec_b)
{
    extern __shared__ __attribute__((aligned(16))) float gm_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pl = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GM_T, n0 = blockIdx.x * GM_T;
    const int z = blockIdx.z, bz = z / splits, sp = z - bz * splits;
    const long kbeg = (long)sp * kper, kend = min(K, kbeg + kper);
    A.base += (long)bz * a_bstride;
    B.base += (long)bz * b_bstride;
    const int mrows = min(GM_T, M - m0), ncols = min(GM_T, N - n0);

    gm_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = gm_f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    GmRegs ra, rb;
    const long nchunks = (kend - kbeg + GM_KC - 1) / GM_KC;
    if (nchunks > 0) {
        gm_load<A_KCONT>(ra, A, m0, mrows, kbeg, kend, vec_a);
        gm_load<B_KCONT>(rb, B, n0, ncols, kbeg, kend, vec_b);
        gm_store<A_KCONT>(ra, gm_smem);
        gm_store<B_KCONT>(rb, gm_smem + GM_T * GM_S);
    }
    __syncthreads();
    for (long c = 0; c < nchunks; ++c) {
        const float *as = gm_smem + (c & 1) * GM_STAGE, *bs = as + GM_T * GM_S;
        const bool more = c + 1 < nchunks;
        if (more) {  // in flight while this chunk is multiplied
            gm_load<A_KCONT>(ra, A, m0, mrows, kbeg + (c + 1) * GM_KC, kend, vec_a);
            gm_load<B_KCONT>(rb, B, n0, ncols, kbeg + (c + 1) * GM_KC, kend, vec_b);
        }
        // this lane's 8 k-values (k = kh*8 + t) of its two A rows and two B columns
        float av[2][8], bv[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ar = wm * 64 + i * 32 + pl, br = wn * 64 + i * 32 + pl;
            if (A_KCONT) {
                const float4 lo = *reinterpret_cast<const float4 *>(as + ar * GM_S + kh * 8);
                const float4 hi = *reinterpret_cast<const float4 *>(as + ar * GM_S + kh * 8 + 4);
                av[i][0] = lo.x; av[i][1] = lo.y; av[i][2] = lo.z; av[i][3] = lo.w;
                av[i][4] = hi.x; av[i][5] = hi.y; av[i][6] = hi.z; av[i][7] = hi.w;
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) av[i][t] = as[(kh * 8 + t) * GM_T + ar];
            }
            if (B_KCONT) {
                const float4 lo = *reinterpret_cast<const float4 *>(bs + br * GM_S + kh * 8);
                const float4 hi = *reinterpret_cast<const float4 *>(bs + br * GM_S + kh * 8 + 4);
                bv[i][0] = lo.x; bv[i][1] = lo.y; bv[i][2] = lo.z; bv[i][3] = lo.w;
                bv[i][4] = hi.x; bv[i][5] = hi.y; bv[i][6] = hi.z; bv[i][7] = hi.w;
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) bv[i][t] = bs[(kh * 8 + t) * GM_T + br];
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][t], bv[j][t], acc[i][j], 0, 0, 0);
        if (more) {
            float *nas = gm_smem + ((c + 1) & 1) * GM_STAGE;
            gm_store<A_KCONT>(ra, nas);
            gm_store<B_KCONT>(rb, nas + GM_T * GM_S);
        }
        __syncthreads();
    }
    // accumulator layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float *C = cbase + (long)z * czstride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + pl;
            if (n < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (m < M) C[(long)m * ldc + n] = acc[i][j][r];
                }
            }
        }
}


// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the deep layers, streaming form:  dW[m][n] = sum over positions p of dY[m][p] X[n][p].
// Both operands are read along their contiguous axis and nothing else is read: the pass is HBM-bound (SA2 of PointNeXt-S:
// 295 MB for 6.4 GF), and gm_gemm_kernel<true, true> served it at 1.4 TB/s -- 64-byte row pieces per 16-position chunk, a
// 128 x 128 tile of which a 64-wide layer uses half, a thousand 32 KB partials.  Here
//   * a chunk is 32 positions: every row contributes one whole 128-byte line per chunk;
//   * the tile is 128 x TN with TN = 64 or 128 (template): a wave owns 32 rows of dY and all TN rows of X, TN/32
//     accumulators -- no MFMA issued for channels that do not exist;
//   * [row][k] LDS images with a 36-float stride (8 consecutive rows x 16 bytes cover the banks once: ds_read_b128 fragments),
//     double-buffered, the next chunk's six global loads per thread in flight during the MFMAs, one barrier per chunk;
//   * position ranges long enough for ~2 workgroups per CU (fewer, larger partials; same fixed-order reduction).
// k-pairing of v_mfma_f32_32x32x2_f32 as in gm_gemm_kernel: step t multiplies positions t (lanes 0-31) and 16 + t (32-63).
// Requires P % 4 == 0 and 16-byte aligned operands (float4 loads); anything else keeps the generic kernel.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int GW_KC = 32, GW_S = 36, GW_TM = 128;

template <int TN>
__global__ __launch_bounds__(256) void gw_wgrad_kernel(int M, int N, long P, int splits, long kper, const float *__restrict__ dy,
                                                       const float *__restrict__ x, float *__restrict__ partial)
{
    constexpr int ROWS = GW_TM + TN;               // dY rows then X rows of the tile
    constexpr int LPT = ROWS * (GW_KC / 4) / 256;  // float4 loads per thread and chunk: 6 (TN = 64) or 8 (TN = 128)
    constexpr int NJ = TN / 32;
    extern __shared__ __attribute__((aligned(16))) float gw_smem[];  // 2 stages x ROWS x GW_S
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pl = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * GW_TM, n0 = blockIdx.x * TN;
    const int z = blockIdx.z, bz = z / splits, sp = z - bz * splits;
    const long kbeg = (long)sp * kper, kend = min(P, kbeg + kper);
    const float *A = dy + (size_t)bz * M * P, *B = x + (size_t)bz * N * P;

    gm_f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = gm_f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    float4 r[LPT];
    auto load = [&](long k0) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int row = threadIdx.x / 8 + 32 * i, c4 = threadIdx.x % 8;
            const long k = k0 + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kend) {  // kend - kbeg is a multiple of 4 except at the end of P, and P % 4 == 0: whole float4s
                if (row < GW_TM) { if (m0 + row < M) v = *reinterpret_cast<const float4 *>(A + (size_t)(m0 + row) * P + k); }
                else if (n0 + row - GW_TM < N) v = *reinterpret_cast<const float4 *>(B + (size_t)(n0 + row - GW_TM) * P + k);
            }
            r[i] = v;
        }
    };
    auto store = [&](float *stage) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int row = threadIdx.x / 8 + 32 * i, c4 = threadIdx.x % 8;
            *reinterpret_cast<float4 *>(stage + row * GW_S + c4 * 4) = r[i];
        }
    };
    const long nchunks = (kend - kbeg + GW_KC - 1) / GW_KC;
    if (nchunks > 0) { load(kbeg); store(gw_smem); }
    __syncthreads();
    for (long c = 0; c < nchunks; ++c) {
        const float *as = gw_smem + (c & 1) * ROWS * GW_S, *bs = as + GW_TM * GW_S;
        const bool more = c + 1 < nchunks;
        if (more) load(kbeg + (c + 1) * GW_KC);  // in flight while this chunk is multiplied
        // this lane's 16 positions (kh * 16 + t) of its dY row and of its X rows, in two halves of 8 to bound registers
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float av[8], bv[NJ][8];
            {
                const float *pa = as + (wave * 32 + pl) * GW_S + kh * 16 + h * 8;
                const float4 lo = *reinterpret_cast<const float4 *>(pa), hi = *reinterpret_cast<const float4 *>(pa + 4);
                av[0] = lo.x; av[1] = lo.y; av[2] = lo.z; av[3] = lo.w; av[4] = hi.x; av[5] = hi.y; av[6] = hi.z; av[7] = hi.w;
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float *pb = bs + (j * 32 + pl) * GW_S + kh * 16 + h * 8;
                const float4 lo = *reinterpret_cast<const float4 *>(pb), hi = *reinterpret_cast<const float4 *>(pb + 4);
                bv[j][0] = lo.x; bv[j][1] = lo.y; bv[j][2] = lo.z; bv[j][3] = lo.w;
                bv[j][4] = hi.x; bv[j][5] = hi.y; bv[j][6] = hi.z; bv[j][7] = hi.w;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[j][t], acc[j], 0, 0, 0);
        }
        if (more) store(gw_smem + ((c + 1) & 1) * ROWS * GW_S);
        __syncthreads();
    }
    // accumulator layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float *C = partial + (size_t)z * M * N;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + j * 32 + pl;
        if (n < N) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int m = m0 + wave * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kh;
                if (m < M) C[(size_t)m * N + n] = acc[j][rr];
            }
        }
    }
}

static int aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

template <bool AK, bool BK>
static void gm_launch(int M, int N, long K, int batch, int splits, long kper, GemmView A, long abs_, GemmView B, long bbs,
                      float *c, long czs, long ldc, int va, int vb, hipStream_t stream)
{
    const size_t lds = 2 * GM_STAGE * sizeof(float);
    hipLaunchKernelGGL((gm_gemm_kernel<AK, BK>), dim3(div_up(N, GM_T), div_up(M, GM_T), batch * splits), dim3(256), lds, stream, M,
                       N, K, splits, kper, A, abs_, B, bbs, c, czs, ldc, va, vb);
}

bool gemm_conv_pays(int cin, int cout) { return cin >= 64 && cout >= 64; }

// y (b,cout,P) = W (cout,cin) . x (b,cin,P)
// ---- short layers (SetAbstraction 4, the coarse FeaturePropagation stages: a few hundred positions per cloud, 256-768
// channels): a handful of 128 x 128 tiles per cloud, each walking the whole K axis -- 48 workgroups for 50-70 us.  The K axis
// is split over several workgroups (the kernel's `splits`), the partial products land in the caller's workspace and are
// summed in a fixed order: deterministic, and the chip is filled.
static int gemm_short_splits(int b, int M, long N, long K)
{
    static const bool off = getenv("AMC3D_NO_SPLIT_K") != nullptr;
    const long tiles = (long)div_up(M, GM_T) * div_up(N, GM_T) * b;
    if (off || tiles >= 192 || K < 128) return 1;
    long s = 512 / tiles;
    if (s > 8) s = 8;
    if (s > K / 64) s = K / 64;  // at least four chunks per split
    return s < 2 ? 1 : (int)s;
}

__global__ __launch_bounds__(256) void gm_split_reduce_kernel(long total, long P, int splits, const float *__restrict__ partial,
                                                              const float *__restrict__ bias, float *__restrict__ out)
{
    // out[b][i] = sum_s partial[b * splits + s][i] (+ bias[i / P]); grid (chunks of total, b); total % 4 == 0 is not assumed
    const long i = ((long)blockIdx.x * 256 + threadIdx.x);
    if (i >= total) return;
    const float *p = partial + (size_t)blockIdx.y * splits * total + i;
    float acc = p[0];
    for (int sidx = 1; sidx < splits; ++sidx) acc += p[(size_t)sidx * total];
    out[(size_t)blockIdx.y * total + i] = bias ? acc + bias[i / P] : acc;
}

__global__ __launch_bounds__(256) void gm_bias_kernel(long total, long P, const float *__restrict__ bias, float *__restrict__ y)
{
    const long i = ((long)blockIdx.x * 256 + threadIdx.x);
    if (i < total) y[(size_t)blockIdx.y * total + i] += bias[i / P];
}

bool gemm_conv_short(int b, int rows, long P) { return (long)div_up(rows, GM_T) * div_up(P, GM_T) * b < 192; }

size_t gemm_conv_forward_workspace_bytes(int b, int cin, int cout, long P)
{
    const int sp = gemm_short_splits(b, cout, P, cin);
    return sp > 1 ? (size_t)b * sp * cout * P * sizeof(float) : 0;
}

size_t gemm_conv_backward_data_workspace_bytes(int b, int cin, int cout, long P)
{
    const int sp = gemm_short_splits(b, cin, P, cout);
    return sp > 1 ? (size_t)b * sp * cin * P * sizeof(float) : 0;
}

int gemm_conv_forward(int b, int cin, int cout, long P, const float *x, const float *w, const float *bias, float *y,
                      float *partial, hipStream_t stream)
{
    GemmView A{w, cin, 1}, B{x, 1, P};  // A(m=co,k=ci) k-contiguous; B(n=p,k=ci) at x[k*P + n]: n-contiguous
    const int va = cin % 4 == 0 && aligned16(w), vb = P % 4 == 0 && aligned16(x);
    const int sp = partial ? gemm_short_splits(b, cout, P, cin) : 1;
    if (sp > 1) {
        const long kper = (((long)cin + sp - 1) / sp + GM_KC - 1) / GM_KC * GM_KC;
        gm_launch<true, false>(cout, (int)P, cin, b, sp, kper, A, 0, B, (long)cin * P, partial, (long)cout * P, P, va, vb, stream);
        const long total = (long)cout * P;
        hipLaunchKernelGGL(gm_split_reduce_kernel, dim3(div_up(total, 256), b), dim3(256), 0, stream, total, P, sp,
                           (const float *)partial, bias, y);
        return launch_status("gemm_conv_forward");
    }
    gm_launch<true, false>(cout, (int)P, cin, b, 1, cin, A, 0, B, (long)cin * P, y, (long)cout * P, P, va, vb, stream);
    if (bias) {
        const long total = (long)cout * P;
        hipLaunchKernelGGL(gm_bias_kernel, dim3(div_up(total, 256), b), dim3(256), 0, stream, total, P, bias, y);
    }
    return launch_status("gemm_conv_forward");
}

// dx (b,cin,P) = W^T . dy (b,cout,P)
int gemm_conv_backward_data(int b, int cin, int cout, long P, const float *dy, const float *w, float *dx, float *partial,
                            hipStream_t stream)
{
    GemmView A{w, 1, cin}, B{dy, 1, P};  // A(m=ci,k=co) = w[k*cin + m]: m-contiguous
    const int va = cin % 4 == 0 && aligned16(w), vb = P % 4 == 0 && aligned16(dy);
    const int sp = partial ? gemm_short_splits(b, cin, P, cout) : 1;
    if (sp > 1) {
        const long kper = (((long)cout + sp - 1) / sp + GM_KC - 1) / GM_KC * GM_KC;
        gm_launch<false, false>(cin, (int)P, cout, b, sp, kper, A, 0, B, (long)cout * P, partial, (long)cin * P, P, va, vb, stream);
        const long total = (long)cin * P;
        hipLaunchKernelGGL(gm_split_reduce_kernel, dim3(div_up(total, 256), b), dim3(256), 0, stream, total, P, sp,
                           (const float *)partial, (const float *)nullptr, dx);
        return launch_status("gemm_conv_backward_data");
    }
    gm_launch<false, false>(cin, (int)P, cout, b, 1, cout, A, 0, B, (long)cout * P, dx, (long)cin * P, P, va, vb, stream);
    return launch_status("gemm_conv_backward_data");
}

static bool gw_streaming(int b, int cin, int cout, long P)
{
    static const bool off = getenv("AMC3D_NO_STREAMING_WGRAD") != nullptr;
    return !off && P % 4 == 0 && P >= 4 * GW_KC;
}

static int gemm_wgrad_splits(int b, int cin, int cout, long P, long *kper)
{
    if (gw_streaming(b, cin, cout, P)) {
        // ~512 workgroups (two per CU), at least `minc` chunks each; a range is a whole number of chunks
        static const long minc = getenv("AMC3D_GW_MINCHUNKS") ? atol(getenv("AMC3D_GW_MINCHUNKS")) : 4;
        const int tn = cin <= 64 ? 64 : 128;
        const long tiles = (long)div_up(cout, GW_TM) * div_up(cin, tn);
        long s = 512 / (tiles * b);
        const long cap = P / (minc * GW_KC) > 1 ? P / (minc * GW_KC) : 1;
        if (s > cap) s = cap;
        if (s < 1) s = 1;
        const long per = ((P + s - 1) / s + GW_KC - 1) / GW_KC * GW_KC;
        *kper = per;
        return (int)((P + per - 1) / per);
    }
    // enough workgroups to fill the chip, at least 256 positions each
    const long tiles = (long)div_up(cout, GM_T) * div_up(cin, GM_T);
    long s = 1024 / (tiles * b);
    static const long mink = getenv("AMC3D_WGRAD_MINK") ? atol(getenv("AMC3D_WGRAD_MINK")) : 64;  // (short layers: 375 positions)
    const long cap = P / mink > 1 ? P / mink : 1;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    long per = ((P + s - 1) / s + GM_KC - 1) / GM_KC * GM_KC;
    *kper = per;
    return (int)((P + per - 1) / per);
}

size_t gemm_conv_wgrad_workspace_bytes(int b, int cin, int cout, long P)
{
    long kper;
    const int s = gemm_wgrad_splits(b, cin, cout, P, &kper);
    return (size_t)b * s * cout * cin * sizeof(float);
}

// dw (cout,cin) = sum_b dy[b] . x[b]^T, deterministic (partials summed in a fixed order)
int gemm_conv_backward_weight(int b, int cin, int cout, long P, const float *x, const float *dy, float *dw, float *partial,
                              hipStream_t stream)
{
    long kper;
    const int s = gemm_wgrad_splits(b, cin, cout, P, &kper);
    if (gw_streaming(b, cin, cout, P) && aligned16(dy) && aligned16(x)) {
        if (cin <= 64) {
            const size_t lds = 2 * (GW_TM + 64) * GW_S * sizeof(float);
            hipLaunchKernelGGL(gw_wgrad_kernel<64>, dim3(div_up(cin, 64), div_up(cout, GW_TM), b * s), dim3(256), lds, stream, cout,
                               cin, P, s, kper, dy, x, partial);
        } else {
            const size_t lds = 2 * (GW_TM + 128) * GW_S * sizeof(float);
            (void)hipFuncSetAttribute((const void *)gw_wgrad_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(gw_wgrad_kernel<128>, dim3(div_up(cin, 128), div_up(cout, GW_TM), b * s), dim3(256), lds, stream,
                               cout, cin, P, s, kper, dy, x, partial);
        }
        if (int st = launch_status("gemm_conv_backward_weight")) return st;
        return reduce_partials(cout * cin, b * s, partial, dw, stream);
    }
    GemmView A{dy, P, 1}, B{x, P, 1};  // A(m=co,k=p), B(n=ci,k=p): both k-contiguous
    const int va = P % 4 == 0 && aligned16(dy), vb = P % 4 == 0 && aligned16(x);
    gm_launch<true, true>(cout, cin, P, b, s, kper, A, (long)cout * P, B, (long)cin * P, partial, (long)cout * cin, cin, va, vb,
                          stream);
    if (int st = launch_status("gemm_conv_backward_weight")) return st;
    return reduce_partials(cout * cin, b * s, partial, dw, stream);
}

}  // namespace amc
