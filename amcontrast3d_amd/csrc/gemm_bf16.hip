// bf16-compute / fp32-accumulate GEMM for the pointwise (1x1) convolutions under mixed precision (gfx950).
//
// Reference: use_amp wraps model and criterion in torch.cuda.amp.autocast (examples/segmentation/main_AA.py:389-394;
// BASELINE config 5 asks for bf16): every Conv1d / Conv2d then multiplies bf16 operands with fp32 accumulation.  With the
// grouped convolutions convolved before the gather (lagg.hip) every dense contraction of the model is a pointwise conv on
// the N points of a stage, i.e. the three products below; on fp32 MFMA (v_mfma_f32_32x32x2_f32, 64 cycles for 4 KFLOP)
// they bound PointNeXt-XL, on v_mfma_f32_32x32x16_bf16 (32 cycles for 32 KFLOP) they run at 16x the rate.
//     forward        Y[b]  = W        . X[b]     M = Cout, N = P,   K = Cin
//     backward-data  dX[b] = W^T      . dY[b]    M = Cin,  N = P,   K = Cout
//     backward-weight dW   = sum_b dY[b] . X[b]^T  M = Cout, N = Cin, K = P (split over workgroups, fixed-order reduce)
// Tensors stay fp32 in HBM (parameters, activations, gradients: what the fp32 kernels around this one read and write);
// operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) as they are staged into LDS, products are
// exact in fp32 and accumulated in fp32 -- the arithmetic autocast's bf16 convolutions perform.
//   * 128 x 128 output tile per workgroup, 2 x 2 waves of 64 x 64 (four 32x32 accumulators per wave);
//   * K chunks of 32 (two MFMA k-steps), both operands kept [row][k] in LDS as bf16 with a row stride of 40 elements
//     (80 bytes: 16 lanes x 16 bytes of consecutive rows cover the 64 banks once), so a lane's 8 k-values of a step
//     are one ds_read_b128; an operand that is contiguous along its row index in memory is transposed on the way in
//     (two k rows per thread, one packed 32-bit LDS store per element pair);
//   * register-staged double buffering as in gemm.hip: the next chunk's global loads are issued before the current
//     chunk's MFMAs and written to the other LDS buffer after them.
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

namespace amc {

typedef float gb_f32x16 __attribute__((ext_vector_type(16)));
typedef short gb_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int GB_T = 128;     // tile edge (M and N)
constexpr int GB_KC = 32;     // K chunk
constexpr int GB_S = 40;      // LDS row stride in bf16 elements
constexpr int GB_STAGE = 2 * GB_T * GB_S;  // bf16 elements per stage (A then B)

int reduce_partials(int total, int nparts, const float *partial, float *out, hipStream_t stream);  // gcc.hip

struct GbView {  // element (r, k) of an operand at base[r * sr + k * sk]; exactly one of sr, sk is 1
    const float *base;
    long sr, sk;
};

__device__ __forceinline__ unsigned gb_pack(float lo, float hi)
{
    const __hip_bfloat16 a = __float2bfloat16(lo), b = __float2bfloat16(hi);
    return (unsigned)(*reinterpret_cast<const unsigned short *>(&a)) | ((unsigned)(*reinterpret_cast<const unsigned short *>(&b)) << 16);
}

// registers of one chunk of one operand: 128 rows x 32 k floats, sixteen per thread
struct GbRegs {
    float f[16];
};

// KCONT (contiguous along k in memory): thread t -> row = t / 8 + 32 j, k = (t % 8) * 4 .. + 3  (one float4 per j)
// else (contiguous along the row index): item j of a thread is ONE row and one k pair (2 kp, 2 kp + 1):
//   half-wave hw = t / 32 (0..7), r8 = t % 8, q4 = (t / 8) % 4;  row = 8 (hw + 8 (j & 1)) + r8,  kp = 4 (j >> 1) + q4
// so that a half-wave's packed 32-bit LDS stores cover 8 consecutive rows x 4 consecutive words: with the 80-byte row
// stride those are 32 different banks (rows r and r + 8 would collide), and the two halves of a wave read 16
// consecutive rows = one 64-byte segment per k row from memory.
template <bool KCONT>
__device__ __forceinline__ void gb_load(GbRegs &r, const GbView &g, int row0, int nrows, long k0, long kend, bool vec)
{
    const int t = threadIdx.x;
    if (KCONT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int row = t / 8 + 32 * j;
            const long k = k0 + (t % 8) * 4;
            if (row < nrows && k < kend) {
                const float *p = g.base + (long)(row0 + row) * g.sr + k;
                if (vec && k + 3 < kend) v = *reinterpret_cast<const float4 *>(p);
                else { v.x = p[0]; if (k + 1 < kend) v.y = p[1]; if (k + 2 < kend) v.z = p[2]; if (k + 3 < kend) v.w = p[3]; }
            }
            r.f[4 * j] = v.x; r.f[4 * j + 1] = v.y; r.f[4 * j + 2] = v.z; r.f[4 * j + 3] = v.w;
        }
    } else {
        const int hw = t >> 5, r8 = t & 7, q4 = (t >> 3) & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = 8 * (hw + 8 * (j & 1)) + r8;
            const long k = k0 + 2 * (4 * (j >> 1) + q4);
            float a = 0.f, b = 0.f;
            if (row < nrows) {
                const float *p = g.base + k * g.sk + (row0 + row);
                if (k < kend) a = p[0];
                if (k + 1 < kend) b = p[g.sk];
            }
            r.f[2 * j] = a; r.f[2 * j + 1] = b;
        }
    }
}

template <bool KCONT>
__device__ __forceinline__ void gb_store(const GbRegs &r, unsigned short *lds)
{
    const int t = threadIdx.x;
    if (KCONT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = t / 8 + 32 * j, kk = (t % 8) * 4;
            uint2 w;
            w.x = gb_pack(r.f[4 * j], r.f[4 * j + 1]);
            w.y = gb_pack(r.f[4 * j + 2], r.f[4 * j + 3]);
            *reinterpret_cast<uint2 *>(lds + row * GB_S + kk) = w;
        }
    } else {
        const int hw = t >> 5, r8 = t & 7, q4 = (t >> 3) & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = 8 * (hw + 8 * (j & 1)) + r8, kk = 2 * (4 * (j >> 1) + q4);
            *reinterpret_cast<unsigned *>(lds + row * GB_S + kk) = gb_pack(r.f[2 * j], r.f[2 * j + 1]);
        }
    }
}

// grid: (N tiles, M tiles, batch * splits); as gm_gemm_kernel (gemm.hip)
template <bool A_KCONT, bool B_KCONT>
__global__ __launch_bounds__(256) void gb_gemm_kernel(int M, int N, long K, int splits, long kper, GbView A, long a_bstride,
                                                      GbView B, long b_bstride, const float *__restrict__ bias,
                                                      float *__restrict__ cbase, long czstride, long ldc, int vec_a, int vec_b)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short gb_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pl = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GB_T, n0 = blockIdx.x * GB_T;
    const int z = blockIdx.z, bz = z / splits, sp = z - bz * splits;
    const long kbeg = (long)sp * kper, kend = min(K, kbeg + kper);
    A.base += (long)bz * a_bstride;
    B.base += (long)bz * b_bstride;
    const int mrows = min(GB_T, M - m0), ncols = min(GB_T, N - n0);

    gb_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = gb_f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    GbRegs ra, rb;
    const long nchunks = (kend - kbeg + GB_KC - 1) / GB_KC;
    if (nchunks > 0) {
        gb_load<A_KCONT>(ra, A, m0, mrows, kbeg, kend, vec_a);
        gb_load<B_KCONT>(rb, B, n0, ncols, kbeg, kend, vec_b);
        gb_store<A_KCONT>(ra, gb_smem);
        gb_store<B_KCONT>(rb, gb_smem + GB_T * GB_S);
    }
    __syncthreads();
    for (long c = 0; c < nchunks; ++c) {
        const unsigned short *as = gb_smem + (c & 1) * GB_STAGE, *bs = as + GB_T * GB_S;
        const bool more = c + 1 < nchunks;
        if (more) {  // in flight while this chunk is multiplied
            gb_load<A_KCONT>(ra, A, m0, mrows, kbeg + (c + 1) * GB_KC, kend, vec_a);
            gb_load<B_KCONT>(rb, B, n0, ncols, kbeg + (c + 1) * GB_KC, kend, vec_b);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {  // two k-steps of 16: lane (r, h) holds k = 16 ks + 8 h + 0..7 of its row
            gb_bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const gb_bf16x8 *>(as + (wm * 64 + i * 32 + pl) * GB_S + 16 * ks + 8 * kh);
                bf[i] = *reinterpret_cast<const gb_bf16x8 *>(bs + (wn * 64 + i * 32 + pl) * GB_S + 16 * ks + 8 * kh);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            unsigned short *nas = gb_smem + ((c + 1) & 1) * GB_STAGE;
            gb_store<A_KCONT>(ra, nas);
            gb_store<B_KCONT>(rb, nas + GB_T * GB_S);
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
                    if (m < M) C[(long)m * ldc + n] = bias ? acc[i][j][r] + bias[m] : acc[i][j][r];
                }
            }
        }
}

static int gb_aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

template <bool AK, bool BK>
static void gb_launch(int M, int N, long K, int batch, int splits, long kper, GbView A, long abs_, GbView B, long bbs,
                      const float *bias, float *c, long czs, long ldc, int va, int vb, hipStream_t stream)
{
    const size_t lds = 2 * GB_STAGE * sizeof(unsigned short);
    hipLaunchKernelGGL((gb_gemm_kernel<AK, BK>), dim3(div_up(N, GB_T), div_up(M, GB_T), batch * splits), dim3(256), lds, stream, M,
                       N, K, splits, kper, A, abs_, B, bbs, bias, c, czs, ldc, va, vb);
}

static int gb_wgrad_splits(int b, int cin, int cout, long P, long *kper)
{
    const long tiles = (long)div_up(cout, GB_T) * div_up(cin, GB_T);
    long s = 1024 / (tiles * b);
    const long cap = P / 512 > 1 ? P / 512 : 1;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    const long per = ((P + s - 1) / s + GB_KC - 1) / GB_KC * GB_KC;
    *kper = per;
    return (int)((P + per - 1) / per);
}

}  // namespace amc

using namespace amc;

// y (b,cout,P) fp32 = bf16(weight (cout,cin)) . bf16(x (b,cin,P)) (+ bias), fp32 accumulation
AMC_API int amc3d_pointwise_conv_forward_bf16(int b, int cin, int cout, long P, const float *x, const float *weight,
                                              const float *bias, float *y, void *stream)
{
    if (b <= 0 || P <= 0 || cout <= 0) return 0;
    if (cin <= 0 || !x || !weight || !y || P >= (1L << 31)) return bad_arg("amc3d_pointwise_conv_forward_bf16: bad argument");
    GbView A{weight, cin, 1}, B{x, 1, P};  // A(m=co,k=ci) k-contiguous; B(n=p,k=ci) at x[k*P + n]: n-contiguous
    const int va = cin % 4 == 0 && gb_aligned16(weight), vb = P % 4 == 0 && gb_aligned16(x);
    gb_launch<true, false>(cout, (int)P, cin, b, 1, cin, A, 0, B, (long)cin * P, bias, y, (long)cout * P, P, va, vb,
                           (hipStream_t)stream);
    return launch_status("amc3d_pointwise_conv_forward_bf16");
}

AMC_API size_t amc3d_pointwise_conv_workspace_bytes_bf16(int b, int cin, int cout, long P)
{
    if (b <= 0 || P <= 0 || cin <= 0 || cout <= 0) return 0;
    long kper;
    const int s = gb_wgrad_splits(b, cin, cout, P, &kper);
    return (size_t)b * s * cout * cin * sizeof(float);
}

// dx (b,cin,P) = bf16(weight)^T . bf16(dy) (NULL to skip); dweight (cout,cin) = sum_{b,p} bf16(dy) bf16(x)^T (NULL to skip),
// partial sums reduced in a fixed order
AMC_API int amc3d_pointwise_conv_backward_bf16(int b, int cin, int cout, long P, const float *x, const float *weight,
                                               const float *dy, float *dx, float *dweight, void *workspace,
                                               size_t workspace_bytes, void *stream_)
{
    if (b <= 0 || P <= 0) return 0;
    if (cin <= 0 || cout <= 0 || !weight || !dy || P >= (1L << 31)) return bad_arg("amc3d_pointwise_conv_backward_bf16: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    if (dx) {
        GbView A{weight, 1, cin}, B{dy, 1, P};  // A(m=ci,k=co) = w[k*cin + m]: m-contiguous
        const int va = cin % 4 == 0 && gb_aligned16(weight), vb = P % 4 == 0 && gb_aligned16(dy);
        gb_launch<false, false>(cin, (int)P, cout, b, 1, cout, A, 0, B, (long)cout * P, nullptr, dx, (long)cin * P, P, va, vb, stream);
    }
    if (dweight) {
        if (!x || !workspace || workspace_bytes < amc3d_pointwise_conv_workspace_bytes_bf16(b, cin, cout, P))
            return bad_arg("amc3d_pointwise_conv_backward_bf16: null pointer or workspace too small");
        long kper;
        const int s = gb_wgrad_splits(b, cin, cout, P, &kper);
        GbView A{dy, P, 1}, B{x, P, 1};  // A(m=co,k=p), B(n=ci,k=p): both k-contiguous
        const int va = P % 4 == 0 && gb_aligned16(dy), vb = P % 4 == 0 && gb_aligned16(x);
        gb_launch<true, true>(cout, cin, P, b, s, kper, A, (long)cout * P, B, (long)cin * P, nullptr, (float *)workspace,
                              (long)cout * cin, cin, va, vb, stream);
        if (int st = launch_status("amc3d_pointwise_conv_backward_bf16")) return st;
        return reduce_partials(cout * cin, b * s, (const float *)workspace, dweight, stream);
    }
    return launch_status("amc3d_pointwise_conv_backward_bf16");
}
