// Tail of a two-layer SetAbstraction block, fused and recomputed instead of materialised (gfx950).
//
// Reference (pointnext_AA.py:104-127, 164-166; models/layers/conv.py:24-102): after the first grouped conv the block runs
//     BN1 -> ReLU -> Conv2d 1x1 (C1 -> C2) -> BN2 [-> ReLU] -> max over the K = 32 neighbours
// as separate layers over (B, C, M, 32) tensors: the C2-channel pre-BN activation alone is 393 MB at the first
// stage of the benchmark, written once and re-read by the statistics, the max-pool, and three backward passes.
// Here that tensor never exists.  Every pass re-creates it tile by tile from the first conv's raw output y1:
//     stage  x1 = relu(bn1(y1 tile))  into LDS   (128 positions = 4 centroids x 32 neighbours, all C1 <= 64 channels)
//     MFMA   D[position][channel] = x1^T . W2^T  (v_mfma_f32_32x32x2_f32; positions are the 32 rows of a wave's tile,
//                                                 i.e. ONE centroid's neighbours; a lane owns one output channel)
// With that operand order a lane holds its channel's 16 of the 32 neighbour values (the other 16 sit in lane + 32),
// so per-channel sums, the max over the neighbours and its arg-max are in-register reductions:
//     mode 0  forward: statistics of BN2 (per-workgroup partial sums, fp64) AND the raw max / min of every
//             (b, c2, centroid) over its 32 neighbours.  BN2's affine and the ReLU are monotone in fp32 (every
//             operation of ((x - mean) * invstd) * gamma + beta rounds monotonically), so
//             max_k relu(bn2(z_k)) = relu(bn2(max_k z_k)) for gamma >= 0 and relu(bn2(min_k z_k)) for gamma < 0:
//             the finalize kernel that turns the sums into mean / invstd also writes the pooled (B, C2, M) output
//             from the raw extrema -- one recomputation pass forward, not two
//     mode 2  backward statistics                -> sum dq, sum dq * xhat (dq: pooled gradient at the arg-max)
//     mode 3  backward: dz = BN2-backward(dq) per element (dense), then on the same tile
//               dx1 = W2^T . dz   (written, (B, C1, M, 32))   and   dW2 += dz . x1^T  (per-workgroup partials)
// The recomputation is bit-identical in every pass (same instruction sequence).  Modes 2 and 3 find the arg-max
// themselves: the first neighbour (ascending index, torch.max's rule) attaining the maximum of the normalised values --
// exactly the element the reference's max-pool routes the gradient to, also when two raw values round to the same
// normalised value or a ReLU clamps several to zero.  BN1's own backward (from dx1) stays with bn.hip.
#include <stdlib.h>

#include "common.h"

namespace amc {

typedef float sat_f32x16 __attribute__((ext_vector_type(16)));

constexpr int SAT_TP = 128;          // positions per tile: 4 centroids x 32 neighbours
constexpr int SAT_XS = SAT_TP + 1;   // odd row stride of the position-contiguous LDS images
constexpr int SAT_TILES = 8;         // tiles per workgroup
constexpr int SAT_MAX_C1 = 64, SAT_MAX_C2 = 128;

int reduce_partials(int total, int nparts, const float *partial, float *out, hipStream_t stream);  // gcc.hip

struct SatArgs {
    int B, C1, C2, M;                      // K is fixed at 32
    const float *y1;                       // (B, C1, M, 32) raw output of the first conv
    const float *mean1, *invstd1, *g1, *b1;  // BN1 (batch statistics of y1) -- followed by ReLU
    const float *w2;                       // (C2, C1)
    const float *mean2, *invstd2, *g2, *b2;  // BN2 (modes 1-3)
    int relu2;
    // mode 0 / 2: per-workgroup partial sums, [c2][part][2] doubles
    double *partial;
    // mode 0: raw extrema over the neighbours, (B, C2, M) each
    float *rmax, *rmin;
    // mode 2 / 3
    const float *dpooled;
    const float *mean_dq, *mean_dqx;       // mode 3: per-channel means of dq and dq * xhat over all B*M*32 positions
    float *dx1;                            // mode 3: (B, C1, M, 32), or (B, M, 32, C1) when dx1_pm
    int dx1_pm;                            // mode 3: position-major rows (what the gathering backward of the first layer reads)
    unsigned char *arg_out;                // mode 2, optional: (B, C2, M) the arg-max the gradient is routed to
    float *partial_w;                      // mode 3: [part][C2][C1]
};

__device__ __forceinline__ float sat_bn(float x, float mean, float invstd, float gamma, float beta)
{
    return __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(x, mean), invstd), gamma), beta);
}

template <int NCT, int MODE, int NIT>  // NIT: 32-channel tiles of C1 (used by the weight gradient of mode 3)
__global__ __launch_bounds__(256) void sat_kernel(SatArgs a)
{
    extern __shared__ float sat_smem[];
    const int C1 = a.C1, C2 = a.C2, M = a.M;
    const int WS = C1 + 1;
    float *xs = sat_smem;                         // [C1][SAT_XS]   x1 tile, position-contiguous
    float *ws = xs + C1 * SAT_XS;                 // [C2][WS]       W2
    float *bn1 = ws + C2 * WS;                    // [4][C1]        mean, invstd, gamma, beta of BN1
    float *ts = bn1 + 4 * C1;                     // mode 3: [C2][SAT_XS] dz tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pl = lane & 31, kh = lane >> 5;
    const int b = blockIdx.y;
    const long P = (long)M * 32;
    const int ntiles = (M + 3) / 4;

    for (int i = threadIdx.x; i < C2 * C1; i += 256) ws[(i / C1) * WS + i % C1] = a.w2[i];
    for (int i = threadIdx.x; i < C1; i += 256) {
        bn1[i] = a.mean1[i]; bn1[C1 + i] = a.invstd1[i]; bn1[2 * C1 + i] = a.g1[i]; bn1[3 * C1 + i] = a.b1[i];
    }
    // this lane's channels c2 = ct*32 + pl and their BN2 constants
    float m2[NCT], is2[NCT], g2[NCT], b2[NCT], ma[NCT], mb[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int c2 = ct * 32 + pl;
        const bool ok = c2 < C2;
        m2[ct] = (MODE >= 1 && ok) ? a.mean2[c2] : 0.f;
        is2[ct] = (MODE >= 1 && ok) ? a.invstd2[c2] : 0.f;
        g2[ct] = (MODE >= 1 && ok) ? a.g2[c2] : 0.f;
        b2[ct] = (MODE >= 1 && ok) ? a.b2[c2] : 0.f;
        ma[ct] = (MODE == 3 && ok) ? a.mean_dq[c2] : 0.f;
        mb[ct] = (MODE == 3 && ok) ? a.mean_dqx[c2] : 0.f;
    }
    double acc_a[NCT], acc_b[NCT];  // modes 0 / 2: running sums of this lane's channels
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) { acc_a[ct] = 0.0; acc_b[ct] = 0.0; }
    sat_f32x16 accw[MODE == 3 ? NCT : 1][MODE == 3 ? NIT : 1];
    if (MODE == 3) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int it = 0; it < NIT; ++it) accw[ct][it] = sat_f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    }

    // C1 <= 32: the raw y1 of the NEXT tile is fetched into four float4 registers while this tile is multiplied
    // (two to four workgroups share a CU: too few to hide 2-3 us of load latency per tile behind each other's MFMA work)
    const bool PRE = C1 <= 32;  // (workgroup-uniform)
    float4 pre[4];
    auto fetch = [&](int tile_) {
        const long q0 = (long)tile_ * SAT_TP;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = threadIdx.x + j * 256;
            const int k = i / (SAT_TP / 4), c4 = i - k * (SAT_TP / 4);
            const long p = q0 + c4 * 4;
            pre[j] = (k < C1 && tile_ < ntiles && p < P) ? *reinterpret_cast<const float4 *>(a.y1 + ((size_t)b * C1 + k) * P + p)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (PRE) fetch(blockIdx.x * SAT_TILES);

    for (int tt = 0; tt < SAT_TILES; ++tt) {
        const int tile = blockIdx.x * SAT_TILES + tt;
        if (tile >= ntiles) break;  // workgroup-uniform
        const long p0 = (long)tile * SAT_TP;
        __syncthreads();  // previous tile consumed (and ws / bn1 written, first time round)
        // ---- stage x1 = relu(bn1(y1)): 16-byte loads along the positions --------------------------------------
        if (PRE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = threadIdx.x + j * 256;
                const int k = i / (SAT_TP / 4), c4 = i - k * (SAT_TP / 4);
                if (k < C1) {
                    float4 v = pre[j];
                    if (p0 + c4 * 4 < P) {  // P is a multiple of 32: the four elements are in range together
                        const float mu = bn1[k], is = bn1[C1 + k], g = bn1[2 * C1 + k], bt = bn1[3 * C1 + k];
                        v.x = fmaxf(sat_bn(v.x, mu, is, g, bt), 0.f); v.y = fmaxf(sat_bn(v.y, mu, is, g, bt), 0.f);
                        v.z = fmaxf(sat_bn(v.z, mu, is, g, bt), 0.f); v.w = fmaxf(sat_bn(v.w, mu, is, g, bt), 0.f);
                    }
                    float *d = xs + k * SAT_XS + c4 * 4;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                }
            }
            if (tt + 1 < SAT_TILES) fetch(tile + 1);  // in flight during the products below
        } else {
            for (int i = threadIdx.x; i < C1 * (SAT_TP / 4); i += 256) {
                const int k = i / (SAT_TP / 4), c4 = i - k * (SAT_TP / 4);
                const long p = p0 + c4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p < P) {  // P is a multiple of 32: the four elements are in range together
                    v = *reinterpret_cast<const float4 *>(a.y1 + ((size_t)b * C1 + k) * P + p);
                    const float mu = bn1[k], is = bn1[C1 + k], g = bn1[2 * C1 + k], bt = bn1[3 * C1 + k];
                    v.x = fmaxf(sat_bn(v.x, mu, is, g, bt), 0.f); v.y = fmaxf(sat_bn(v.y, mu, is, g, bt), 0.f);
                    v.z = fmaxf(sat_bn(v.z, mu, is, g, bt), 0.f); v.w = fmaxf(sat_bn(v.w, mu, is, g, bt), 0.f);
                }
                float *d = xs + k * SAT_XS + c4 * 4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
        __syncthreads();
        // ---- z[position][channel] = x1^T . W2^T: A[i = position][k] = x1, B[k][j = channel] = W2[channel][k] ---------
        sat_f32x16 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = sat_f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < C1; k += 2) {
            const float av = xs[(k + kh) * SAT_XS + wave * 32 + pl];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int c2 = ct * 32 + pl;
                const float bv = c2 < C2 ? ws[c2 * WS + k + kh] : 0.f;
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ct], 0, 0, 0);
            }
        }
        // register r of a lane = neighbour s(r) = (r & 3) + 8 (r >> 2) + 4 kh of centroid m = tile*4 + wave
        const int m = tile * 4 + wave;
        const bool live = m < M;

        if (MODE == 0) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s1 += acc[ct][r]; s2 += acc[ct][r] * acc[ct][r]; }
                if (live) { acc_a[ct] += (double)s1; acc_b[ct] += (double)s2; }
                float mx = acc[ct][0], mn = acc[ct][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) { mx = fmaxf(mx, acc[ct][r]); mn = fminf(mn, acc[ct][r]); }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                mn = fminf(mn, __shfl_xor(mn, 32, 64));
                const int c2 = ct * 32 + pl;
                if (live && kh == 0 && c2 < C2) {
                    a.rmax[((size_t)b * C2 + c2) * M + m] = mx;
                    a.rmin[((size_t)b * C2 + c2) * M + m] = mn;
                }
            }
        }
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int c2 = ct * 32 + pl;
                const bool ok = live && c2 < C2;
                const float gq = ok ? a.dpooled[((size_t)b * C2 + c2) * M + m] : 0.f;
                // arg-max of the forward max-pool: first neighbour attaining the maximum of [relu](bn2(z))
                float best = -__builtin_inff();
                int as = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {  // ascending s within this half: strict '>' keeps the first maximum
                    float y = sat_bn(acc[ct][r], m2[ct], is2[ct], g2[ct], b2[ct]);
                    if (a.relu2) y = fmaxf(y, 0.f);
                    if (y > best) { best = y; as = (r & 3) + 8 * (r >> 2) + 4 * kh; }
                }
                {
                    const float ob = __shfl_xor(best, 32, 64);
                    const int os = __shfl_xor(as, 32, 64);
                    if (ob > best || (ob == best && os < as)) { best = ob; as = os; }  // torch.max: first index on ties
                }
                const bool owner = ok && ((as >> 2) & 1) == kh;       // the half that holds neighbour `as`
                const int rstar = (as & 3) + 4 * (as >> 3);
                if (MODE == 2) {
                    if (a.arg_out && ok && kh == 0) a.arg_out[((size_t)b * C2 + c2) * M + m] = (unsigned char)as;
                    float xv = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) xv = r == rstar ? acc[ct][r] : xv;
                    const float xh = __fmul_rn(__fsub_rn(xv, m2[ct]), is2[ct]);
                    float q = gq;
                    if (a.relu2 && !(__fadd_rn(__fmul_rn(xh, g2[ct]), b2[ct]) > 0.f)) q = 0.f;
                    if (owner) { acc_a[ct] += (double)q; acc_b[ct] += (double)q * (double)xh; }
                } else {
                    const float gi = __fmul_rn(g2[ct], is2[ct]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float xh = __fmul_rn(__fsub_rn(acc[ct][r], m2[ct]), is2[ct]);
                        float q = (owner && r == rstar) ? gq : 0.f;
                        if (a.relu2 && !(__fadd_rn(__fmul_rn(xh, g2[ct]), b2[ct]) > 0.f)) q = 0.f;
                        const float dz = ok ? gi * (q - ma[ct] - xh * mb[ct]) : 0.f;
                        if (c2 < C2) ts[c2 * SAT_XS + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh] = dz;
                    }
                }
            }
        }
        if (MODE == 3) {
            __syncthreads();  // dz tile complete
            // ---- dx1[c1][position] = sum_c2 W2[c2][c1] dz[c2][position]: A[i = c1][k = c2], B[k = c2][j = position] ------
            for (int it = 0; it < (C1 + 31) / 32; ++it) {
                sat_f32x16 accd = sat_f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                const int c1 = it * 32 + pl;
                for (int k2 = 0; k2 < C2; k2 += 2) {
                    const float av = c1 < C1 ? ws[(k2 + kh) * WS + c1] : 0.f;
                    const float bv = ts[(k2 + kh) * SAT_XS + wave * 32 + pl];
                    accd = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accd, 0, 0, 0);
                }
                const long p = p0 + wave * 32 + pl;
                if (p < P && a.dx1_pm) {
                    // a lane holds channels 8q + 4kh .. +3 of its position: four 16-byte stores, the kh pair of a
                    // position covering 32 contiguous bytes per instruction (C1 % 4 == 0, checked by the host)
                    float4 *row4 = reinterpret_cast<float4 *>(a.dx1 + ((size_t)b * P + p) * C1 + it * 32 + 4 * kh);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        if (it * 32 + 8 * q4 + 4 * kh < C1)
                            row4[2 * q4] = make_float4(accd[4 * q4], accd[4 * q4 + 1], accd[4 * q4 + 2], accd[4 * q4 + 3]);
                } else if (p < P) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = it * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        if (row < C1) a.dx1[((size_t)b * C1 + row) * P + p] = accd[r];
                    }
                }
            }
            // ---- dW2[c2][c1] += sum_position dz[c2][position] x1[c1][position] (this wave: its 32 positions) ----------
            for (int s = 0; s < 32; s += 2) {
                const int pp = wave * 32 + s + kh;
                float bv[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int c1 = it * 32 + pl;
                    bv[it] = c1 < C1 ? xs[c1 * SAT_XS + pp] : 0.f;
                }
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int c2 = ct * 32 + pl;
                    const float av = c2 < C2 ? ts[c2 * SAT_XS + pp] : 0.f;
#pragma unroll
                    for (int it = 0; it < NIT; ++it) accw[ct][it] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[it], accw[ct][it], 0, 0, 0);
                }
            }
        }
    }

    const int part = blockIdx.y * gridDim.x + blockIdx.x, nparts = gridDim.x * gridDim.y;
    if (MODE == 0 || MODE == 2) {
        // combine the two neighbour halves (lanes l, l+32) and the four waves, then one partial per channel
        __syncthreads();
        double *red = reinterpret_cast<double *>(sat_smem);  // [4][C2][2]
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            acc_a[ct] += __shfl_xor(acc_a[ct], 32, 64);
            acc_b[ct] += __shfl_xor(acc_b[ct], 32, 64);
            const int c2 = ct * 32 + pl;
            if (kh == 0 && c2 < C2) { red[(wave * C2 + c2) * 2] = acc_a[ct]; red[(wave * C2 + c2) * 2 + 1] = acc_b[ct]; }
        }
        __syncthreads();
        for (int c2 = threadIdx.x; c2 < C2; c2 += 256) {
            double s1 = 0.0, s2 = 0.0;
            for (int w = 0; w < 4; ++w) { s1 += red[(w * C2 + c2) * 2]; s2 += red[(w * C2 + c2) * 2 + 1]; }
            a.partial[((size_t)c2 * nparts + part) * 2] = s1;
            a.partial[((size_t)c2 * nparts + part) * 2 + 1] = s2;
        }
    }
    if (MODE == 3) {
        __syncthreads();
        float *red = sat_smem;  // [C2][C1]
        for (int i = threadIdx.x; i < C2 * C1; i += 256) red[i] = 0.f;
        __syncthreads();
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int c1 = it * 32 + pl;
                        if (c1 < C1) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int c2 = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                                if (c2 < C2) red[c2 * C1 + c1] += accw[ct][it][r];
                            }
                        }
                    }
            }
            __syncthreads();
        }
        float *out = a.partial_w + (size_t)part * C2 * C1;
        for (int i = threadIdx.x; i < C2 * C1; i += 256) out[i] = red[i];
    }
}

// mean / invstd / unbiased variance of BN2 from the partials (+ nn.BatchNorm's running update), or the backward means.
// One 256-thread workgroup per channel sums its partials in a fixed order.
__global__ __launch_bounds__(256) void sat_finalize_kernel(int C2, int nparts, double count, float eps, float momentum,
                                                           int backward, const double *__restrict__ partial,
                                                           float *__restrict__ o0, float *__restrict__ o1,
                                                           float *__restrict__ o2, float *__restrict__ o3,
                                                           float *__restrict__ running_mean, float *__restrict__ running_var,
                                                           long long *__restrict__ tracked, int B, int M, int relu2,
                                                           const float *__restrict__ gamma2, const float *__restrict__ beta2,
                                                           const float *__restrict__ rmax, const float *__restrict__ rmin,
                                                           float *__restrict__ pooled)
{
    __shared__ double s_a[4], s_b[4];
    __shared__ float s_stat[2];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 256) {
        s1 += partial[((size_t)c * nparts + k) * 2];
        s2 += partial[((size_t)c * nparts + k) * 2 + 1];
    }
    for (int s = 32; s >= 1; s >>= 1) { s1 += __shfl_xor(s1, s, 64); s2 += __shfl_xor(s2, s, 64); }
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = s1; s_b[threadIdx.x >> 6] = s2; }
    __syncthreads();
    // gridDim.y slices of the pooled output share a channel: every slice reduces the partials (same order, same values),
    // slice 0 publishes the statistics
    const bool first = blockIdx.y == 0;
    if (threadIdx.x == 0) {
        s1 = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
        s2 = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
        if (backward) {  // o0 = dbeta = sum dq, o1 = dgamma = sum dq xhat, o2 / o3 = their means
            if (first) { o0[c] = (float)s1; o1[c] = (float)s2; o2[c] = (float)(s1 / count); o3[c] = (float)(s2 / count); }
        } else {
            const double mu = s1 / count;
            double var = s2 / count - mu * mu;
            if (var < 0.0) var = 0.0;
            const float mf = (float)mu, vu = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
            const float isf = (float)(1.0 / sqrt(var + (double)eps));
            if (first) { o0[c] = mf; o1[c] = isf; o2[c] = vu; }
            s_stat[0] = mf; s_stat[1] = isf;
            if (first && running_mean && momentum >= 0.f) {
                running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * mf;
                running_var[c] = running_var[c] * (1.f - momentum) + momentum * vu;
                if (c == 0 && tracked) *tracked += 1;
            }
        }
    }
    if (backward || !pooled) return;
    __syncthreads();
    // the max-pool of the normalised values, from the raw extrema (monotonicity: see the file header)
    const float mf = s_stat[0], isf = s_stat[1], g = gamma2[c], bt = beta2[c];
    const float *src = g >= 0.f ? rmax : rmin;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < B * M; i += 256 * gridDim.y) {
        const int bb = i / M, m = i - bb * M;
        const size_t q = ((size_t)bb * C2 + c) * M + m;
        float y = sat_bn(src[q], mf, isf, g, bt);
        pooled[q] = relu2 ? fmaxf(y, 0.f) : y;
    }
}

static bool sat_aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }
static bool sat_supported(int C1, int C2, int K) { return K == 32 && C1 >= 2 && C1 <= SAT_MAX_C1 && C1 % 2 == 0 && C2 >= 1 && C2 <= SAT_MAX_C2; }
// the recomputation costs 2-3x the layer's MFMA work: it pays while the layer is HBM-bound (measured on MI355X:
// 32 -> 64 channels faster fused, 64 -> 128 faster layer by layer, also with the single forward pass)
static bool sat_pays(int C1, int C2)
{
    static const long lim = getenv("AMC3D_SAT_PAYS") ? atol(getenv("AMC3D_SAT_PAYS")) : 32 * 64;
    return (long)C1 * C2 <= lim;
}

static size_t sat_lds(int C1, int C2, int mode)
{
    size_t f = (size_t)C1 * SAT_XS + (size_t)C2 * (C1 + 1) + 4 * (size_t)C1 + (mode == 3 ? (size_t)C2 * SAT_XS : 0);
    const size_t red = mode == 3 ? (size_t)C2 * C1 : (size_t)4 * C2 * 4;  // floats (doubles counted as 2)
    if (f < red) f = red;
    return f * sizeof(float);
}

template <int MODE>
static void sat_launch(const SatArgs &a, int groups, hipStream_t stream)
{
    const size_t lds = sat_lds(a.C1, a.C2, MODE);
    const int nct = (a.C2 + 31) / 32;
#define AMC_SAT(N, I)                                                                                                         \
    (void)hipFuncSetAttribute((const void *)sat_kernel<N, MODE, I>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
    hipLaunchKernelGGL((sat_kernel<N, MODE, I>), dim3(groups, a.B), dim3(256), lds, stream, a)
    if (MODE == 3 && a.C1 > 32) {
        if (nct == 1) { AMC_SAT(1, 2); } else if (nct == 2) { AMC_SAT(2, 2); } else if (nct == 3) { AMC_SAT(3, 2); } else { AMC_SAT(4, 2); }
    } else {
        if (nct == 1) { AMC_SAT(1, 1); } else if (nct == 2) { AMC_SAT(2, 1); } else if (nct == 3) { AMC_SAT(3, 1); } else { AMC_SAT(4, 1); }
    }
#undef AMC_SAT
}

}  // namespace amc

using namespace amc;

AMC_API int amc3d_sa_tail_supported(int C1, int C2, int K) { return sat_supported(C1, C2, K) ? 1 : 0; }
AMC_API int amc3d_sa_tail_pays(int C1, int C2) { return sat_pays(C1, C2) ? 1 : 0; }

// bytes for the statistics partials (forward and backward) and, for backward, the weight-gradient partials
AMC_API size_t amc3d_sa_tail_workspace_bytes(int B, int C1, int C2, int M)
{
    if (B <= 0 || M <= 0) return 0;
    const size_t parts = (size_t)B * div_up(div_up(M, 4), SAT_TILES);
    const size_t shared = parts * C2 * 2 * sizeof(double);                                   // statistics partials
    const size_t fwd = 2 * (size_t)B * C2 * M * sizeof(float);                               // raw max / min
    const size_t bwd = parts * (size_t)C2 * C1 * sizeof(float) + 4 * (size_t)C2 * sizeof(float);  // dW2 partials, means
    return shared + (fwd > bwd ? fwd : bwd) + 64;
}

// pooled (B,C2,M) = max_k [relu2](bn2(W2 . relu(bn1(y1)))) with batch statistics for BN2 (returned in
// mean2, invstd2, var_unbiased2; running buffers updated when given, momentum < 0: not here)
AMC_API int amc3d_sa_tail_forward(int B, int C1, int C2, int M, int K, const float *y1, const float *mean1,
                                  const float *invstd1, const float *gamma1, const float *beta1, const float *w2,
                                  const float *gamma2, const float *beta2, float eps2, float momentum2, int relu2,
                                  float *pooled, float *mean2, float *invstd2, float *var_unbiased2,
                                  float *running_mean2, float *running_var2, long long *tracked2, void *workspace,
                                  size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || M <= 0) return 0;
    if (!sat_supported(C1, C2, K) || !y1 || !mean1 || !invstd1 || !gamma1 || !beta1 || !w2 || !gamma2 || !beta2 || !pooled ||
        !mean2 || !invstd2 || !var_unbiased2 || !workspace || !sat_aligned16(y1) ||
        workspace_bytes < amc3d_sa_tail_workspace_bytes(B, C1, C2, M))
        return bad_arg("amc3d_sa_tail_forward: unsupported shape, null pointer or workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const int groups = div_up(div_up(M, 4), SAT_TILES);
    SatArgs a{};
    a.B = B; a.C1 = C1; a.C2 = C2; a.M = M; a.y1 = y1; a.mean1 = mean1; a.invstd1 = invstd1; a.g1 = gamma1; a.b1 = beta1;
    a.w2 = w2; a.g2 = gamma2; a.b2 = beta2; a.relu2 = relu2; a.partial = (double *)workspace;
    a.rmax = (float *)((char *)workspace + (size_t)groups * B * C2 * 2 * sizeof(double));
    a.rmin = a.rmax + (size_t)B * C2 * M;
    sat_launch<0>(a, groups, stream);
    hipLaunchKernelGGL(sat_finalize_kernel, dim3(C2, (unsigned)((long)B * M >= 16384 ? 8 : 1)), dim3(256), 0, stream, C2, groups * B,
                       (double)B * (double)M * 32.0, eps2, momentum2, 0, (const double *)workspace, mean2, invstd2,
                       var_unbiased2, (float *)nullptr, running_mean2, running_var2, tracked2, B, M, relu2, gamma2, beta2,
                       (const float *)a.rmax, (const float *)a.rmin, pooled);
    return launch_status("amc3d_sa_tail_forward");
}

// dx1 (B,C1,M,32): gradient w.r.t. relu(bn1(y1)) -- BN1 / ReLU backward is the caller's (amc3d_bn_backward on y1);
// dw2 (C2,C1), dgamma2, dbeta2 (C2)
AMC_API int amc3d_sa_tail_backward(int B, int C1, int C2, int M, int K, const float *y1, const float *mean1,
                                   const float *invstd1, const float *gamma1, const float *beta1, const float *w2,
                                   const float *mean2, const float *invstd2, const float *gamma2, const float *beta2,
                                   int relu2, const float *dpooled, float *dx1, int dx1_position_major, float *dw2,
                                   float *dgamma2, float *dbeta2, unsigned char *arg_out, void *workspace,
                                   size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || M <= 0) return 0;
    if (dx1_position_major && ((C1 & 3) || !sat_aligned16(dx1)))
        return bad_arg("amc3d_sa_tail_backward: position-major dx1 needs C1 % 4 == 0 and a 16-byte aligned buffer");
    if (!sat_supported(C1, C2, K) || !y1 || !mean1 || !invstd1 || !gamma1 || !beta1 || !w2 || !mean2 || !invstd2 || !gamma2 ||
        !beta2 || !dpooled || !dx1 || !dw2 || !dgamma2 || !dbeta2 || !workspace || !sat_aligned16(y1) ||
        workspace_bytes < amc3d_sa_tail_workspace_bytes(B, C1, C2, M))
        return bad_arg("amc3d_sa_tail_backward: unsupported shape, null pointer or workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const int groups = div_up(div_up(M, 4), SAT_TILES);
    const size_t parts = (size_t)groups * B;
    double *partial = (double *)workspace;
    float *partial_w = (float *)((char *)workspace + parts * C2 * 2 * sizeof(double));
    float *means = partial_w + parts * (size_t)C2 * C1;  // mean_dq, mean_dqx
    SatArgs a{};
    a.B = B; a.C1 = C1; a.C2 = C2; a.M = M; a.y1 = y1; a.mean1 = mean1; a.invstd1 = invstd1; a.g1 = gamma1; a.b1 = beta1;
    a.w2 = w2; a.mean2 = mean2; a.invstd2 = invstd2; a.g2 = gamma2; a.b2 = beta2; a.relu2 = relu2;
    a.dpooled = dpooled; a.partial = partial; a.arg_out = arg_out;
    sat_launch<2>(a, groups, stream);
    hipLaunchKernelGGL(sat_finalize_kernel, dim3(C2), dim3(256), 0, stream, C2, groups * B,
                       (double)B * (double)M * 32.0, 0.f, -1.f, 1, (const double *)partial, dbeta2, dgamma2, means, means + C2,
                       (float *)nullptr, (float *)nullptr, (long long *)nullptr, B, M, relu2, (const float *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (float *)nullptr);
    a.mean_dq = means; a.mean_dqx = means + C2; a.dx1 = dx1; a.dx1_pm = dx1_position_major ? 1 : 0; a.partial_w = partial_w;
    sat_launch<3>(a, groups, stream);
    if (int st = launch_status("amc3d_sa_tail_backward")) return st;
    return reduce_partials(C2 * C1, (int)parts, partial_w, dw2, stream);
}
