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
    // mode 0: per (b, c2, centroid) the raw extreme over the 32 neighbours that BN2 + max-pool select (the maximum for
    // gamma2 >= 0, the minimum otherwise), (B, C2, M), and optionally the first neighbour attaining it (bytes)
    float *zext;
    unsigned char *arg_ext;
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
__global__ __launch_bounds__(256, (MODE == 0 && NCT <= 2) ? 3 : 1) void sat_kernel(SatArgs a)  // (forward: three workgroups per CU)
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
        g2[ct] = ok ? a.g2[c2] : 0.f;
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
        if ((C1 & 7) == 0) {  // four steps at a time, their LDS reads (unconditional: clamped index + select) ahead of the products
            for (int k0 = 0; k0 < C1; k0 += 8) {
                float av[4], bv[NCT][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + 2 * u + kh;
                    av[u] = xs[k * SAT_XS + wave * 32 + pl];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        const int c2 = ct * 32 + pl;
                        const float t = ws[(c2 < C2 ? c2 : 0) * WS + k];
                        bv[ct][u] = c2 < C2 ? t : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[ct][u], acc[ct], 0, 0, 0);
            }
        } else
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
                // the extreme the monotone BN2 [+ ReLU] + max-pool selects, and the first neighbour (ascending index,
                // torch.max's rule on the raw values) that attains it: what the backward routes the gradient to
                const bool up = g2[ct] >= 0.f;
                float ex = acc[ct][0];
                int as = 4 * kh;
#pragma unroll
                for (int r = 1; r < 16; ++r) {  // ascending neighbour index within this half: a strict comparison keeps the first
                    const float zr = acc[ct][r];
                    const bool better = up ? zr > ex : zr < ex;
                    ex = better ? zr : ex;
                    as = better ? (r & 3) + 8 * (r >> 2) + 4 * kh : as;
                }
                {
                    const float oe = __shfl_xor(ex, 32, 64);
                    const int os = __shfl_xor(as, 32, 64);
                    if ((up ? oe > ex : oe < ex) || (oe == ex && os < as)) { ex = oe; as = os; }
                }
                const int c2 = ct * 32 + pl;
                if (live && kh == 0 && c2 < C2) {
                    a.zext[((size_t)b * C2 + c2) * M + m] = ex;
                    if (a.arg_ext) a.arg_ext[((size_t)b * C2 + c2) * M + m] = (unsigned char)as;
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
                                                           const float *__restrict__ zext, float *__restrict__ pooled)
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
    for (int i = blockIdx.y * 256 + threadIdx.x; i < B * M; i += 256 * gridDim.y) {
        const int bb = i / M, m = i - bb * M;
        const size_t q = ((size_t)bb * C2 + c) * M + m;
        float y = sat_bn(zext[q], mf, isf, g, bt);  // (mode 0 kept the maximum for gamma >= 0, the minimum otherwise)
        pooled[q] = relu2 ? fmaxf(y, 0.f) : y;
    }
}


// =====================================================================================================================
// Backward in ALGEBRAIC form (round 3): no recomputation of z = W2 . x1 and no dense dz.
// The pooled gradient reaches ONE neighbour per (b, c2, centroid) -- q is sparse -- and BatchNorm's backward adds terms that are
// affine in z:            dz = Dq q - E (z - mu 1^T) - u 1^T,    Dq = diag(gamma is), E = diag(gamma is^2 mb), u = gamma is ma,
// ma = sum(q) / P, mb = sum(q xhat) / P over the pooled elements only (xhat at the routed neighbour follows from the raw
// extreme mode 0 kept).  With z = W2 x1:
//     dx1 = W2^T dz = W2^T Dq q  -  A x1  -  c 1^T,        A = W2^T E W2  (C1 x C1, symmetric),  c = W2^T (u - E mu)
//     dW2 = dz x1^T = Dq q x1^T  -  E W2 S  -  (u - E mu) s^T,   S = x1 x1^T (C1 x C1 Gram matrix),  s = x1 1
// so one pass over x1 does: a C1 x C1 product per position (MFMA), the Gram matrix (MFMA), and the sparse terms -- C2 rank-1
// updates per centroid instead of 32 x C2: the arg-max bytes of mode 0 say where.  Per 128-position tile and wave that is
// 2 x C1/2 x (C1/32)^2 MFMAs (32 at SA1) where modes 2 + 3 issue 32 + 96, and ~300 instead of ~850 VALU instructions; the
// second recomputation pass (mode 2) is replaced by a pass over the (B, C2, M) pooled tensors.
// =====================================================================================================================

// q, xhat at the routed neighbour; v = gamma is q and the arg bytes transposed to (B, M, C2) rows (what a wave of the main pass
// reads for its centroid); per-block partial sums of q and q xhat (fp64, fixed order)
__global__ __launch_bounds__(256) void sat_pool_grad_kernel(int C2, int M, int relu2, const float *__restrict__ g,
                                                            const float *__restrict__ zext, const unsigned char *__restrict__ arg,
                                                            const float *__restrict__ mean2, const float *__restrict__ invstd2,
                                                            const float *__restrict__ gamma2, const float *__restrict__ beta2,
                                                            float *__restrict__ vT, unsigned char *__restrict__ argT,
                                                            double *__restrict__ partial)
{
    extern __shared__ float spg_smem[];
    float *vt = spg_smem;                                                   // [64][C2 + 1]
    unsigned char *at = reinterpret_cast<unsigned char *>(vt + 64 * (C2 + 1));  // [64][C2]
    const int b = blockIdx.y, m0 = blockIdx.x * 64, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = m0 + lane;
    const bool ok = m < M;
    for (int c2 = wave; c2 < C2; c2 += 4) {
        const size_t idx = ((size_t)b * C2 + c2) * M + m;
        const float gq = ok ? g[idx] : 0.f, ze = ok ? zext[idx] : 0.f;
        const unsigned char ar = ok ? arg[idx] : (unsigned char)0;
        const float mu = mean2[c2], is = invstd2[c2], ga = gamma2[c2], be = beta2[c2];
        const float xh = __fmul_rn(__fsub_rn(ze, mu), is);
        float q = gq;
        if (relu2 && !(__fadd_rn(__fmul_rn(xh, ga), be) > 0.f)) q = 0.f;
        if (!ok) q = 0.f;
        vt[lane * (C2 + 1) + c2] = __fmul_rn(__fmul_rn(ga, is), q);
        at[lane * C2 + c2] = ar;
        float sq = q, sqx = __fmul_rn(q, xh);  // 64 values: an fp32 tree, fp64 across the blocks
        for (int o = 32; o >= 1; o >>= 1) { sq += __shfl_xor(sq, o, 64); sqx += __shfl_xor(sqx, o, 64); }
        if (lane == 0) {
            const size_t slot = ((size_t)c2 * gridDim.y + b) * gridDim.x + blockIdx.x;
            partial[slot * 2] = (double)sq;
            partial[slot * 2 + 1] = (double)sqx;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * C2; e += 256) {
        const int r = e / C2, c2 = e - r * C2;
        if (m0 + r < M) {
            const size_t o = ((size_t)b * M + m0 + r) * C2 + c2;
            vT[o] = vt[r * (C2 + 1) + c2];
            argT[o] = at[r * C2 + c2];
        }
    }
}

// dbeta2, dgamma2 and the per-channel coefficients E, t = u - E mu of the algebraic form: one 64-thread workgroup per channel
__global__ __launch_bounds__(64) void sat_bwd_chan_kernel(int nblk, double count, const double *__restrict__ partial,
                                                          const float *__restrict__ mean2, const float *__restrict__ invstd2,
                                                          const float *__restrict__ gamma2, float *__restrict__ dbeta,
                                                          float *__restrict__ dgamma, double *__restrict__ coefE,
                                                          double *__restrict__ coeft)
{
    const int c2 = blockIdx.x, lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int k = lane; k < nblk; k += 64) { s1 += partial[((size_t)c2 * nblk + k) * 2]; s2 += partial[((size_t)c2 * nblk + k) * 2 + 1]; }
    for (int o = 32; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (lane == 0) {
        dbeta[c2] = (float)s1; dgamma[c2] = (float)s2;
        const double gi = (double)gamma2[c2] * (double)invstd2[c2];
        const double E = gi * (double)invstd2[c2] * (s2 / count), u = gi * (s1 / count);
        coefE[c2] = E; coeft[c2] = u - E * (double)mean2[c2];
    }
}

// A = W2^T E W2 (row blockIdx.x) and c = W2^T t
__global__ __launch_bounds__(64) void sat_bwd_coef_kernel(int C1, int C2, const float *__restrict__ w2, const double *__restrict__ coefE,
                                                          const double *__restrict__ coeft, float *__restrict__ coefA,
                                                          float *__restrict__ coefc)
{
    __shared__ double er[SAT_MAX_C2];
    const int r = blockIdx.x;
    for (int c2 = threadIdx.x; c2 < C2; c2 += 64) er[c2] = (double)w2[c2 * C1 + r] * coefE[c2];
    __syncthreads();
    for (int c = threadIdx.x; c < C1; c += 64) {
        double s = 0.0;
#pragma unroll 8
        for (int c2 = 0; c2 < C2; ++c2) s += er[c2] * (double)w2[c2 * C1 + c];
        coefA[r * C1 + c] = (float)s;
    }
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int c2 = 0; c2 < C2; ++c2) s += (double)w2[c2 * C1 + r] * coeft[c2];
        coefc[r] = (float)s;
    }
}

#ifndef SAT_ALG_WGS
#define SAT_ALG_WGS 2
#endif
struct SatAlgArgs {
    int B, C1, C2, M;
    const float *y1, *mean1, *invstd1, *g1, *b1;  // x1 = relu(bn1(y1)) (identity BN1 for an activated input)
    const float *w2, *coefA, *coefc;
    const float *vT;                  // (B, M, C2)
    const unsigned char *argT;        // (B, M, C2)
    float *dx1;                       // (B, C1, M, 32), or (B, M, 32, C1) when dx1_pm
    int dx1_pm;
    float *partial;                   // [part][C2*C1 (sparse dW2) | C1*C1 (Gram) | C1 (sums of x1)]
};

// WR = C1 * C2 / 256: dW2 entries a lane owns (lane = (channel c2 of the wave's quarter, chunk of WR input channels))
// FULL: C1 == 32 * NIT and C2 % 16 == 0 -- every range guard below is then constant (a guarded LDS read is a branch of its own:
// the eight reads ahead of eight products would not be issued together)
template <int NIT, int WR, bool FULL>
__global__ __launch_bounds__(256, NIT == 1 ? SAT_ALG_WGS : 1) void sat_bwd_alg_kernel(SatAlgArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sat_smem[];
    const int C1 = a.C1, C2 = a.C2, M = a.M;
    const int WS = C1 + 1;
    float *xs = sat_smem;                         // [C1][SAT_XS]   x1 tile, position-contiguous
    float *as = xs + C1 * SAT_XS;                 // [C1][WS]       -A
    float *ws = as + C1 * WS;                     // [C2][WS]       W2
    float *bn1 = ws + C2 * WS;                    // [4][C1]
    float *cv = bn1 + 4 * C1;                     // [C1]           c
    float2 *va = reinterpret_cast<float2 *>(cv + C1 + (C1 & 1));  // [4][C2]  (v, routed neighbour as int bits) of the tile's four centroids
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pl = lane & 31, kh = lane >> 5;
    const int b = blockIdx.y;
    const long P = (long)M * 32;
    const int ntiles = (M + 3) / 4;

    for (int i = threadIdx.x; i < C2 * C1; i += 256) ws[(i / C1) * WS + i % C1] = a.w2[i];
    for (int i = threadIdx.x; i < C1 * C1; i += 256) as[(i / C1) * WS + i % C1] = -a.coefA[i];
    for (int i = threadIdx.x; i < C1; i += 256) {
        bn1[i] = a.mean1[i]; bn1[C1 + i] = a.invstd1[i]; bn1[2 * C1 + i] = a.g1[i]; bn1[3 * C1 + i] = a.b1[i];
        cv[i] = a.coefc[i];
    }
    // dW2 (sparse part): this lane's channel and input-channel chunk
    const int q4n = C2 / 4;                       // channels per wave
    const int wj = lane % q4n, wc = lane / q4n;   // (256 / C2 chunks of WR channels: q4n * (256 / C2) = 64 lanes)
    const int wc2 = wave * q4n + wj, wc1 = wc * WR;
    float accw[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) accw[i] = 0.f;
    sat_f32x16 accS[NIT][NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i)
#pragma unroll
        for (int j = 0; j < NIT; ++j) accS[i][j] = sat_f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float xsum[NIT * 4];                          // sums of x1 over this thread's positions, channels threadIdx.x / 32 + 8 j
#pragma unroll
    for (int j = 0; j < NIT * 4; ++j) xsum[j] = 0.f;

    float4 pre[NIT * 4];
    auto fetch = [&](int tile_) {
        const long q0 = (long)tile_ * SAT_TP;
#pragma unroll
        for (int j = 0; j < NIT * 4; ++j) {
            const int i = threadIdx.x + j * 256;
            const int k = i / (SAT_TP / 4), c4 = i - k * (SAT_TP / 4);
            const long p = q0 + c4 * 4;
            pre[j] = ((FULL || k < C1) && tile_ < ntiles && p < P) ? *reinterpret_cast<const float4 *>(a.y1 + ((size_t)b * C1 + k) * P + p)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float pv[2] = {0.f, 0.f};
    int pk[2] = {0, 0};
    auto fetch_v = [&](int tile_) {
        const int m = tile_ * 4 + wave;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c2 = lane + 64 * h;
            const bool ok = c2 < C2 && m < M;
            pv[h] = ok ? a.vT[((size_t)b * M + m) * C2 + c2] : 0.f;
            pk[h] = ok ? (a.argT[((size_t)b * M + m) * C2 + c2] & 31) : 0;
        }
    };
    fetch(blockIdx.x * SAT_TILES);
    fetch_v(blockIdx.x * SAT_TILES);

    for (int tt = 0; tt < SAT_TILES; ++tt) {
        const int tile = blockIdx.x * SAT_TILES + tt;
        if (tile >= ntiles) break;  // workgroup-uniform
        const long p0 = (long)tile * SAT_TP;
        __syncthreads();  // previous tile consumed (and the constant images written, first time round)
#pragma unroll
        for (int j = 0; j < NIT * 4; ++j) {
            const int i = threadIdx.x + j * 256;
            const int k = i / (SAT_TP / 4), c4 = i - k * (SAT_TP / 4);
            if (FULL || k < C1) {
                float4 v = pre[j];
                if (p0 + c4 * 4 < P) {  // P is a multiple of 32: the four elements are in range together
                    const float mu = bn1[k], is = bn1[C1 + k], g = bn1[2 * C1 + k], bt = bn1[3 * C1 + k];
                    v.x = fmaxf(sat_bn(v.x, mu, is, g, bt), 0.f); v.y = fmaxf(sat_bn(v.y, mu, is, g, bt), 0.f);
                    v.z = fmaxf(sat_bn(v.z, mu, is, g, bt), 0.f); v.w = fmaxf(sat_bn(v.w, mu, is, g, bt), 0.f);
                }
                float *d = xs + k * SAT_XS + c4 * 4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                xsum[j] += (v.x + v.y) + (v.z + v.w);
            }
        }
        // v and the routed neighbour of this wave's centroid, one record per channel (fetched one tile ahead like y1)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (lane + 64 * h < C2) va[wave * C2 + lane + 64 * h] = make_float2(pv[h], __int_as_float(pk[h]));
        if (tt + 1 < SAT_TILES) { fetch(tile + 1); fetch_v(tile + 1); }  // in flight during the products below
        __syncthreads();

        // ---- dx1[c1'][position] = W2^T (Dq q) - A x1: A operand [i = c1'][k], B operand [k][j = position] ---------------------
        sat_f32x16 accd[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) accd[it] = sat_f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        constexpr int UB = NIT == 1 ? 8 : 4;  // steps per batch: their LDS reads are issued ahead of the products
        for (int k0 = 0; k0 < C1; k0 += 2 * UB) {
            float bv[UB], av[NIT][UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int k = k0 + 2 * u + kh;
                const int kc = (FULL || k < C1) ? k : 0;  // (clamped index + select: the read itself is unconditional)
                const float xb = xs[kc * SAT_XS + wave * 32 + pl];
                bv[u] = (FULL || k < C1) ? xb : 0.f;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const bool in = FULL || (k < C1 && it * 32 + pl < C1);
                    const float t = as[kc * WS + (in ? it * 32 + pl : 0)];
                    av[it][u] = in ? t : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int it = 0; it < NIT; ++it) accd[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[it][u], bv[u], accd[it], 0, 0, 0);
        }
        for (int k0 = 0; k0 < C2; k0 += 2 * UB) {
            float bv[UB], av[NIT][UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int k = k0 + 2 * u + kh;
                // the sparse Dq q tile is never stored: channel k of this wave's centroid has ONE entry, at its routed neighbour
                const int kc = (FULL || k < C2) ? k : 0;
                const float2 r = va[wave * C2 + kc];
                bv[u] = ((FULL || k < C2) && __float_as_int(r.y) == pl) ? r.x : 0.f;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const bool in = FULL || (k < C2 && it * 32 + pl < C1);
                    const float t = ws[kc * WS + (in ? it * 32 + pl : 0)];
                    av[it][u] = in ? t : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int it = 0; it < NIT; ++it) accd[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[it][u], bv[u], accd[it], 0, 0, 0);
        }
        // ---- Gram matrix S[c1][c1'] += sum over this wave's 32 positions x1[c1][p] x1[c1'][p] --------------------------------
#pragma unroll NIT == 1 ? 16 : 4
        for (int s = 0; s < 32; s += 2) {  // (S is symmetric: the blocks j >= i, mirrored when the partial is written)
            const int pp = wave * 32 + s + kh;
            float xv[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const bool in = FULL || it * 32 + pl < C1;
                const float t = xs[(in ? it * 32 + pl : 0) * SAT_XS + pp];
                xv[it] = in ? t : 0.f;
            }
#pragma unroll
            for (int i = 0; i < NIT; ++i)
#pragma unroll
                for (int j = i; j < NIT; ++j) accS[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[i], xv[j], accS[i][j], 0, 0, 0);
        }
        // ---- dW2 (sparse part): dW2[c2][c1] += v[c2, centroid] x1[c1][its routed neighbour], four centroids of the tile ------
#pragma unroll
        for (int ce = 0; ce < 4; ++ce) {
            const float2 r = va[ce * C2 + wc2];
            const float v = r.x;
            const int pos = ce * 32 + __float_as_int(r.y);
#pragma unroll
            for (int i = 0; i < WR; ++i) accw[i] = __fmaf_rn(v, xs[(wc1 + i) * SAT_XS + pos], accw[i]);
        }
        // ---- dx1 = (sparse - A x1) - c -----------------------------------------------------------------------------------------
        const long p = p0 + wave * 32 + pl;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (p < P && a.dx1_pm) {
                float4 *row4 = reinterpret_cast<float4 *>(a.dx1 + ((size_t)b * P + p) * C1 + it * 32 + 4 * kh);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    if (FULL || it * 32 + 8 * q4 + 4 * kh < C1) {
                        const float *cc = cv + it * 32 + 8 * q4 + 4 * kh;
                        row4[2 * q4] = make_float4(__fsub_rn(accd[it][4 * q4], cc[0]), __fsub_rn(accd[it][4 * q4 + 1], cc[1]),
                                                   __fsub_rn(accd[it][4 * q4 + 2], cc[2]), __fsub_rn(accd[it][4 * q4 + 3], cc[3]));
                    }
            } else if (p < P) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = it * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (FULL || row < C1) a.dx1[((size_t)b * C1 + row) * P + p] = __fsub_rn(accd[it][r], cv[row]);
                }
            }
        }
    }

    // ---- this workgroup's partials: [C2*C1 sparse dW2 | C1*C1 Gram | C1 sums] ---------------------------------------------------
    const int part = blockIdx.y * gridDim.x + blockIdx.x;
    float *out = a.partial + (size_t)part * ((size_t)C2 * C1 + (size_t)C1 * C1 + C1);
#pragma unroll
    for (int i = 0; i < WR; ++i) out[(size_t)wc2 * C1 + wc1 + i] = accw[i];  // every entry has one owner in the workgroup
    __syncthreads();
    float *red = sat_smem;  // [C1][C1]
    for (int i = threadIdx.x; i < C1 * C1; i += 256) red[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NIT; ++i)
#pragma unroll
                for (int j = i; j < NIT; ++j) {
                    const int cj = j * 32 + pl;
                    if (cj < C1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ci = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                            if (ci < C1) {
                                red[ci * C1 + cj] += accS[i][j][r];
                                if (j > i) red[cj * C1 + ci] += accS[i][j][r];  // the mirrored block (distinct addresses per lane)
                            }
                        }
                    }
                }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < C1 * C1; i += 256) out[(size_t)C2 * C1 + i] = red[i];
    // sums of x1: thread t holds channels t / 32 + 8 j over the positions c4 = t % 32 (+ tiles): fold the 32 threads of a half wave
#pragma unroll
    for (int j = 0; j < NIT * 4; ++j) {
        float v = xsum[j];
        for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        const int k = threadIdx.x / 32 + 8 * j;
        if ((threadIdx.x & 31) == 0 && k < C1) out[(size_t)C2 * C1 + (size_t)C1 * C1 + k] = v;
    }
}

// the partials of every workgroup summed in a fixed order, in double: red[total].  A workgroup owns 16 elements; its 64 slices
// each sum every 64th partial (two interleaved sub-sums), then the slices are added in order.
__global__ __launch_bounds__(1024) void sat_alg_reduce_kernel(int total, int nparts, const float *__restrict__ partial, double *__restrict__ red)
{
    __shared__ double sm[64][16];
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;
    double s0 = 0.0, s1 = 0.0;
    if (i < total) {
        int k = sl;
        for (; k + 64 < nparts; k += 128) { s0 += (double)partial[(size_t)k * total + i]; s1 += (double)partial[(size_t)(k + 64) * total + i]; }
        if (k < nparts) s0 += (double)partial[(size_t)k * total + i];
    }
    sm[sl][e] = s0 + s1;
    __syncthreads();
    if (sl == 0 && i < total) {
        double t = sm[0][e];
        for (int k = 1; k < 64; ++k) t += sm[k][e];
        red[i] = t;
    }
}

// dW2 = sparse part - E W2 S - t s^T
__global__ __launch_bounds__(256) void sat_alg_dw_kernel(int C1, int C2, const double *__restrict__ red, const float *__restrict__ w2,
                                                         const double *__restrict__ coefE, const double *__restrict__ coeft,
                                                         float *__restrict__ dw2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C2 * C1) return;
    const int c2 = i / C1, c1 = i - c2 * C1;
    const double *S = red + (size_t)C2 * C1, *sx = S + (size_t)C1 * C1;
    double ws = 0.0;
    for (int k = 0; k < C1; ++k) ws += (double)w2[c2 * C1 + k] * S[(size_t)k * C1 + c1];
    dw2[i] = (float)(red[i] - coefE[c2] * ws - coeft[c2] * sx[c1]);
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

static bool sat_alg_supported(int C1, int C2)
{
    static const bool off = getenv("AMC3D_SAT_RECOMPUTE_BACKWARD") != nullptr;
    return !off && C1 % 4 == 0 && C2 % 4 == 0 && 256 % C2 == 0 && C2 <= 128 && ((long)C1 * C2) % 256 == 0 && C1 % (256 / C2) == 0 && (C1 * C2) / 256 <= 32;
}

static size_t sat_al(size_t n) { return (n + 255) & ~(size_t)255; }

struct SatAlgLayout { size_t partial, pg, vT, argT, coefA, coefc, coefE, coeft, red, total; int nblk; size_t per; };
static SatAlgLayout sat_alg_layout(int B, int C1, int C2, int M)
{
    SatAlgLayout l{};
    const size_t parts = (size_t)B * div_up(div_up(M, 4), SAT_TILES);
    l.per = (size_t)C2 * C1 + (size_t)C1 * C1 + C1;
    l.nblk = B * div_up(M, 64);
    size_t o = 0;
    l.partial = o; o += sat_al(parts * l.per * sizeof(float));
    l.pg = o; o += sat_al((size_t)C2 * l.nblk * 2 * sizeof(double));
    l.vT = o; o += sat_al((size_t)B * M * C2 * sizeof(float));
    l.argT = o; o += sat_al((size_t)B * M * C2);
    l.coefA = o; o += sat_al((size_t)C1 * C1 * sizeof(float));
    l.coefc = o; o += sat_al((size_t)C1 * sizeof(float));
    l.coefE = o; o += sat_al((size_t)C2 * sizeof(double));
    l.coeft = o; o += sat_al((size_t)C2 * sizeof(double));
    l.red = o; o += sat_al(l.per * sizeof(double));
    l.total = o;
    return l;
}

// bytes for the statistics partials (forward and backward) and, for backward, the weight-gradient partials
AMC_API size_t amc3d_sa_tail_workspace_bytes(int B, int C1, int C2, int M)
{
    if (B <= 0 || M <= 0) return 0;
    const size_t parts = (size_t)B * div_up(div_up(M, 4), SAT_TILES);
    const size_t shared = parts * C2 * 2 * sizeof(double);                                   // statistics partials
    const size_t fwd = (size_t)B * C2 * M * sizeof(float);                                   // the raw extremes (when not handed out)
    const size_t bwd = parts * (size_t)C2 * C1 * sizeof(float) + 4 * (size_t)C2 * sizeof(float);  // dW2 partials, means
    size_t need = shared + (fwd > bwd ? fwd : bwd) + 64;
    const size_t alg = sat_alg_layout(B, C1, C2, M).total + 256;
    return need > alg ? need : alg;
}

// pooled (B,C2,M) = max_k [relu2](bn2(W2 . relu(bn1(y1)))) with batch statistics for BN2 (returned in
// mean2, invstd2, var_unbiased2; running buffers updated when given, momentum < 0: not here).
// zext_out (B,C2,M) fp32 / arg_out (B,C2,M) bytes, both or neither: the raw extreme of z over the 32 neighbours that the pool
// selects and the first neighbour attaining it -- with them amc3d_sa_tail_backward needs no recomputation of z
AMC_API int amc3d_sa_tail_forward(int B, int C1, int C2, int M, int K, const float *y1, const float *mean1,
                                  const float *invstd1, const float *gamma1, const float *beta1, const float *w2,
                                  const float *gamma2, const float *beta2, float eps2, float momentum2, int relu2,
                                  float *pooled, float *mean2, float *invstd2, float *var_unbiased2,
                                  float *running_mean2, float *running_var2, long long *tracked2, float *zext_out,
                                  unsigned char *arg_out, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || M <= 0) return 0;
    if (!sat_supported(C1, C2, K) || !y1 || !mean1 || !invstd1 || !gamma1 || !beta1 || !w2 || !gamma2 || !beta2 || !pooled ||
        !mean2 || !invstd2 || !var_unbiased2 || !workspace || !sat_aligned16(y1) || (!zext_out) != (!arg_out) ||
        workspace_bytes < amc3d_sa_tail_workspace_bytes(B, C1, C2, M))
        return bad_arg("amc3d_sa_tail_forward: unsupported shape, null pointer or workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const int groups = div_up(div_up(M, 4), SAT_TILES);
    SatArgs a{};
    a.B = B; a.C1 = C1; a.C2 = C2; a.M = M; a.y1 = y1; a.mean1 = mean1; a.invstd1 = invstd1; a.g1 = gamma1; a.b1 = beta1;
    a.w2 = w2; a.g2 = gamma2; a.b2 = beta2; a.relu2 = relu2; a.partial = (double *)workspace;
    a.zext = zext_out ? zext_out : (float *)((char *)workspace + (size_t)groups * B * C2 * 2 * sizeof(double));
    a.arg_ext = arg_out;
    sat_launch<0>(a, groups, stream);
    hipLaunchKernelGGL(sat_finalize_kernel, dim3(C2, (unsigned)((long)B * M >= 16384 ? 8 : 1)), dim3(256), 0, stream, C2, groups * B,
                       (double)B * (double)M * 32.0, eps2, momentum2, 0, (const double *)workspace, mean2, invstd2,
                       var_unbiased2, (float *)nullptr, running_mean2, running_var2, tracked2, B, M, relu2, gamma2, beta2,
                       (const float *)a.zext, pooled);
    return launch_status("amc3d_sa_tail_forward");
}

namespace amc {
template <int NIT>
static int sat_alg_launch(const SatAlgArgs &a, int groups, size_t lds, hipStream_t stream)
{
    const int wr = a.C1 * a.C2 / 256;
    const bool full = a.C1 == 32 * NIT && a.C2 % 16 == 0;
#define AMC_SATA2(W, F)                                                                                                          \
    (void)hipFuncSetAttribute((const void *)sat_bwd_alg_kernel<NIT, W, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((sat_bwd_alg_kernel<NIT, W, F>), dim3(groups, a.B), dim3(256), lds, stream, a)
#define AMC_SATA(W)                  \
    if (full) { AMC_SATA2(W, true); } \
    else { AMC_SATA2(W, false); }
    switch (wr) {
        case 1: AMC_SATA(1); break; case 2: AMC_SATA(2); break; case 4: AMC_SATA(4); break; case 8: AMC_SATA(8); break;
        case 16: AMC_SATA(16); break; case 32: AMC_SATA(32); break;
        default: return bad_arg("amc3d_sa_tail_backward: unsupported C1 * C2");
    }
#undef AMC_SATA
#undef AMC_SATA2
    return 0;
}
}  // namespace amc

// dx1 (B,C1,M,32): gradient w.r.t. relu(bn1(y1)) -- BN1 / ReLU backward is the caller's (amc3d_bn_backward on y1);
// dw2 (C2,C1), dgamma2, dbeta2 (C2)
AMC_API int amc3d_sa_tail_backward(int B, int C1, int C2, int M, int K, const float *y1, const float *mean1,
                                   const float *invstd1, const float *gamma1, const float *beta1, const float *w2,
                                   const float *mean2, const float *invstd2, const float *gamma2, const float *beta2,
                                   int relu2, const float *dpooled, const float *zext, const unsigned char *arg_ext,
                                   float *dx1, int dx1_position_major, float *dw2,
                                   float *dgamma2, float *dbeta2, unsigned char *arg_out, void *workspace,
                                   size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || M <= 0) return 0;
    if ((!zext) != (!arg_ext)) return bad_arg("amc3d_sa_tail_backward: zext and arg_ext come together");
    if (dx1_position_major && ((C1 & 3) || !sat_aligned16(dx1)))
        return bad_arg("amc3d_sa_tail_backward: position-major dx1 needs C1 % 4 == 0 and a 16-byte aligned buffer");
    if (!sat_supported(C1, C2, K) || !y1 || !mean1 || !invstd1 || !gamma1 || !beta1 || !w2 || !mean2 || !invstd2 || !gamma2 ||
        !beta2 || !dpooled || !dx1 || !dw2 || !dgamma2 || !dbeta2 || !workspace || !sat_aligned16(y1) ||
        workspace_bytes < amc3d_sa_tail_workspace_bytes(B, C1, C2, M))
        return bad_arg("amc3d_sa_tail_backward: unsupported shape, null pointer or workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const int groups = div_up(div_up(M, 4), SAT_TILES);
    const size_t parts = (size_t)groups * B;
    if (zext && sat_alg_supported(C1, C2) && (long)B * M * C2 < (1L << 31)) {  // the algebraic form: no recomputation of z
        const SatAlgLayout l = sat_alg_layout(B, C1, C2, M);
        char *w = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
        float *vT = (float *)(w + l.vT);
        unsigned char *argT = (unsigned char *)(w + l.argT);
        double *pg = (double *)(w + l.pg);
        const size_t lds_pg = (size_t)64 * (C2 + 1) * sizeof(float) + (size_t)64 * C2;
        hipLaunchKernelGGL(sat_pool_grad_kernel, dim3(div_up(M, 64), B), dim3(256), lds_pg, stream, C2, M, relu2, dpooled, zext, arg_ext,
                           mean2, invstd2, gamma2, beta2, vT, argT, pg);
        hipLaunchKernelGGL(sat_bwd_chan_kernel, dim3(C2), dim3(64), 0, stream, l.nblk, (double)B * (double)M * 32.0, (const double *)pg,
                           mean2, invstd2, gamma2, dbeta2, dgamma2, (double *)(w + l.coefE), (double *)(w + l.coeft));
        hipLaunchKernelGGL(sat_bwd_coef_kernel, dim3(C1), dim3(64), 0, stream, C1, C2, w2, (const double *)(w + l.coefE),
                           (const double *)(w + l.coeft), (float *)(w + l.coefA), (float *)(w + l.coefc));
        SatAlgArgs g{};
        g.B = B; g.C1 = C1; g.C2 = C2; g.M = M; g.y1 = y1; g.mean1 = mean1; g.invstd1 = invstd1; g.g1 = gamma1; g.b1 = beta1;
        g.w2 = w2; g.coefA = (const float *)(w + l.coefA); g.coefc = (const float *)(w + l.coefc); g.vT = vT; g.argT = argT;
        g.dx1 = dx1; g.dx1_pm = dx1_position_major ? 1 : 0; g.partial = (float *)(w + l.partial);
        const int WSd = C1 + 1;
        const size_t lds = ((size_t)C1 * SAT_XS + (size_t)C1 * WSd + (size_t)C2 * WSd + 5 * (size_t)C1 + 2 + 8 * (size_t)C2) * sizeof(float) + 16;
        if (int st = C1 > 32 ? sat_alg_launch<2>(g, groups, lds, stream) : sat_alg_launch<1>(g, groups, lds, stream)) return st;
        hipLaunchKernelGGL(sat_alg_reduce_kernel, dim3(div_up((long)l.per, 16)), dim3(1024), 0, stream, (int)l.per, (int)parts,
                           (const float *)(w + l.partial), (double *)(w + l.red));
        hipLaunchKernelGGL(sat_alg_dw_kernel, dim3(div_up(C2 * C1, 256)), dim3(256), 0, stream, C1, C2, (const double *)(w + l.red), w2,
                           (const double *)(w + l.coefE), (const double *)(w + l.coeft), dw2);
        if (arg_out) {
            const hipError_t e = hipMemcpyAsync(arg_out, arg_ext, (size_t)B * C2 * M, hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) { set_error("amc3d_sa_tail_backward: %s", hipGetErrorString(e)); return (int)e; }
        }
        return launch_status("amc3d_sa_tail_backward");
    }
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
                       (const float *)nullptr, (const float *)nullptr, (float *)nullptr);
    a.mean_dq = means; a.mean_dqx = means + C2; a.dx1 = dx1; a.dx1_pm = dx1_position_major ? 1 : 0; a.partial_w = partial_w;
    sat_launch<3>(a, groups, stream);
    if (int st = launch_status("amc3d_sa_tail_backward")) return st;
    return reduce_partials(C2 * C1, (int)parts, partial_w, dw2, stream);
}
