// Adaptive-margin contrastive loss kernels for gfx950.
//
// The reference evaluates this part of the path as a few dozen torch ops per stage plus a Python
// loop (AMContrast3D/MarginContrast.py:220-259, AEF/ambiguity.py:11-93, AEF/utils.py:11-43),
// materialising (m,23,ncls) and (m,23,C) neighbour tensors.  Here it is five kernels per stage
// that read the stage's xyz / labels / embeddings through the k-NN index and write per-point
// scalars; nothing of size m*k*C is ever stored.
//
// Arithmetic follows the reference's CPU evaluation where it is sensitive:
//  * squared distances for d+/d- use the expanded form of AEF/function.py:18-39,
//    ((-2*(x x' + y y' + z z')) + |p|^2) + |p'|^2 in fp32 without FMA and in that order --
//    the form cancels catastrophically for close points (relative error up to 1e-1), and the
//    ambiguity a_i is a steep function of it, so a "more accurate" distance would NOT match;
//  * cosine similarity is sum_c (f_i[c]/max(|f_i|,eps)) * (f_j[c]/max(|f_j|,eps)), the form
//    torch 2.x's F.cosine_similarity evaluates (MarginContrast.py:77-79).
#include "common.h"

#ifndef AMC_CONTRAST_DIAG
#define AMC_CONTRAST_DIAG 0  // 1 / 2: timing-only builds of contrast_backward_kernel (scratch/contrast_diag.sh), never shipped
#endif

namespace amc {

// ---------------------------------------------------------------------------------------------
// Sub-sampled stage labels: arg-max of the mean one-hot label of the kr nearest full-resolution
// points (AEF/utils.py:29-41) followed by the arg-max of MarginContrast.py:111-113 -- i.e. the
// majority class, lowest class id on equal counts.  One wavefront per point; lanes hold the
// neighbours' classes, one __ballot per class counts them.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vote_labels_kernel(int m, int kr, int ncls, const int *__restrict__ labels0,
                                                          const int *__restrict__ nbr, int *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= m) return;
    int c0 = -1, c1 = -1;  // kr <= 128
    if (lane < kr) c0 = labels0[nbr[(size_t)q * kr + lane]];
    if (lane + 64 < kr) c1 = labels0[nbr[(size_t)q * kr + lane + 64]];
    int best = -1, bestc = 0;
    for (int c = 0; c < ncls; ++c) {
        const int cnt = (int)__popcll(__ballot(c0 == c)) + (int)__popcll(__ballot(c1 == c));
        if (cnt > best) { best = cnt; bestc = c; }
    }
    if (lane == 0) out[q] = bestc;
}

// ---------------------------------------------------------------------------------------------
// Neighbourhood statistics of every point (MarginContrast.py:228-231, ambiguity.py:12,18-21,26-39):
// n+ = #neighbours of the same class, d+/d- = summed "squared distances" to the same-/other-class
// neighbours, and the global max of n+ (ambiguity.py:13 divides by it).
// nbr points at the first kept column of the (m, nbr_stride) k-NN index (the self match in
// column 0 is skipped by the caller's pointer offset, MarginContrast.py:225-226).
// ---------------------------------------------------------------------------------------------
// posmask[i,j] = (class of neighbour j) == (class of i)   (MarginContrast.py:111-115, 228-230)
__global__ void posmask_kernel(int m, int k, int nbr_stride, const int *__restrict__ labels,
                               const int *__restrict__ nbr, unsigned char *__restrict__ posmask)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)m * k) return;
    const int i = (int)(t / k), j = (int)(t - (size_t)i * k);
    posmask[t] = labels[nbr[(size_t)i * nbr_stride + j]] == labels[i] ? 1 : 0;
}

// mode: 1 = constant distance 5 (cctype Method1), 2 = "squared distance" (Method2), 3 = its root (Method3)
__global__ __launch_bounds__(256) void ambiguity_stats_kernel(int m, int k, int nbr_stride, int mode,
                                                              const float *__restrict__ p,
                                                              const unsigned char *__restrict__ posmask,
                                                              const int *__restrict__ nbr, int *__restrict__ n_pos,
                                                              float *__restrict__ d_pos, float *__restrict__ d_neg,
                                                              int *__restrict__ max_npos)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int np = 0;
    if (i < m) {
        const float px = p[(size_t)i * 3], py = p[(size_t)i * 3 + 1], pz = p[(size_t)i * 3 + 2];
        const float ss = __fadd_rn(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)), __fmul_rn(pz, pz));
        float dp = 0.f, dn = 0.f;
        for (int j = 0; j < k; ++j) {
            const int nb = nbr[(size_t)i * nbr_stride + j];
            const float qx = p[(size_t)nb * 3], qy = p[(size_t)nb * 3 + 1], qz = p[(size_t)nb * 3 + 2];
            const float dot = __fadd_rn(__fadd_rn(__fmul_rn(px, qx), __fmul_rn(py, qy)), __fmul_rn(pz, qz));
            const float sd = __fadd_rn(__fadd_rn(__fmul_rn(qx, qx), __fmul_rn(qy, qy)), __fmul_rn(qz, qz));
            float dd = __fadd_rn(__fadd_rn(__fmul_rn(-2.f, dot), ss), sd);
            if (mode == 3) dd = sqrtf(__fadd_rn(fabsf(dd), 1e-12f));  // ambiguity.py:49
            const bool pos = posmask[(size_t)i * k + j] != 0;
            np += pos ? 1 : 0;
            dp = __fadd_rn(dp, pos ? dd : 0.f);
            dn = __fadd_rn(dn, pos ? 0.f : dd);
        }
        n_pos[i] = np;
        d_pos[i] = mode == 1 ? 5.f : dp;  // ambiguity.py:24-25
        d_neg[i] = mode == 1 ? 5.f : dn;
    }
    // wave max, then one atomic per wave
    int w = np;
    for (int s = 32; s >= 1; s >>= 1) w = max(w, __shfl_xor(w, s, 64));
    if ((threadIdx.x & 63) == 0 && w > 0) atomicMax(max_npos, w);
}

// a_i (ambiguity.py:13-14, 56-61, 71): |n+ - max|/max; for 0 < n+ < max:
// 1 / (1 + e^(beta * (n+/d+ - n-/d-)))  with e = fp32(math.e) raised by pow as torch does
__global__ void ambiguity_kernel(int m, int k, float beta, const int *__restrict__ n_pos,
                                 const float *__restrict__ d_pos, const float *__restrict__ d_neg,
                                 const int *__restrict__ max_npos, float *__restrict__ a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int top = *max_npos;
    const int np = n_pos[i];
    float v = __fdiv_rn((float)abs(np - top), (float)top);
    if (0 < np && np < top) {
        const float cc = __fsub_rn(__fdiv_rn((float)np, d_pos[i]), __fdiv_rn((float)(k - np), d_neg[i]));
        v = __fdiv_rn(1.f, __fadd_rn(1.f, powf(2.718281828459045f, __fmul_rn(beta, cc))));
    }
    a[i] = v;
}

// max(||f_i||_2, eps) per row (F.cosine_similarity's clamp_min(eps), eps = 1e-8)
__global__ __launch_bounds__(256) void row_norm_kernel(int m, int C, const float *__restrict__ f,
                                                       float *__restrict__ norm)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = f[(size_t)i * C + c];
        s += v * v;
    }
    for (int sft = 32; sft >= 1; sft >>= 1) s += __shfl_xor(s, sft, 64);
    if (lane == 0) norm[i] = fmaxf(sqrtf(s), 1e-8f);
}

// the same norm and the UNIT rows h_i = f_i / max(||f_i||, eps) (C == 4 * LPR, LPR lanes per row, 16-byte pieces).  The
// contrast kernels below gather h instead of f: the four IEEE divisions per lane and fetched row -- 40 of the ~85 VALU
// instructions a round of rows cost; the kernels were VALU-bound, not gather-bound: with every gather redirected to the
// anchor's own row they took the same time -- are done once per row here, and no neighbour norm is fetched.
template <int LPR>
__global__ __launch_bounds__(256) void row_unit_kernel(int m, const float *__restrict__ f, float *__restrict__ norm,
                                                       float *__restrict__ unit)
{
    const int q = threadIdx.x & (LPR - 1);
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) / LPR;
    if (i >= m) return;  // (whole LPR-lane groups leave together: the shuffles below stay inside a group)
    float4 v = reinterpret_cast<const float4 *>(f)[i * LPR + q];
    float s = v.x * v.x;
    s += v.y * v.y;
    s += v.z * v.z;
    s += v.w * v.w;
#pragma unroll
    for (int d = LPR / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    const float n = fmaxf(sqrtf(s), 1e-8f);
    v.x = __fdiv_rn(v.x, n); v.y = __fdiv_rn(v.y, n); v.z = __fdiv_rn(v.z, n); v.w = __fdiv_rn(v.w, n);
    reinterpret_cast<float4 *>(unit)[i * LPR + q] = v;
    if (q == 0) norm[i] = n;
}

// the same from CHANNEL-major embeddings f_cm (B, C, n), as the decoder leaves them (pointnext_AA.py:518-519 makes the
// point-major copy the loss reads with flatten(transpose)): a 64-point tile goes through LDS, the row-major f is never
// written.  Same arithmetic and summation order as row_unit_kernel.
template <int LPR, int TP>  // TP points per tile (16 at the wide, short stages: a cloud of 375 points is 24 tiles, not 6)
__global__ __launch_bounds__(256) void row_unit_cm_kernel(int n, const float *__restrict__ f_cm, float *__restrict__ norm,
                                                          float *__restrict__ unit)
{
    constexpr int C = 4 * LPR, TS = TP + 1;
    extern __shared__ float s_tile[];  // [C][TP + 1]
    const int b = blockIdx.y, n0 = blockIdx.x * TP;
    for (int e = threadIdx.x; e < C * TP; e += 256) {
        const int ch = e / TP, pt = e - ch * TP;
        s_tile[ch * TS + pt] = n0 + pt < n ? f_cm[((size_t)b * C + ch) * n + n0 + pt] : 0.f;
    }
    __syncthreads();
    const int q = threadIdx.x & (LPR - 1);
    for (int pl = threadIdx.x / LPR; pl < TP; pl += 256 / LPR) {  // (the same trip count for every thread)
        float4 v = make_float4(s_tile[(4 * q) * TS + pl], s_tile[(4 * q + 1) * TS + pl], s_tile[(4 * q + 2) * TS + pl],
                               s_tile[(4 * q + 3) * TS + pl]);
        float s = v.x * v.x;
        s += v.y * v.y;
        s += v.z * v.z;
        s += v.w * v.w;
#pragma unroll
        for (int d = LPR / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        const float nv = fmaxf(sqrtf(s), 1e-8f);
        v.x = __fdiv_rn(v.x, nv); v.y = __fdiv_rn(v.y, nv); v.z = __fdiv_rn(v.z, nv); v.w = __fdiv_rn(v.w, nv);
        if (n0 + pl < n) {
            const size_t i = (size_t)b * n + n0 + pl;
            reinterpret_cast<float4 *>(unit)[i * LPR + q] = v;
            if (q == 0) norm[i] = nv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The anchors that enter the loss, 0 < a <= 1 (MarginContrast.py:250-252), as a compact ascending list:
// sel[0] = count, sel[1..count] = anchor ids, sel[m+1..] = per-256-block counts (scratch).  The list depends on
// coordinates and labels only, so it is built with the stage's plan; the contrast kernels then run one full
// wave per SELECTED anchor instead of testing a_i per anchor (two thirds of the full-resolution points on the synthetic
// rooms, nearly all at the coarse stages).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_count_kernel(int m, const float *__restrict__ a, int *__restrict__ sel)
{
    __shared__ int s_cnt[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool s = i < m && 0.f < a[i] && a[i] <= 1.f;
    const int c = (int)__popcll(__ballot(s));
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) sel[(size_t)m + 1 + blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

__global__ __launch_bounds__(256) void select_write_kernel(int m, const float *__restrict__ a, int *__restrict__ sel)
{
    __shared__ int s_part[4], s_cnt[4];
    const int *bc = sel + (size_t)m + 1;
    int before = 0;  // selected anchors in the blocks in front of this one
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += bc[b];
    for (int s = 32; s >= 1; s >>= 1) before += __shfl_xor(before, s, 64);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool sl = i < m && 0.f < a[i] && a[i] <= 1.f;
    const unsigned long long mask = __ballot(sl);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { s_part[wv] = before; s_cnt[wv] = (int)__popcll(mask); }
    __syncthreads();
    int base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    for (int w = 0; w < wv; ++w) base += s_cnt[w];
    if (sl) sel[1 + base + (int)__popcll(mask & ((1ull << lane) - 1ull))] = i;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        sel[0] = s_part[0] + s_part[1] + s_part[2] + s_part[3] + s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// ---------------------------------------------------------------------------------------------
// Contrast forward, one wave per anchor (C == 4 * LPR) over the UNIT rows: LPR lanes read one row as 16-byte pieces, so
// every wave instruction fetches R = 64/LPR whole rows (full 64-byte sectors), U = 4 such rounds in flight.  The cosine
// of a slot is four products and a tree over the row's LPR lanes (every lane of the row then holds it); the exponential of
// slot (round t, row r) is evaluated ONCE, by lane q = t of row r, not by all LPR lanes of the row in every round.
// Same per-element arithmetic as contrast_forward_kernel below (u_c / n_i * (v_c / n_x), channel order of the tree).
// sel == nullptr: every anchor is visited and tested.
// ---------------------------------------------------------------------------------------------
// four channels of the cosine of two unit rows: one multiply and three fused multiply-adds, the same in the forward and in
// the backward that recomputes it
__device__ __forceinline__ float unit_dot4(const float4 &a, const float4 &b)
{
    return __fmaf_rn(a.w, b.w, __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, __fmul_rn(a.x, b.x))));
}

template <int LPR>
__global__ __launch_bounds__(256) void contrast_forward_unit_kernel(
    int m, int k, int nbr_stride, const float *__restrict__ unit, const int *__restrict__ nbr,
    const unsigned char *__restrict__ posmask, const float *__restrict__ a, const int *__restrict__ sel, float mu, float nu,
    float temperature, float *__restrict__ sim, float *__restrict__ stats, float *__restrict__ loss_pt)
{
    constexpr int R = 64 / LPR;  // rows per round
    constexpr int U = 4;         // rounds in flight (LPR >= U: lane q < U of a row owns the row's slot of round q)
    const int lane = threadIdx.x & 63, q = lane & (LPR - 1), r = lane / LPR;
    const int cnt = sel ? sel[0] : m;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= cnt) return;
    const int i = sel ? sel[1 + w] : w;
    const float ai = a[i];
    if (!(0.f < ai && ai <= 1.f)) {
        if (lane == 0) loss_pt[i] = 0.f;
        return;
    }
    const float margin = __fadd_rn(__fmul_rn(mu, ai), nu);
    const float4 *h4 = reinterpret_cast<const float4 *>(unit);
    const float4 u = h4[(size_t)i * LPR + q];
    float psum = 0.f, tsum = 0.f;
    for (int j0 = 0; j0 < k; j0 += U * R) {
        int nb[U];
        float4 v[U];
#pragma unroll
        for (int t = 0; t < U; ++t) {
            const int j = j0 + t * R + r;
            nb[t] = j < k ? nbr[(size_t)i * nbr_stride + j] : -1;
        }
        const int jq = j0 + q * R + r;  // the slot this lane evaluates
        const bool mine = q < U && jq < k;
        const bool pos = mine ? posmask[(size_t)i * k + jq] != 0 : false;
#pragma unroll
        for (int t = 0; t < U; ++t) v[t] = nb[t] >= 0 ? h4[(unsigned)nb[t] * (unsigned)LPR + (unsigned)q] : make_float4(0.f, 0.f, 0.f, 0.f);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < U; ++t) {
            float acc = unit_dot4(u, v[t]);
#pragma unroll
            for (int d = LPR / 2; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
            s = q == t ? acc : s;
        }
        if (mine) {
            if (sim) sim[(size_t)i * k + jq] = s;
            const float e = expf(__fdiv_rn(pos ? __fsub_rn(s, margin) : s, temperature));
            psum += pos ? e : 0.f;
            tsum += e;
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        psum += __shfl_xor(psum, d, 64);
        tsum += __shfl_xor(tsum, d, 64);
    }
    if (lane == 0) {
        loss_pt[i] = -logf(__fadd_rn(__fdiv_rn(psum, tsum), 1e-12f));
        if (stats) { stats[(size_t)i * 2] = psum; stats[(size_t)i * 2 + 1] = tsum; }  // what the backward's records need of this pass
    }
}

// ---------------------------------------------------------------------------------------------
// Contrast forward (MarginContrast.py:250-257, 117-174 with margin 'adaptive', db '-m', Method1):
// 32 lanes per anchor, lane j owns neighbour j (k <= 32 per round); the anchor row is a broadcast
// load, the neighbour row a per-lane 16-byte stream that stays in L1 across the channel loop.
// Writes the cosine similarities (for the backward) and the per-anchor loss; anchors outside
// 0 < a <= 1 (perfectly consistent neighbourhoods) contribute nothing.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void contrast_forward_kernel(
    int m, int C, int k, int nbr_stride, const float *__restrict__ f, const float *__restrict__ norm,
    const int *__restrict__ nbr, const unsigned char *__restrict__ posmask, const float *__restrict__ a,
    const int *__restrict__ sel, float mu, float nu, float temperature, float *__restrict__ sim,
    float *__restrict__ loss_pt)
{
    const int sub = threadIdx.x & 31;
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (w >= (sel ? sel[0] : m)) return;
    const int i = sel ? sel[1 + w] : w;
    const float ai = a[i];
    if (!(0.f < ai && ai <= 1.f)) {
        if (sub == 0) loss_pt[i] = 0.f;
        return;
    }
    const float ni = norm[i];
    const float margin = __fadd_rn(__fmul_rn(mu, ai), nu);
    const float *fi = f + (size_t)i * C;
    float psum = 0.f, tsum = 0.f;
    for (int j0 = 0; j0 < k; j0 += 32) {
        const int j = j0 + sub;
        float e = 0.f;
        bool pos = false;
        if (j < k) {
            const int nb = nbr[(size_t)i * nbr_stride + j];
            const float nj = norm[nb];
            const float *fj = f + (size_t)nb * C;
            float acc = 0.f;
            if ((C & 3) == 0) {
                for (int c = 0; c < C; c += 4) {
                    const float4 u = *reinterpret_cast<const float4 *>(fi + c);
                    const float4 v = *reinterpret_cast<const float4 *>(fj + c);
                    acc += __fdiv_rn(u.x, ni) * __fdiv_rn(v.x, nj);
                    acc += __fdiv_rn(u.y, ni) * __fdiv_rn(v.y, nj);
                    acc += __fdiv_rn(u.z, ni) * __fdiv_rn(v.z, nj);
                    acc += __fdiv_rn(u.w, ni) * __fdiv_rn(v.w, nj);
                }
            } else {
                for (int c = 0; c < C; ++c) acc += __fdiv_rn(fi[c], ni) * __fdiv_rn(fj[c], nj);
            }
            sim[(size_t)i * k + j] = acc;
            pos = posmask[(size_t)i * k + j] != 0;
            const float s = pos ? __fsub_rn(acc, margin) : acc;
            e = expf(__fdiv_rn(s, temperature));
        }
        psum += pos ? e : 0.f;
        tsum += e;
    }
    for (int s = 16; s >= 1; s >>= 1) {
        psum += __shfl_xor(psum, s, 64);
        tsum += __shfl_xor(tsum, s, 64);
    }
    if (sub == 0) loss_pt[i] = -logf(__fadd_rn(__fdiv_rn(psum, tsum), 1e-12f));
}

// mean of loss_pt over the selected anchors, deterministically (one workgroup, fixed order):
// out[0] = sum / count, out[1] = count.  torch.mean of an empty selection is NaN in the
// reference; so is 0/0 here.
__global__ __launch_bounds__(1024) void masked_mean_kernel(int m, const float *__restrict__ loss_pt,
                                                           const float *__restrict__ a, float *__restrict__ out)
{
    __shared__ double s_sum[16];
    __shared__ int s_cnt[16];
    double sum = 0.0;
    int cnt = 0;
    // one workgroup (fixed order, no workspace): keep many 16-byte loads in flight per thread instead of a
    // dependent scalar load per element
    const int m4 = ((((uintptr_t)loss_pt | (uintptr_t)a) & 15) == 0) ? (m & ~3) : 0;
    for (int i0 = threadIdx.x * 4; i0 < m4; i0 += 4096 * 4) {
        float4 lv[4], av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 4096;
            const bool ok = i < m4;
            lv[u] = ok ? *reinterpret_cast<const float4 *>(loss_pt + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            av[u] = ok ? *reinterpret_cast<const float4 *>(a + i) : make_float4(-1.f, -1.f, -1.f, -1.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float l4[4] = {lv[u].x, lv[u].y, lv[u].z, lv[u].w}, a4[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (0.f < a4[e] && a4[e] <= 1.f) { sum += (double)l4[e]; cnt += 1; }
        }
    }
    for (int i = m4 + threadIdx.x; i < m; i += 1024) {
        const float ai = a[i];
        if (0.f < ai && ai <= 1.f) { sum += (double)loss_pt[i]; cnt += 1; }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        sum += __shfl_xor(sum, s, 64);
        cnt += __shfl_xor(cnt, s, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_cnt[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        int c = 0;
        for (int w = 0; w < 16; ++w) { t += s_sum[w]; c += s_cnt[w]; }
        out[0] = (float)(t / (double)c);
        out[1] = (float)c;
    }
}

// ---------------------------------------------------------------------------------------------
// Contrast backward: d(mean loss)/d f.  LPA lanes per anchor, lane owns channels c, c+LPA, ...;
// neighbours are walked one at a time so every atomic wave-instruction adds LPA contiguous floats
// of one row (the shape global float atomics run at full rate for).
//   l = -log(r + eps), r = P/S, e_j = exp(s'_j / T):  dl/ds_j = -(e_j (pos_j S - P)) / ((r+eps) S^2 T)
//   ds_j/df_i = (fhat_j - s_j fhat_i)/|f_i|,  ds_j/df_j = (fhat_i - s_j fhat_j)/|f_j|
// Where the time goes (scratch/contrast_diag.sh + contrast_bench.py, S3DIS-like batch, 66-99 % of the anchors listed): the
// kernel adds 387 / 229 / 134 / 73 MB of rows at the four stages in 0.33 / 0.20 / 0.12 / 0.06 ms = 1.15-1.17 TB/s, the
// chip-wide float-atomic rate (MI355X_MICROARCH.md: 1.26-1.36 TB/s); without the neighbour-row atomics it takes a third
// of that, without the gathers the same.  Splitting an anchor's neighbours over several lane groups (tried for the deep
// stages) only adds prologues: +15-25 %.
// Memory-level parallelism: lane j of the group first loads everything neighbour j needs that is not a row
// (its index, norm, similarity, mask -- one round trip for all k), the walk then broadcasts those by shuffle and
// keeps U neighbour rows in flight ahead of the atomics.  sel as in the forward.
// ---------------------------------------------------------------------------------------------
template <int LPA, int VPT>
__global__ __launch_bounds__(256) void contrast_backward_kernel(
    int m, int C, int k, int nbr_stride, const float *__restrict__ f, const float *__restrict__ norm,
    const int *__restrict__ nbr, const unsigned char *__restrict__ posmask, const float *__restrict__ a,
    const int *__restrict__ sel, float mu, float nu, float temperature, const float *__restrict__ sim,
    const float *__restrict__ mean_cnt, const float *__restrict__ grad_out, float *__restrict__ grad_f)
{
    constexpr int U = VPT >= 8 ? 2 : (VPT >= 4 ? 4 : 8);  // neighbour rows in flight
    const int sub = threadIdx.x & (LPA - 1);
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) / LPA;
    if (w >= (sel ? sel[0] : m)) return;
    const int i = sel ? sel[1 + w] : w;
    const float ai = a[i];
    if (!(0.f < ai && ai <= 1.f)) return;
    const float scale = grad_out[0] / mean_cnt[1];
    const float ni = norm[i];
    const float margin = __fadd_rn(__fmul_rn(mu, ai), nu);

    float fhi[VPT], gi[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int c = sub + v * LPA;
        fhi[v] = c < C ? __fdiv_rn(f[(size_t)i * C + c], ni) : 0.f;
        gi[v] = 0.f;
    }
    // lane `sub` holds neighbour j0 + sub of the current chunk of LPA neighbours
    int nb_l = -1;
    bool pos_l = false;
    float sj_l = 0.f, nj_l = 1.f, e_l = 0.f;
    auto load_chunk = [&](int j0) {
        const int j = j0 + sub;
        nb_l = -1; pos_l = false; sj_l = 0.f; nj_l = 1.f; e_l = 0.f;
        if (j < k) {
            nb_l = nbr[(size_t)i * nbr_stride + j];
            pos_l = posmask[(size_t)i * k + j] != 0;
            sj_l = sim[(size_t)i * k + j];
            nj_l = norm[nb_l];
            e_l = expf(__fdiv_rn(pos_l ? __fsub_rn(sj_l, margin) : sj_l, temperature));
        }
    };
    // P and S over all neighbours
    float psum = 0.f, tsum = 0.f;
    for (int j0 = (k - 1) / LPA * LPA; j0 >= 0; j0 -= LPA) {  // ends on chunk 0, which the walk starts with
        load_chunk(j0);
        psum += pos_l ? e_l : 0.f;
        tsum += e_l;
    }
#pragma unroll
    for (int s = LPA / 2; s >= 1; s >>= 1) {
        psum += __shfl_xor(psum, s, 64);
        tsum += __shfl_xor(tsum, s, 64);
    }
    const float r = psum / tsum;
    const float coef = -scale / ((r + 1e-12f) * tsum * tsum * temperature);
    if (psum == 0.f) return;  // no positive neighbour: constant loss, zero gradient

    for (int j0 = 0; j0 < k; j0 += LPA) {
        if (j0 > 0) load_chunk(j0);
        const float g_l = coef * e_l * ((pos_l ? tsum : 0.f) - psum);  // dL/ds_j
        const int cnt = min(LPA, k - j0);
        for (int jj = 0; jj < cnt; jj += U) {
            int nb[U];
            float fj[U][VPT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int src = min(jj + u, LPA - 1);
                const int t = __shfl(nb_l, src, LPA);
                nb[u] = jj + u < cnt ? t : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    const int c = sub + v * LPA;
#if AMC_CONTRAST_DIAG == 2  // diagnostic build 2: no neighbour-row gathers (timing only)
                    fj[u][v] = (float)nb[u];
#else
                    fj[u][v] = (nb[u] >= 0 && c < C) ? f[(size_t)nb[u] * C + c] : 0.f;
#endif
                }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int src = min(jj + u, LPA - 1);
                const float g = __shfl(g_l, src, LPA), sj = __shfl(sj_l, src, LPA), nj = __shfl(nj_l, src, LPA);
                if (nb[u] < 0) continue;
                const float gin = g / ni, gjn = g / nj;
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    const int c = sub + v * LPA;
                    if (c < C) {
                        const float fhj = __fdiv_rn(fj[u][v], nj);
                        gi[v] += gin * (fhj - sj * fhi[v]);
#if AMC_CONTRAST_DIAG != 1  // diagnostic build 1: no neighbour-row atomics (timing only)
                        atomicAdd(grad_f + (size_t)nb[u] * C + c, gjn * (fhi[v] - sj * fhj));
#endif
                    }
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int c = sub + v * LPA;
        if (c < C) atomicAdd(grad_f + (size_t)i * C + c, gi[v]);
    }
}

// ---------------------------------------------------------------------------------------------
// Contrast backward as a gather (no float atomics, fixed summation order, every row of grad_f written once).
// With g_ij = dL/ds_ij (0 for anchors that are not selected or have no positive neighbour) both halves of the gradient
// have one form:   dL/df_n = sum over edges (n, x) of  g/|f_n| * (fhat_x - s * fhat_n)
//   own edges       x = neighbour j of n           (g, s) = (g_nj, s_nj)       when n is a selected anchor
//   incoming edges  x = anchor i with nbr[i,j] = n (g, s) = (g_ij, s_ij)       listed in rev (amc3d_contrast_csr)
// contrast_coef_kernel writes g for the selected anchors; contrast_backward_rows_kernel walks, one wave per row n
// (C == 4 * LPR, 64/LPR edges per round, U rounds of rows in flight), both edge sets and stores the row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void contrast_coef_kernel(
    int m, int k, const unsigned char *__restrict__ posmask, const float *__restrict__ a, const int *__restrict__ sel,
    float mu, float nu, float temperature, const float *__restrict__ sim, const float *__restrict__ mean_cnt,
    const float *__restrict__ grad_out, float *__restrict__ gco)
{
    const int sub = threadIdx.x & 31;
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (w >= sel[0]) return;
    const int i = sel[1 + w];
    const float ai = a[i];
    const float scale = grad_out[0] / mean_cnt[1];
    const float margin = __fadd_rn(__fmul_rn(mu, ai), nu);
    float psum = 0.f, tsum = 0.f;
    for (int j = sub; j < k; j += 32) {
        const bool pos = posmask[(size_t)i * k + j] != 0;
        const float sj = sim[(size_t)i * k + j];
        const float e = expf(__fdiv_rn(pos ? __fsub_rn(sj, margin) : sj, temperature));
        psum += pos ? e : 0.f;
        tsum += e;
    }
    for (int s = 16; s >= 1; s >>= 1) {
        psum += __shfl_xor(psum, s, 64);
        tsum += __shfl_xor(tsum, s, 64);
    }
    const float r = psum / tsum;
    const float coef = psum == 0.f ? 0.f : -scale / ((r + 1e-12f) * tsum * tsum * temperature);
    for (int j = sub; j < k; j += 32) {
        const bool pos = posmask[(size_t)i * k + j] != 0;
        const float sj = sim[(size_t)i * k + j];
        const float e = expf(__fdiv_rn(pos ? __fsub_rn(sj, margin) : sj, temperature));
        gco[(size_t)i * k + j] = psum == 0.f ? 0.f : coef * e * ((pos ? tsum : 0.f) - psum);
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void contrast_backward_rows_kernel(
    int m, int k, int nbr_stride, const float *__restrict__ f, const float *__restrict__ norm,
    const int *__restrict__ nbr, const float *__restrict__ a, const int *__restrict__ rev,
    const float *__restrict__ sim, const float *__restrict__ gco, float *__restrict__ grad_f)
{
    constexpr int R = 64 / LPR;  // edges per round
    constexpr int U = 4;         // rounds in flight
    const int lane = threadIdx.x & 63, q = lane & (LPR - 1), r = lane / LPR;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= m) return;
    const float an = a[n];
    const int own = (0.f < an && an <= 1.f) ? k : 0;
    const int e0 = rev[n], deg = rev[n + 1] - e0;
    const int *rev_edge = rev + m + 1;
    const int total = own + deg;
    const float4 *f4 = reinterpret_cast<const float4 *>(f);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (total > 0) {
        const float nn = norm[n];
        float4 fn = f4[(size_t)n * LPR + q];
        fn.x = __fdiv_rn(fn.x, nn); fn.y = __fdiv_rn(fn.y, nn); fn.z = __fdiv_rn(fn.z, nn); fn.w = __fdiv_rn(fn.w, nn);
        for (int t0 = 0; t0 < total; t0 += U * R) {
            int x[U];
            float g[U], sv[U], nx[U];
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * R + r;
                x[u] = -1; g[u] = 0.f; sv[u] = 0.f;
                if (t < total) {
                    int pos;
                    if (t < own) { pos = n * k + t; x[u] = nbr[(size_t)n * nbr_stride + t]; }
                    else { pos = rev_edge[e0 + t - own]; x[u] = pos / k; }
                    g[u] = gco[pos];
                    sv[u] = sim[pos];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                nx[u] = x[u] >= 0 ? norm[x[u]] : 1.f;
                v[u] = x[u] >= 0 ? f4[(size_t)x[u] * LPR + q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (x[u] < 0) continue;
                const float gn = g[u] / nn;
                acc.x += gn * (__fdiv_rn(v[u].x, nx[u]) - sv[u] * fn.x);
                acc.y += gn * (__fdiv_rn(v[u].y, nx[u]) - sv[u] * fn.y);
                acc.z += gn * (__fdiv_rn(v[u].z, nx[u]) - sv[u] * fn.z);
                acc.w += gn * (__fdiv_rn(v[u].w, nx[u]) - sv[u] * fn.w);
            }
        }
#pragma unroll
        for (int s = LPR; s < 64; s <<= 1) {
            acc.x += __shfl_xor(acc.x, s, 64);
            acc.y += __shfl_xor(acc.y, s, 64);
            acc.z += __shfl_xor(acc.z, s, 64);
            acc.w += __shfl_xor(acc.w, s, 64);
        }
    }
    if (r == 0) reinterpret_cast<float4 *>(grad_f)[(size_t)n * LPR + q] = acc;
}

// ---------------------------------------------------------------------------------------------
// Contrast backward over the MUTUAL edges (round 3; the default).  91 % of the edges of the stages' 24-NN graphs are mutual
// (x in N(n) and n in N(x)), and everything an edge contributes is symmetric in its two ends: the cosine s_nx = s_xn (the
// products commute and the channel sum runs in the same order, so even the bits agree with what the forward computed at
// the other end) and the positive mask (equal classes).  So point n, walking its OWN list once, has in registers what both
// directions of a mutual edge need:
//     dL/df_n = sum_{x in N(n)} (g_nx [n selected] + g_xn [x selected, edge mutual]) / |f_n| * (fhat_x - s_nx fhat_n)
//             + sum over the non-mutual incoming edges (x -> n), listed in rev (amc3d_contrast_mutual: a tenth of all edges)
// with g_ix = dL/ds_ix = coef_i e_ix ([pos] S_i - P_i), e_ix = exp((s - [pos] margin_i)/T), from a 32-byte record per anchor
// (norm, coef, S, P, margin: contrast_record_kernel).  Every neighbour row is fetched ONCE per (n, x) pair -- the traffic of
// the forward kernel -- every gradient row is written once with a plain store, the summation order is fixed, and no float
// atomic is left (the atomic form added 387 / 229 / 134 / 73 MB of rows per step at the chip's float-atomic rate of 1.15 TB/s).
// ---------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) ContrastRecord { float norm, coef, tsum, psum, margin, pad0, pad1, pad2; };

__global__ __launch_bounds__(256) void contrast_record_kernel(
    int m, int k, const float *__restrict__ norm, const unsigned char *__restrict__ posmask, const float *__restrict__ a,
    float mu, float nu, float temperature, const float *__restrict__ sim, const float *__restrict__ mean_cnt,
    const float *__restrict__ grad_out, ContrastRecord *__restrict__ rec)
{
    const int sub = threadIdx.x & 31;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    if (i >= m) return;
    const float ai = a[i];
    const bool selected = 0.f < ai && ai <= 1.f;
    const float margin = __fadd_rn(__fmul_rn(mu, ai), nu);
    float psum = 0.f, tsum = 0.f;
    if (selected) {  // wave-uniform per half-wave: sim holds values for the selected anchors only
        // the forward's order: slot-local sums over the rounds, then a tree over the slots (contrast_forward_rows_kernel
        // folds 64/LPR slots; any fixed order is within rounding of it)
        for (int j = sub; j < k; j += 32) {
            const bool pos = posmask[(size_t)i * k + j] != 0;
            const float sj = sim[(size_t)i * k + j];
            const float e = expf(__fdiv_rn(pos ? __fsub_rn(sj, margin) : sj, temperature));
            psum += pos ? e : 0.f;
            tsum += e;
        }
    }
    for (int s = 16; s >= 1; s >>= 1) {
        psum += __shfl_xor(psum, s, 64);
        tsum += __shfl_xor(tsum, s, 64);
    }
    if (sub == 0) {
        const float scale = grad_out[0] / mean_cnt[1];
        const float r = psum / tsum;
        ContrastRecord o;
        o.norm = norm[i];
        o.coef = (selected && psum != 0.f) ? -scale / ((r + 1e-12f) * tsum * tsum * temperature) : 0.f;
        o.tsum = tsum; o.psum = psum; o.margin = margin; o.pad0 = o.pad1 = o.pad2 = 0.f;
        rec[i] = o;
    }
}

// (unit rows, see row_unit_kernel: hx = f_x / n_x is fetched, not recomputed; lane q < U of a row decodes and evaluates the
// row's slot of round q -- index, mask, mutual count, the record of x, the two exponentials -- once, and hands x down to /
// the finished coefficient back to the row's lanes by shuffles)
// the same records from the two sums the forward pass kept per anchor (stats[2 i] = sum of the positives' exponentials,
// stats[2 i + 1] = sum of all): nothing to recompute, no similarities to read; one thread per anchor
__global__ __launch_bounds__(256) void contrast_record_stats_kernel(int m, const float *__restrict__ norm, const float *__restrict__ a,
                                                                    float mu, float nu, float temperature,
                                                                    const float *__restrict__ stats, const float *__restrict__ mean_cnt,
                                                                    const float *__restrict__ grad_out, ContrastRecord *__restrict__ rec)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float ai = a[i];
    const bool selected = 0.f < ai && ai <= 1.f;  // (the forward visited the selected anchors only: the others' stats are unwritten)
    const float psum = selected ? stats[(size_t)i * 2] : 0.f, tsum = selected ? stats[(size_t)i * 2 + 1] : 1.f;
    const float scale = grad_out[0] / mean_cnt[1];
    const float r = psum / tsum;
    ContrastRecord o;
    o.norm = norm[i];
    o.coef = (selected && psum != 0.f) ? -scale / ((r + 1e-12f) * tsum * tsum * temperature) : 0.f;
    o.tsum = tsum; o.psum = psum; o.margin = __fadd_rn(__fmul_rn(mu, ai), nu); o.pad0 = o.pad1 = o.pad2 = 0.f;
    rec[i] = o;
}

template <int LPR>
__global__ __launch_bounds__(256) void contrast_backward_mutual_kernel(
    int m, int k, int nbr_stride, const float *__restrict__ unit, const int *__restrict__ nbr,
    const unsigned char *__restrict__ posmask, const unsigned char *__restrict__ mutual, const int *__restrict__ rev,
    const ContrastRecord *__restrict__ rec, float temperature, float *__restrict__ grad_f)
{
    constexpr int R = 64 / LPR;  // rows per round
    constexpr int U = 4;         // rounds in flight
    const int lane = threadIdx.x & 63, q = lane & (LPR - 1), r = lane / LPR, row0 = lane & ~(LPR - 1);
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= m) return;
    const float4 *h4 = reinterpret_cast<const float4 *>(unit);
    const float4 *rec4 = reinterpret_cast<const float4 *>(rec);
    const float4 rn0 = rec4[(size_t)n * 2];       // norm, coef, tsum, psum
    const float margin_n = rec[n].margin;
    const float nn = rn0.x, coef_n = rn0.y, tsum_n = rn0.z, psum_n = rn0.w;
    const float4 fn = h4[(size_t)n * LPR + q];
    const int e0 = rev[n], deg = rev[n + 1] - e0;
    const int *rev_edge = rev + m + 1;
    const int total = k + deg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < total; t0 += U * R) {
        // the slot this lane decodes and evaluates: round q, row r
        const int tq = t0 + q * R + r;
        int myx = -1;
        bool pos = false, own = false;
        float inc = 0.f;  // incoming edges x -> n met at this slot (their number: 0 or 1 in a k-NN graph)
        if (q < U) {
            if (tq < k) {  // own list: the edge n -> x, and x -> n when it is mutual
                myx = nbr[(size_t)n * nbr_stride + tq];
                pos = posmask[(size_t)n * k + tq] != 0;
                inc = (float)(mutual[(size_t)n * k + tq] & 0x7f);
                own = coef_n != 0.f;
                if (!own && inc == 0.f) myx = -1;  // nothing flows along this edge
            } else if (tq < total) {  // a non-mutual incoming edge x -> n
                const int p = rev_edge[e0 + tq - k];
                myx = p / k;
                pos = posmask[p] != 0;
                inc = 1.f;
            }
        }
        int x[U];
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = __shfl(myx, row0 | u, 64);
        const float4 rx = myx >= 0 ? rec4[(size_t)myx * 2] : make_float4(1.f, 0.f, 1.f, 0.f);
        const float mx = myx >= 0 ? rec[myx].margin : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[u] >= 0 ? h4[(unsigned)x[u] * (unsigned)LPR + (unsigned)q] : make_float4(0.f, 0.f, 0.f, 0.f);
        float s[U], smine = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float t = unit_dot4(fn, v[u]);   // the forward's expression and order (contrast_forward_unit_kernel)
#pragma unroll
            for (int d = LPR / 2; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
            s[u] = t;
            smine = q == u ? t : smine;
        }
        float gn = 0.f;
        if (myx >= 0) {
            float g = 0.f;
            if (own) {
                const float e = expf(__fdiv_rn(pos ? __fsub_rn(smine, margin_n) : smine, temperature));
                g += coef_n * e * ((pos ? tsum_n : 0.f) - psum_n);
            }
            if (inc != 0.f && rx.y != 0.f) {
                const float e = expf(__fdiv_rn(pos ? __fsub_rn(smine, mx) : smine, temperature));
                g += inc * (rx.y * e * ((pos ? rx.z : 0.f) - rx.w));
            }
            gn = g / nn;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float gu = __shfl(gn, row0 | u, 64);
            acc.x = __fmaf_rn(gu, __fmaf_rn(-s[u], fn.x, v[u].x), acc.x);
            acc.y = __fmaf_rn(gu, __fmaf_rn(-s[u], fn.y, v[u].y), acc.y);
            acc.z = __fmaf_rn(gu, __fmaf_rn(-s[u], fn.z, v[u].z), acc.z);
            acc.w = __fmaf_rn(gu, __fmaf_rn(-s[u], fn.w, v[u].w), acc.w);
        }
    }
#pragma unroll
    for (int d = LPR; d < 64; d <<= 1) {
        acc.x += __shfl_xor(acc.x, d, 64);
        acc.y += __shfl_xor(acc.y, d, 64);
        acc.z += __shfl_xor(acc.z, d, 64);
        acc.w += __shfl_xor(acc.w, d, 64);
    }
    if (r == 0) reinterpret_cast<float4 *>(grad_f)[(size_t)n * LPR + q] = acc;
}

// ---------------------------------------------------------------------------------------------
// Cross entropy over channel-major logits (B, C, N) with class targets (B, N), mean over the targets
// != ignore_index: nn.CrossEntropyLoss() with its defaults, as loss/build.py:328,338 applies it after a
// transpose + reshape copy of the logits to (B*N, C).  One thread per point walks the classes (coalesced
// along N per class plane); the mean is reduced in two fixed-order stages (deterministic).
// ---------------------------------------------------------------------------------------------
constexpr int CE_THREADS = 256;
__global__ __launch_bounds__(CE_THREADS) void ce_forward_kernel(int C, long N, const float *__restrict__ logits,
                                                                const long long *__restrict__ target, long long ignore,
                                                                float *__restrict__ lse, double *__restrict__ partial)
{
    __shared__ double s_sum[CE_THREADS / 64];
    __shared__ int s_cnt[CE_THREADS / 64];
    const int b = blockIdx.y;
    const long n = (long)blockIdx.x * CE_THREADS + threadIdx.x;
    double term = 0.0;
    int cnt = 0;
    if (n < N) {
        const float *x = logits + (size_t)b * C * N + n;
        float mx = -__builtin_inff();
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, x[(size_t)c * N]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(x[(size_t)c * N] - mx);
        const float l = mx + __logf(se);
        lse[(size_t)b * N + n] = l;
        const long long t = target[(size_t)b * N + n];
        if (t != ignore && t >= 0 && t < C) { term = (double)(l - x[(size_t)t * N]); cnt = 1; }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        term += __shfl_xor(term, s, 64);
        cnt += __shfl_xor(cnt, s, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = term; s_cnt[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        int c = 0;
        for (int w = 0; w < CE_THREADS / 64; ++w) { t += s_sum[w]; c += s_cnt[w]; }
        const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[slot * 2] = t;
        partial[slot * 2 + 1] = (double)c;
    }
}

// out[0] = mean loss, out[1] = number of counted targets
__global__ __launch_bounds__(1024) void ce_finalize_kernel(int nparts, const double *__restrict__ partial,
                                                           float *__restrict__ out)
{
    __shared__ double s_sum[16], s_cnt[16];
    double t = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 1024) { t += partial[(size_t)i * 2]; c += partial[(size_t)i * 2 + 1]; }
    for (int s = 32; s >= 1; s >>= 1) {
        t += __shfl_xor(t, s, 64);
        c += __shfl_xor(c, s, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = t; s_cnt[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tt = 0.0, cc = 0.0;
        for (int w = 0; w < 16; ++w) { tt += s_sum[w]; cc += s_cnt[w]; }
        out[0] = (float)(tt / cc);  // 0/0 = NaN, as torch's mean over no targets
        out[1] = (float)cc;
    }
}

// dlogits[b,c,n] = g * (softmax_c - [c == target]) / count for counted targets, 0 otherwise
__global__ __launch_bounds__(CE_THREADS) void ce_backward_kernel(int C, long N, const float *__restrict__ logits,
                                                                 const long long *__restrict__ target, long long ignore,
                                                                 const float *__restrict__ lse,
                                                                 const float *__restrict__ mean_cnt,
                                                                 const float *__restrict__ grad_out,
                                                                 float *__restrict__ dlogits)
{
    const int b = blockIdx.y;
    const long n = (long)blockIdx.x * CE_THREADS + threadIdx.x;
    if (n >= N) return;
    const float *x = logits + (size_t)b * C * N + n;
    float *d = dlogits + (size_t)b * C * N + n;
    const long long t = target[(size_t)b * N + n];
    const bool counted = t != ignore && t >= 0 && t < C;
    const float scale = counted ? grad_out[0] / mean_cnt[1] : 0.f;
    const float l = lse[(size_t)b * N + n];
    for (int c = 0; c < C; ++c) {
        const float p = __expf(x[(size_t)c * N] - l);
        d[(size_t)c * N] = scale * (p - (c == (int)t ? 1.f : 0.f));
    }
}

}  // namespace amc

using namespace amc;

AMC_API int amc3d_vote_labels(int m, int kr, int num_classes, const int *labels0, const int *nbr_idx, int *labels,
                              void *stream)
{
    if (m <= 0) return 0;
    if (kr <= 0 || kr > 128 || num_classes <= 0 || !labels0 || !nbr_idx || !labels)
        return bad_arg("amc3d_vote_labels: bad argument (kr must be in 1..128)");
    hipLaunchKernelGGL(vote_labels_kernel, dim3(div_up(m, 4)), dim3(256), 0, (hipStream_t)stream, m, kr, num_classes,
                       labels0, nbr_idx, labels);
    return launch_status("amc3d_vote_labels");
}

AMC_API int amc3d_posmask(int m, int k, int nbr_stride, const int *labels, const int *nbr, unsigned char *posmask,
                          void *stream)
{
    if (m <= 0) return 0;
    if (k <= 0 || nbr_stride < k || !labels || !nbr || !posmask) return bad_arg("amc3d_posmask: bad argument");
    hipLaunchKernelGGL(posmask_kernel, dim3(div_up((long)m * k, 256)), dim3(256), 0, (hipStream_t)stream, m, k,
                       nbr_stride, labels, nbr, posmask);
    return launch_status("amc3d_posmask");
}

AMC_API size_t amc3d_ambiguity_workspace_bytes(int m) { return (size_t)m * 12 + 64; }

AMC_API int amc3d_ambiguity(int m, int k, int nbr_stride, int mode, float beta, const float *p,
                            const unsigned char *posmask, const int *nbr, float *a, void *workspace,
                            size_t workspace_bytes, void *stream_)
{
    if (m <= 0) return 0;
    if (k <= 0 || nbr_stride < k || mode < 1 || mode > 3 || !p || !posmask || !nbr || !a || !workspace ||
        workspace_bytes < amc3d_ambiguity_workspace_bytes(m))
        return bad_arg("amc3d_ambiguity: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    int *max_npos = (int *)workspace;  // [0..63] header, then n_pos, d_pos, d_neg
    int *n_pos = (int *)((char *)workspace + 64);
    float *d_pos = (float *)(n_pos + m);
    float *d_neg = d_pos + m;
    if (int st = fill_i32(max_npos, 0, 1, stream)) return st;
    hipLaunchKernelGGL(ambiguity_stats_kernel, dim3(div_up(m, 256)), dim3(256), 0, stream, m, k, nbr_stride, mode, p,
                       posmask, nbr, n_pos, d_pos, d_neg, max_npos);
    hipLaunchKernelGGL(ambiguity_kernel, dim3(div_up(m, 256)), dim3(256), 0, stream, m, k, beta, n_pos, d_pos, d_neg,
                       max_npos, a);
    return launch_status("amc3d_ambiguity");
}

AMC_API size_t amc3d_select_anchors_ints(int m) { return (size_t)(m > 0 ? m : 0) + 1 + (size_t)div_up(m > 0 ? m : 1, 256); }

AMC_API int amc3d_select_anchors(int m, const float *a, int *sel, size_t sel_ints, void *stream_)
{
    if (m < 0 || !sel || sel_ints < amc3d_select_anchors_ints(m) || (m > 0 && !a))
        return bad_arg("amc3d_select_anchors: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    if (m == 0) return fill_i32(sel, 0, 1, stream);
    const int nb = div_up(m, 256);
    hipLaunchKernelGGL(select_count_kernel, dim3(nb), dim3(256), 0, stream, m, a, sel);
    hipLaunchKernelGGL(select_write_kernel, dim3(nb), dim3(256), 0, stream, m, a, sel);
    return launch_status("amc3d_select_anchors");
}

// cm_b > 0: f is channel-major (cm_b, C, m / cm_b) and only the unit-row kernels apply
static int contrast_forward_launch(int cm_b, int m, int C, int k, int nbr_stride, const float *f, const int *nbr,
                                   const unsigned char *posmask, const float *a, const int *sel, float mu, float nu,
                                   float temperature, float *norm, float *unit, float *sim, float *stats, float *loss_pt,
                                   float *mean_cnt, hipStream_t stream)
{
    // (the row kernels index 16-byte pieces with 32 bits: m * C / 4 < 2^32)
    const bool rows = unit && ((((uintptr_t)f) | ((uintptr_t)unit)) & 15) == 0 && amc3d_contrast_backward_csr_supported(C) &&
                      (long)m * C < (1L << 34);
    if (cm_b > 0 && !rows) return bad_arg("amc3d_contrast_forward_cm: C must be 16, 32, 64, 128 or 256; unit required, 16-byte aligned");
    if (!sim && !(rows && stats)) return bad_arg("amc3d_contrast_forward: sim may be NULL only with the unit-row kernels and stats");
#define AMC_FWD(LPR)                                                                                                       \
    do {                                                                                                                   \
        if (cm_b > 0) {                                                                                                    \
            constexpr int TP = LPR >= 32 ? 16 : 64;                                                                        \
            const size_t lds = (size_t)4 * LPR * (TP + 1) * sizeof(float);                                                 \
            hipLaunchKernelGGL((row_unit_cm_kernel<LPR, TP>), dim3(div_up(m / cm_b, TP), cm_b), dim3(256), lds, stream,    \
                               m / cm_b, f, norm, unit);                                                                   \
        } else                                                                                                             \
            hipLaunchKernelGGL((row_unit_kernel<LPR>), dim3(div_up((long)m * LPR, 256)), dim3(256), 0, stream, m, f, norm, unit); \
        hipLaunchKernelGGL((contrast_forward_unit_kernel<LPR>), dim3(div_up(m, 4)), dim3(256), 0, stream, m, k, nbr_stride, \
                           (const float *)unit, nbr, posmask, a, sel, mu, nu, temperature, sim, stats, loss_pt);           \
    } while (0)
    if (rows && C == 16) AMC_FWD(4);
    else if (rows && C == 32) AMC_FWD(8);
    else if (rows && C == 64) AMC_FWD(16);
    else if (rows && C == 128) AMC_FWD(32);
    else if (rows && C == 256) AMC_FWD(64);
    else {
        hipLaunchKernelGGL(row_norm_kernel, dim3(div_up(m, 4)), dim3(256), 0, stream, m, C, f, norm);
        hipLaunchKernelGGL(contrast_forward_kernel, dim3(div_up((long)m * 32, 256)), dim3(256), 0, stream, m, C, k,
                           nbr_stride, f, norm, nbr, posmask, a, sel, mu, nu, temperature, sim, loss_pt);
    }
#undef AMC_FWD
    // anchors the list skips never write loss_pt; masked_mean_kernel reads the selected ones only
    hipLaunchKernelGGL(masked_mean_kernel, dim3(1), dim3(1024), 0, stream, m, loss_pt, a, mean_cnt);
    return launch_status("amc3d_contrast_forward");
}

AMC_API int amc3d_contrast_forward(int m, int C, int k, int nbr_stride, const float *f, const int *nbr,
                                   const unsigned char *posmask, const float *a, const int *sel, float mu, float nu,
                                   float temperature, float *norm, float *unit, float *sim, float *stats, float *loss_pt,
                                   float *mean_cnt, void *stream_)
{
    if (m <= 0) return 0;
    if (C <= 0 || k <= 0 || nbr_stride < k || !f || !nbr || !posmask || !a || !norm || !loss_pt || !mean_cnt)
        return bad_arg("amc3d_contrast_forward: bad argument");
    return contrast_forward_launch(0, m, C, k, nbr_stride, f, nbr, posmask, a, sel, mu, nu, temperature, norm, unit, sim, stats,
                                   loss_pt, mean_cnt, (hipStream_t)stream_);
}

// the same on channel-major embeddings f_cm (b, C, n), m = b * n anchors in cloud-major order (the decoder's layout: no
// point-major copy of f is made; unit receives the point-major unit rows the backward reads)
AMC_API int amc3d_contrast_forward_cm(int b, int C, int n, int k, int nbr_stride, const float *f_cm, const int *nbr,
                                      const unsigned char *posmask, const float *a, const int *sel, float mu, float nu,
                                      float temperature, float *norm, float *unit, float *sim, float *stats, float *loss_pt,
                                      float *mean_cnt, void *stream_)
{
    if (b <= 0 || n <= 0) return 0;
    if ((long)b * n > 0x7fffffffL || C <= 0 || k <= 0 || nbr_stride < k || !f_cm || !nbr || !posmask || !a || !norm || !unit ||
        !loss_pt || !mean_cnt)
        return bad_arg("amc3d_contrast_forward_cm: bad argument");
    return contrast_forward_launch(b, b * n, C, k, nbr_stride, f_cm, nbr, posmask, a, sel, mu, nu, temperature, norm, unit, sim,
                                   stats, loss_pt, mean_cnt, (hipStream_t)stream_);
}

AMC_API int amc3d_contrast_backward(int m, int C, int k, int nbr_stride, const float *f, const float *norm,
                                    const int *nbr, const unsigned char *posmask, const float *a, const int *sel,
                                    float mu, float nu, float temperature, const float *sim, const float *mean_cnt,
                                    const float *grad_out, float *grad_f, void *stream_)
{
    if (m <= 0) return 0;
    if (C <= 0 || C > 512 || k <= 0 || !f || !norm || !nbr || !posmask || !a || !sim || !mean_cnt || !grad_out || !grad_f)
        return bad_arg("amc3d_contrast_backward: bad argument (C must be in 1..512)");
    hipStream_t stream = (hipStream_t)stream_;
#define AMC_BWD(LPA, VPT)                                                                                           \
    hipLaunchKernelGGL((contrast_backward_kernel<LPA, VPT>), dim3(div_up((long)m * LPA, 256)), dim3(256), 0, stream, \
                       m, C, k, nbr_stride, f, norm, nbr, posmask, a, sel, mu, nu, temperature, sim, mean_cnt,       \
                       grad_out, grad_f)
    if (C <= 32) AMC_BWD(32, 1);
    else if (C <= 64) AMC_BWD(64, 1);
    else if (C <= 128) AMC_BWD(64, 2);
    else if (C <= 256) AMC_BWD(64, 4);
    else AMC_BWD(64, 8);
#undef AMC_BWD
    return launch_status("amc3d_contrast_backward");
}

AMC_API int amc3d_contrast_backward_csr_supported(int C)
{
    return C == 16 || C == 32 || C == 64 || C == 128 || C == 256;
}

AMC_API int amc3d_contrast_backward_csr(int m, int C, int k, int nbr_stride, const float *f, const float *norm,
                                        const int *nbr, const unsigned char *posmask, const float *a, const int *sel,
                                        const int *rev, float mu, float nu, float temperature, const float *sim,
                                        const float *mean_cnt, const float *grad_out, float *gco, float *grad_f,
                                        void *stream_)
{
    if (m <= 0) return 0;
    if (!amc3d_contrast_backward_csr_supported(C) || k <= 0 || nbr_stride < k || !f || !norm || !nbr || !posmask || !a ||
        !sel || !rev || !sim || !mean_cnt || !grad_out || !gco || !grad_f || (((uintptr_t)f | (uintptr_t)grad_f) & 15))
        return bad_arg("amc3d_contrast_backward_csr: bad argument (C must be 16, 32, 64, 128 or 256; 16-byte aligned rows)");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(contrast_coef_kernel, dim3(div_up((long)m * 32, 256)), dim3(256), 0, stream, m, k, posmask, a, sel, mu,
                       nu, temperature, sim, mean_cnt, grad_out, gco);
#define AMC_BWD(LPR)                                                                                                  \
    hipLaunchKernelGGL((contrast_backward_rows_kernel<LPR>), dim3(div_up(m, 4)), dim3(256), 0, stream, m, k, nbr_stride, f, \
                       norm, nbr, a, rev, sim, (const float *)gco, grad_f)
    if (C == 16) AMC_BWD(4);
    else if (C == 32) AMC_BWD(8);
    else if (C == 64) AMC_BWD(16);
    else if (C == 128) AMC_BWD(32);
    else AMC_BWD(64);
#undef AMC_BWD
    return launch_status("amc3d_contrast_backward_csr");
}

AMC_API size_t amc3d_contrast_backward_mutual_workspace_bytes(int m) { return (size_t)(m > 0 ? m : 0) * sizeof(ContrastRecord) + 64; }

// grad_f (m,C): every row written once (no zero-initialisation, no atomics).  mutual / rev from amc3d_contrast_mutual on the
// same neighbour lists and ambiguities; unit (the rows f_i / norm_i), sim, norm, mean_cnt as amc3d_contrast_forward left them.
AMC_API int amc3d_contrast_backward_mutual(int m, int C, int k, int nbr_stride, const float *unit, const float *norm,
                                           const int *nbr, const unsigned char *posmask, const float *a,
                                           const unsigned char *mutual, const int *rev, float mu, float nu, float temperature,
                                           const float *sim, const float *stats, const float *mean_cnt, const float *grad_out,
                                           void *workspace, size_t workspace_bytes, float *grad_f, void *stream_)
{
    if (m <= 0) return 0;
    if (!amc3d_contrast_backward_csr_supported(C) || k <= 0 || nbr_stride < k || !unit || !norm || !nbr || !posmask || !a || !mutual ||
        !rev || (!sim && !stats) || !mean_cnt || !grad_out || !grad_f || !workspace ||
        workspace_bytes < amc3d_contrast_backward_mutual_workspace_bytes(m) || (long)m * C >= (1L << 34) ||
        (((uintptr_t)unit | (uintptr_t)grad_f | (uintptr_t)workspace) & 15))
        return bad_arg("amc3d_contrast_backward_mutual: bad argument (C must be 16, 32, 64, 128 or 256; 16-byte aligned rows)");
    hipStream_t stream = (hipStream_t)stream_;
    ContrastRecord *rec = (ContrastRecord *)workspace;
    if (stats)
        hipLaunchKernelGGL(contrast_record_stats_kernel, dim3(div_up(m, 256)), dim3(256), 0, stream, m, norm, a, mu, nu, temperature, stats,
                           mean_cnt, grad_out, rec);
    else
        hipLaunchKernelGGL(contrast_record_kernel, dim3(div_up((long)m * 32, 256)), dim3(256), 0, stream, m, k, norm, posmask, a, mu, nu,
                           temperature, sim, mean_cnt, grad_out, rec);
#define AMC_BWD(LPR)                                                                                                    \
    hipLaunchKernelGGL((contrast_backward_mutual_kernel<LPR>), dim3(div_up(m, 4)), dim3(256), 0, stream, m, k, nbr_stride, unit, \
                       nbr, posmask, mutual, rev, (const ContrastRecord *)rec, temperature, grad_f)
    if (C == 16) AMC_BWD(4);
    else if (C == 32) AMC_BWD(8);
    else if (C == 64) AMC_BWD(16);
    else if (C == 128) AMC_BWD(32);
    else AMC_BWD(64);
#undef AMC_BWD
    return launch_status("amc3d_contrast_backward_mutual");
}

// ---------------------------------------------------------------------------------------------
// Confusion matrix of the training predictions (main_AA.py:414-415: cm.update(logits.argmax(dim=1), target), every
// iteration): arg-max over the class planes of channel-major logits (first maximum, as torch.argmax) and a v x v histogram,
// v = C (+1 when an ignore label exists: points with target == ignore count in the extra row/column, utils/metrics.py).  One
// launch: a histogram per workgroup in LDS, then one 64-bit atomic per non-empty bin -- the tensor form is an arg-max, ten
// elementwise launches and 192000 int64 atomics onto 169 addresses.  invalid += points whose target is outside [0, v).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void confusion_kernel(int C, long N, int v, long long ignore, int has_ignore,
                                                        const float *__restrict__ logits, const long long *__restrict__ target,
                                                        long long *__restrict__ cm, long long *__restrict__ invalid)
{
    extern __shared__ int s_bins[];  // v * v + 1
    for (int i = threadIdx.x; i <= v * v; i += 256) s_bins[i] = 0;
    __syncthreads();
    const int b = blockIdx.y;
    for (long n = (long)blockIdx.x * 256 + threadIdx.x; n < N; n += (long)gridDim.x * 256) {
        const float *lp = logits + (size_t)b * C * N + n;
        float best = lp[0];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float x = lp[(size_t)c * N];
            if (x > best || (x != x && best == best)) { best = x; arg = c; }  // first maximum; a NaN wins, as in torch
        }
        long long t = target[(size_t)b * N + n];
        int p = arg;
        if (has_ignore && t == ignore) { t = v - 1; p = v - 1; }
        if (t >= 0 && t < v) atomicAdd(&s_bins[(int)t * v + p], 1);
        else atomicAdd(&s_bins[v * v], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < v * v; i += 256)
        if (s_bins[i]) atomicAdd((unsigned long long *)(cm + i), (unsigned long long)s_bins[i]);
    if (threadIdx.x == 0 && s_bins[v * v]) atomicAdd((unsigned long long *)invalid, (unsigned long long)s_bins[v * v]);
}

// cm (v*v) int64 += histogram of (target, argmax_c logits) over the B*N points; invalid (1) int64 += out-of-range targets.
// v = C + has_ignore <= 64.
AMC_API int amc3d_confusion_update(int B, int C, long N, const float *logits, const long long *target, long long ignore,
                                   int has_ignore, long long *cm, long long *invalid, void *stream_)
{
    if (B <= 0 || N <= 0) return 0;
    const int v = C + (has_ignore ? 1 : 0);
    if (C <= 0 || v > 64 || !logits || !target || !cm || !invalid) return bad_arg("amc3d_confusion_update: bad argument (at most 64 classes)");
    const int bx = (int)(div_up(N, 256) < 64 ? div_up(N, 256) : 64);
    hipLaunchKernelGGL(confusion_kernel, dim3(bx, B), dim3(256), (size_t)(v * v + 1) * sizeof(int), (hipStream_t)stream_, C, N, v, ignore,
                       has_ignore, logits, target, cm, invalid);
    return launch_status("amc3d_confusion_update");
}

AMC_API size_t amc3d_cross_entropy_workspace_bytes(int B, long N)
{
    return (size_t)(B > 0 ? B : 0) * (size_t)div_up(N > 0 ? N : 1, CE_THREADS) * 2 * sizeof(double);
}

AMC_API int amc3d_cross_entropy_forward(int B, int C, long N, const float *logits, const long long *target,
                                        long long ignore_index, float *lse, float *mean_cnt, void *workspace,
                                        size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || N <= 0) return 0;
    if (C <= 0 || !logits || !target || !lse || !mean_cnt || !workspace ||
        workspace_bytes < amc3d_cross_entropy_workspace_bytes(B, N))
        return bad_arg("amc3d_cross_entropy_forward: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const int gx = div_up(N, CE_THREADS);
    hipLaunchKernelGGL(ce_forward_kernel, dim3(gx, B), dim3(CE_THREADS), 0, stream, C, N, logits, target, ignore_index, lse,
                       (double *)workspace);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1024), 0, stream, gx * B, (const double *)workspace, mean_cnt);
    return launch_status("amc3d_cross_entropy_forward");
}

AMC_API int amc3d_cross_entropy_backward(int B, int C, long N, const float *logits, const long long *target,
                                         long long ignore_index, const float *lse, const float *mean_cnt,
                                         const float *grad_out, float *dlogits, void *stream)
{
    if (B <= 0 || N <= 0) return 0;
    if (C <= 0 || !logits || !target || !lse || !mean_cnt || !grad_out || !dlogits)
        return bad_arg("amc3d_cross_entropy_backward: bad argument");
    hipLaunchKernelGGL(ce_backward_kernel, dim3(div_up(N, CE_THREADS), B), dim3(CE_THREADS), 0, (hipStream_t)stream, C, N,
                       logits, target, ignore_index, lse, mean_cnt, grad_out, dlogits);
    return launch_status("amc3d_cross_entropy_backward");
}
