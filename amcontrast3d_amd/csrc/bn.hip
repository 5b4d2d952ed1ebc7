// Training-mode BatchNorm fused with what follows it in the PointNeXt blocks, for gfx950.
//
// The reference runs every 1x1-conv block as separate layers (models/layers/conv.py:24-102:
// nn.Conv -> nn.BatchNorm -> nn.ReLU(inplace)) and the set-abstraction blocks follow the last one by
// torch.max over the 32 neighbours (pointnext_AA.py:166): per block that is 5 passes over the
// (B,C,M,32) activation forward and 7 backward.  The arithmetic is HBM-bound, so the kernels here fuse
// passes, not math:
//     forward   bn_stats  (1 read)  ->  bn_act: y = relu(bn(x)) (1 read, 1 write)
//                                   or  bn_max: y = max_k bn(x) (1 read, a (B,C,M) write + arg-max byte)
//     backward  *_bwd_stats (1 read) -> *_bwd_apply (1-2 reads, 1 write); the max variant never
//               materialises the sparse gradient of the pooled tensor.
// Layout is the reference's channel-major (B, C, L), L = M*K contiguous; statistics are over (B, L) per
// channel, accumulated in fp64 (sums of up to 1.5M fp32 values), normalisation as PyTorch writes it:
// ((x - mean) * invstd) * gamma + beta with the biased variance (torch.nn.BatchNorm semantics).
#include "common.h"

namespace amc {

constexpr int BN_THREADS = 256;

__device__ __forceinline__ double block_sum_f64(double v, double *s_buf)
{
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_buf[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < BN_THREADS / 64; ++w) t += s_buf[w];
    __syncthreads();
    return t;
}

// The (B, Lq) positions of a channel are cut into B * cps contiguous segments ("units") of `seg` positions
// (a multiple of 4); workgroup `chunk` of a channel takes units chunk, chunk + nchunks, ...  One integer division
// per unit, 16-byte loads inside.
struct BnSplit {
    int cps, units, nchunks;
    long seg;
};

// partial[c][chunk] = {sum, sumsq} over the chunk's units; grid (nchunks, C)
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(int B, int C, long L, BnSplit sp, int vec,
                                                              const float *__restrict__ x, double *__restrict__ partial)
{
    __shared__ double s_buf[BN_THREADS / 64];
    const int c = blockIdx.y, chunk = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int u = chunk; u < sp.units; u += sp.nchunks) {
        const int b = u / sp.cps, sg = u - b * sp.cps;
        const long l0 = sg * sp.seg, l1 = min(L, l0 + sp.seg);
        const float *row = x + ((size_t)b * C + c) * L;
        if (vec) {
            for (long i = l0 + threadIdx.x * 4; i < l1; i += BN_THREADS * 4) {
                const float4 v = *reinterpret_cast<const float4 *>(row + i);
                s1 += (double)v.x + (double)v.y + ((double)v.z + (double)v.w);
                s2 += (double)v.x * (double)v.x + (double)v.y * (double)v.y +
                      ((double)v.z * (double)v.z + (double)v.w * (double)v.w);
            }
        } else {
            for (long i = l0 + threadIdx.x; i < l1; i += BN_THREADS) {
                const float v = row[i];
                s1 += (double)v;
                s2 += (double)v * (double)v;
            }
        }
    }
    s1 = block_sum_f64(s1, s_buf);
    s2 = block_sum_f64(s2, s_buf);
    if (threadIdx.x == 0) {
        partial[((size_t)c * sp.nchunks + chunk) * 2 + 0] = s1;
        partial[((size_t)c * sp.nchunks + chunk) * 2 + 1] = s2;
    }
}

// mean, biased var -> invstd; also the unbiased variance for the running estimate
__global__ void bn_finalize_kernel(int C, int nchunks, double count, float eps, const double *__restrict__ partial,
                                   float *__restrict__ mean, float *__restrict__ invstd, float *__restrict__ var_unbiased)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nchunks; ++k) {
        s1 += partial[((size_t)c * nchunks + k) * 2 + 0];
        s2 += partial[((size_t)c * nchunks + k) * 2 + 1];
    }
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    var_unbiased[c] = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
}

__device__ __forceinline__ float bn_sigmoid(float v) { return __fdiv_rn(1.f, __fadd_rn(1.f, expf(-v))); }

__device__ __forceinline__ float bn_val(float x, float mean, float invstd, float gamma, float beta)
{
    return __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(x, mean), invstd), gamma), beta);
}

// When `partial` is set, the consumer kernels derive mean / invstd from the statistics kernel's partial sums
// themselves (<= 64 pairs per channel: one wave, one LDS broadcast) instead of waiting for a separate finalize
// launch; the first workgroup of every channel also publishes them (backward needs them) and performs
// nn.BatchNorm's running-stat update, so a training-mode BN layer is two launches, not four.
struct BnFused {
    const double *partial;
    int nchunks;
    double count;
    const double *count_dev;  // set: the element count is read from device memory (cross-rank statistics)
    float eps, momentum;  // momentum < 0: no running update here (cumulative mode keeps its own launch)
    float *mean_out, *invstd_out, *var_out, *running_mean, *running_var;
    long long *tracked;
};

__device__ __forceinline__ void bn_partial_sums(const double *__restrict__ partial, int c, int nchunks, double &s1, double &s2)
{
    s1 = 0.0; s2 = 0.0;  // executed by the first wave only
    for (int k = threadIdx.x; k < nchunks; k += 64) {
        s1 += partial[((size_t)c * nchunks + k) * 2 + 0];
        s2 += partial[((size_t)c * nchunks + k) * 2 + 1];
    }
    for (int s = 32; s >= 1; s >>= 1) { s1 += __shfl_xor(s1, s, 64); s2 += __shfl_xor(s2, s, 64); }
}

// mean and invstd of channel c, either given or derived from the partial sums (`writer`: this workgroup publishes)
__device__ __forceinline__ void bn_channel_stats(const BnFused &f, int c, bool writer, const float *__restrict__ mean,
                                                 const float *__restrict__ invstd, float &m, float &is)
{
    if (!f.partial) { m = mean[c]; is = invstd[c]; return; }
    __shared__ float s_stat[2];
    if (threadIdx.x < 64) {
        double s1, s2;
        bn_partial_sums(f.partial, c, f.nchunks, s1, s2);
        if (threadIdx.x == 0) {
            const double cnt = f.count_dev ? *f.count_dev : f.count;
            const double mu = s1 / cnt;
            double var = s2 / cnt - mu * mu;
            if (var < 0.0) var = 0.0;
            const float mf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)f.eps));
            s_stat[0] = mf; s_stat[1] = isf;
            if (writer) {
                const float vu = (float)(cnt > 1.0 ? var * cnt / (cnt - 1.0) : var);
                f.mean_out[c] = mf; f.invstd_out[c] = isf; f.var_out[c] = vu;
                if (f.running_mean && f.momentum >= 0.f) {  // running.mul_(1 - m).add_(batch, alpha=m)
                    f.running_mean[c] = f.running_mean[c] * (1.f - f.momentum) + f.momentum * mf;
                    f.running_var[c] = f.running_var[c] * (1.f - f.momentum) + f.momentum * vu;
                    if (c == 0 && f.tracked) *f.tracked += 1;
                }
            }
        }
    }
    __syncthreads();
    m = s_stat[0]; is = s_stat[1];
}

// y = [relu](bn(x)), elementwise over (B, C, L); 4 elements per thread when L % 4 == 0
__global__ __launch_bounds__(BN_THREADS) void bn_act_kernel(int C, long L, int relu, const float *__restrict__ x,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float *__restrict__ y, BnFused f,
                                                            const float *__restrict__ res)
{
    const int bc = blockIdx.y;  // b * C + c
    const int c = bc % C;
    float m, is;
    bn_channel_stats(f, c, blockIdx.x == 0 && bc < C, mean, invstd, m, is);
    const float g = gamma[c], bt = beta[c];
    const float *xr = x + (size_t)bc * L;
    float *yr = y + (size_t)bc * L;
    const float *rr = res ? res + (size_t)bc * L : nullptr;  // y = [relu](bn(x) + res): the residual of an InvResMLP block
    if ((L & 3) == 0) {
        for (long i = ((long)blockIdx.x * BN_THREADS + threadIdx.x) * 4; i < L; i += (long)gridDim.x * BN_THREADS * 4) {
            float4 v = *reinterpret_cast<const float4 *>(xr + i);
            v.x = bn_val(v.x, m, is, g, bt); v.y = bn_val(v.y, m, is, g, bt);
            v.z = bn_val(v.z, m, is, g, bt); v.w = bn_val(v.w, m, is, g, bt);
            if (rr) {
                const float4 r4 = *reinterpret_cast<const float4 *>(rr + i);
                v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
            }
            if (relu == 2) {  // sigmoid (the APM towers of AMContrast3D++: Linear -> BatchNorm1d -> Sigmoid)
                v.x = bn_sigmoid(v.x); v.y = bn_sigmoid(v.y); v.z = bn_sigmoid(v.z); v.w = bn_sigmoid(v.w);
            } else if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4 *>(yr + i) = v;
        }
    } else {
        for (long i = (long)blockIdx.x * BN_THREADS + threadIdx.x; i < L; i += (long)gridDim.x * BN_THREADS) {
            float v = bn_val(xr[i], m, is, g, bt);
            if (rr) v += rr[i];
            yr[i] = relu == 2 ? bn_sigmoid(v) : (relu ? fmaxf(v, 0.f) : v);
        }
    }
}

// y[b,c,m] = max_k [relu](bn(x[b,c,m,k])), arg[b,c,m] = first k attaining it (torch.max returns the first)
__global__ __launch_bounds__(BN_THREADS) void bn_max_kernel(int C, int M, int K, int relu, const float *__restrict__ x,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float *__restrict__ y,
                                                            unsigned char *__restrict__ arg, BnFused f)
{
    const int bc = blockIdx.y;
    const int c = bc % C;
    const int mi = blockIdx.x * BN_THREADS + threadIdx.x;
    float m, is;
    bn_channel_stats(f, c, blockIdx.x == 0 && bc < C, mean, invstd, m, is);
    if (mi >= M) return;
    const float g = gamma[c], bt = beta[c];
    const float *xr = x + ((size_t)bc * M + mi) * K;
    float best = -__builtin_inff();
    int bk = 0;
    if ((K & 3) == 0) {
        for (int k = 0; k < K; k += 4) {
            const float4 v4 = *reinterpret_cast<const float4 *>(xr + k);
            const float vs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = bn_val(vs[j], m, is, g, bt);
                if (relu) v = fmaxf(v, 0.f);
                if (v > best) { best = v; bk = k + j; }
            }
        }
    } else {
        for (int k = 0; k < K; ++k) {
            float v = bn_val(xr[k], m, is, g, bt);
            if (relu) v = fmaxf(v, 0.f);
            if (v > best) { best = v; bk = k; }
        }
    }
    y[(size_t)bc * M + mi] = best;
    arg[(size_t)bc * M + mi] = (unsigned char)bk;
}

// Same result, cooperative loads: LPR = K/4 consecutive lanes own one (b,c,m) row, a wave reads 1 KiB contiguous
// per instruction instead of 64 separate 128-byte rows; the row arg-max is combined across the LPR lanes keeping
// the first index among equal values (torch.max semantics).
template <int LPR>
__global__ __launch_bounds__(BN_THREADS) void bn_max_coop_kernel(int C, int M, int relu, const float *__restrict__ x,
                                                                 const float *__restrict__ mean,
                                                                 const float *__restrict__ invstd,
                                                                 const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float *__restrict__ y,
                                                                 unsigned char *__restrict__ arg, BnFused f)
{
    const int bc = blockIdx.y;
    const int c = bc % C;
    float m, is;
    bn_channel_stats(f, c, blockIdx.x == 0 && bc < C, mean, invstd, m, is);
    const float g = gamma[c], bt = beta[c];
    const float4 *xr = reinterpret_cast<const float4 *>(x + (size_t)bc * M * (LPR * 4));
    const long nf4 = (long)M * LPR;
    for (long f = (long)blockIdx.x * BN_THREADS + threadIdx.x; f < ((nf4 + 63) & ~63L); f += (long)gridDim.x * BN_THREADS) {
        const bool live = f < nf4;
        const float4 v4 = live ? xr[f] : make_float4(0.f, 0.f, 0.f, 0.f);
        const int sg = (int)(f & (LPR - 1));
        const float vs[4] = {v4.x, v4.y, v4.z, v4.w};
        float best = -__builtin_inff();
        int bk = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = bn_val(vs[j], m, is, g, bt);
            if (relu) v = fmaxf(v, 0.f);
            if (v > best) { best = v; bk = sg * 4 + j; }
        }
#pragma unroll
        for (int sft = 1; sft < LPR; sft <<= 1) {
            const float ov = __shfl_xor(best, sft, 64);
            const int ok = __shfl_xor(bk, sft, 64);
            if (ov > best || (ov == best && ok < bk)) { best = ov; bk = ok; }
        }
        if (live && sg == 0) {
            const long mi = f / LPR;
            y[(size_t)bc * M + mi] = best;
            arg[(size_t)bc * M + mi] = (unsigned char)bk;
        }
    }
}

// ---- backward statistics: per channel  Sa = sum dq,  Sb = sum dq * xhat  (dq = upstream gradient w.r.t. the
// BN output, after the ReLU mask resp. routed through the arg-max).  partial layout as in bn_stats.
// mode 0: act.  dq = dy * (relu ? bn(x) > 0 : 1) over (B, C, L)
// mode 1: max.  dq is nonzero only at the arg-max: dq = g[b,c,m] * (relu ? y > 0 : 1), xhat at the arg-max
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_stats_kernel(
    int mode, int B, int C, long L, int K, int relu, BnSplit sp, int vec, const float *__restrict__ x,
    const float *__restrict__ dy, const unsigned char *__restrict__ arg, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    double *__restrict__ partial, const float *__restrict__ ymask)
{
    __shared__ double s_buf[BN_THREADS / 64];
    const int c = blockIdx.y, chunk = blockIdx.x;
    const float m = mean[c], is = invstd[c], g = gamma[c], bt = beta[c];
    double sa = 0.0, sb = 0.0;
    const long Lq = mode == 0 ? L : L / K;  // positions carrying a gradient per (b, c)
    auto term = [&](float d, float xv) {
        const float xh = __fmul_rn(__fsub_rn(xv, m), is);
        if (relu && !(__fadd_rn(__fmul_rn(xh, g), bt) > 0.f)) d = 0.f;
        sa += (double)d;
        sb += (double)d * (double)xh;
    };
    // ymask (mode 0): the layer's OUTPUT y = relu(bn(x) + residual); the ReLU passed where y > 0
    auto term_y = [&](float d, float xv, float yv) {
        const float xh = __fmul_rn(__fsub_rn(xv, m), is);
        if (relu == 2) d = d * (yv * (1.f - yv));  // sigmoid: dy * y (1 - y), y = the layer's output
        else if (!(yv > 0.f)) d = 0.f;
        sa += (double)d;
        sb += (double)d * (double)xh;
    };
    for (int u = chunk; u < sp.units; u += sp.nchunks) {
        const int b = u / sp.cps, sg = u - b * sp.cps;
        const long l0 = sg * sp.seg, l1 = min(Lq, l0 + sp.seg);
        const size_t base = ((size_t)b * C + c) * Lq;
        if (mode == 0 && ymask) {
            if (vec) {
                for (long i = l0 + threadIdx.x * 4; i < l1; i += BN_THREADS * 4) {
                    const float4 d4 = *reinterpret_cast<const float4 *>(dy + base + i);
                    const float4 x4 = *reinterpret_cast<const float4 *>(x + base + i);
                    const float4 y4 = *reinterpret_cast<const float4 *>(ymask + base + i);
                    term_y(d4.x, x4.x, y4.x); term_y(d4.y, x4.y, y4.y); term_y(d4.z, x4.z, y4.z); term_y(d4.w, x4.w, y4.w);
                }
            } else {
                for (long i = l0 + threadIdx.x; i < l1; i += BN_THREADS) term_y(dy[base + i], x[base + i], ymask[base + i]);
            }
        } else if (mode == 0 && vec) {
            for (long i = l0 + threadIdx.x * 4; i < l1; i += BN_THREADS * 4) {
                const float4 d4 = *reinterpret_cast<const float4 *>(dy + base + i);
                const float4 x4 = *reinterpret_cast<const float4 *>(x + base + i);
                term(d4.x, x4.x); term(d4.y, x4.y); term(d4.z, x4.z); term(d4.w, x4.w);
            }
        } else {
            for (long i = l0 + threadIdx.x; i < l1; i += BN_THREADS) {
                const size_t q = base + i;
                term(dy[q], mode == 0 ? x[q] : x[q * K + arg[q]]);
            }
        }
    }
    sa = block_sum_f64(sa, s_buf);
    sb = block_sum_f64(sb, s_buf);
    if (threadIdx.x == 0) {
        partial[((size_t)c * sp.nchunks + chunk) * 2 + 0] = sa;
        partial[((size_t)c * sp.nchunks + chunk) * 2 + 1] = sb;
    }
}

// dx = gamma * invstd * (dq - mean(dq) - xhat * mean(dq * xhat))   (training-mode BN backward)
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(
    int mode, int C, long L, int K, int relu, int vec, const float *__restrict__ x, const float *__restrict__ dy,
    const unsigned char *__restrict__ arg, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, const double *__restrict__ partial, int nchunks,
    double count, const double *__restrict__ count_dev, float *__restrict__ dgamma, float *__restrict__ dbeta,
    float *__restrict__ dx, const float *__restrict__ ymask, float *__restrict__ dres)
{
    const int bc = blockIdx.y;
    const int c = bc % C;
    const float m = mean[c], is = invstd[c], g = gamma[c], bt = beta[c];
    // Sa = sum dq, Sb = sum dq * xhat from the statistics kernel's partials (no separate finalize launch);
    // the first workgroup of the channel also publishes dbeta = Sa, dgamma = Sb
    __shared__ float s_co[2];
    if (threadIdx.x < 64) {
        double sa, sb;
        bn_partial_sums(partial, c, nchunks, sa, sb);
        if (threadIdx.x == 0) {
            const double cnt = count_dev ? *count_dev : count;
            s_co[0] = (float)(sa / cnt);
            s_co[1] = (float)(sb / cnt);
            if (blockIdx.x == 0 && bc < C && dbeta) { dbeta[c] = (float)sa; dgamma[c] = (float)sb; }
        }
    }
    __syncthreads();
    const float ma = s_co[0], mb = s_co[1], gi = __fmul_rn(g, is);
    const float *xr = x + (size_t)bc * L;
    float *dr = dx + (size_t)bc * L;
    auto one = [&](float xv, float d) {
        const float xh = __fmul_rn(__fsub_rn(xv, m), is);
        if (relu && !(__fadd_rn(__fmul_rn(xh, g), bt) > 0.f)) d = 0.f;
        return gi * (d - ma - xh * mb);
    };
    if (ymask) {  // mode 0 with a residual: mask from the layer's output, the masked gradient is also the residual's
        const float *yr = ymask + (size_t)bc * L, *dyr = dy + (size_t)bc * L;
        float *rr = dres ? dres + (size_t)bc * L : nullptr;
        auto one_y = [&](float xv, float d) { return gi * (d - ma - __fmul_rn(__fsub_rn(xv, m), is) * mb); };
        if (vec) {
            for (long i = ((long)blockIdx.x * BN_THREADS + threadIdx.x) * 4; i < L; i += (long)gridDim.x * BN_THREADS * 4) {
                const float4 x4 = *reinterpret_cast<const float4 *>(xr + i);
                const float4 y4 = *reinterpret_cast<const float4 *>(yr + i);
                float4 d4 = *reinterpret_cast<const float4 *>(dyr + i);
                if (relu == 2) {
                    d4.x *= y4.x * (1.f - y4.x); d4.y *= y4.y * (1.f - y4.y); d4.z *= y4.z * (1.f - y4.z); d4.w *= y4.w * (1.f - y4.w);
                } else {
                    if (!(y4.x > 0.f)) d4.x = 0.f;
                    if (!(y4.y > 0.f)) d4.y = 0.f;
                    if (!(y4.z > 0.f)) d4.z = 0.f;
                    if (!(y4.w > 0.f)) d4.w = 0.f;
                }
                if (dres) *reinterpret_cast<float4 *>(rr + i) = d4;
                float4 o;
                o.x = one_y(x4.x, d4.x); o.y = one_y(x4.y, d4.y); o.z = one_y(x4.z, d4.z); o.w = one_y(x4.w, d4.w);
                *reinterpret_cast<float4 *>(dr + i) = o;
            }
        } else {
            for (long i = (long)blockIdx.x * BN_THREADS + threadIdx.x; i < L; i += (long)gridDim.x * BN_THREADS) {
                const float d = relu == 2 ? dyr[i] * (yr[i] * (1.f - yr[i])) : (yr[i] > 0.f ? dyr[i] : 0.f);
                if (dres) rr[i] = d;
                dr[i] = one_y(xr[i], d);
            }
        }
        return;
    }
    if (vec) {  // L % 4 == 0, K % 4 == 0 in mode 1: the four elements share their pooled position
        const unsigned Lu = (unsigned)L, Ku = (unsigned)K;
        for (unsigned i = (blockIdx.x * BN_THREADS + threadIdx.x) * 4u; i < Lu; i += gridDim.x * BN_THREADS * 4u) {
            const float4 x4 = *reinterpret_cast<const float4 *>(xr + i);
            float4 d4;
            if (mode == 0) {
                d4 = *reinterpret_cast<const float4 *>(dy + (size_t)bc * L + i);
            } else {
                const unsigned q = i / Ku, k0 = i - q * Ku;
                const size_t qq = (size_t)bc * (Lu / Ku) + q;
                const unsigned ak = arg[qq];
                const float dv = (ak - k0) < 4u ? dy[qq] : 0.f;
                d4.x = ak == k0 ? dv : 0.f; d4.y = ak == k0 + 1 ? dv : 0.f;
                d4.z = ak == k0 + 2 ? dv : 0.f; d4.w = ak == k0 + 3 ? dv : 0.f;
            }
            float4 o;
            o.x = one(x4.x, d4.x); o.y = one(x4.y, d4.y); o.z = one(x4.z, d4.z); o.w = one(x4.w, d4.w);
            *reinterpret_cast<float4 *>(dr + i) = o;
        }
        return;
    }
    for (long i = (long)blockIdx.x * BN_THREADS + threadIdx.x; i < L; i += (long)gridDim.x * BN_THREADS) {
        float d;
        if (mode == 0) {
            d = dy[(size_t)bc * L + i];
        } else {
            const long q = i / K;
            const int k = (int)(i - q * K);
            const size_t qq = (size_t)bc * (L / K) + q;
            d = k == (int)arg[qq] ? dy[qq] : 0.f;
        }
        dr[i] = one(xr[i], d);
    }
}

// ---- short layers: one 1024-thread workgroup per channel does the statistics AND the normalisation (forward) resp. the
// two sums AND dx (backward) in ONE launch.  The coarse FeaturePropagation stages (>= 64 channels, <= 16 k elements per
// channel) are a few hundred KB: the second read of the channel comes from L2, and a launch (plus the dependency gap behind
// it in the captured step) costs more than the pass.  Statistics in fp64 in a fixed order, as in the two-kernel path.
constexpr int BNC_THREADS = 1024;
constexpr long BNC_MAX_ELEMS = 16384;  // measured: at 48 k elements per channel (64 workgroups) the two-launch form is faster

__device__ __forceinline__ void block_sum2_f64(double &a, double &b, double (*s_buf)[2])
{
    for (int s = 32; s >= 1; s >>= 1) { a += __shfl_xor(a, s, 64); b += __shfl_xor(b, s, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_buf[wave][0] = a; s_buf[wave][1] = b; }
    __syncthreads();
    a = 0.0; b = 0.0;
    for (int w = 0; w < BNC_THREADS / 64; ++w) { a += s_buf[w][0]; b += s_buf[w][1]; }
}

__global__ __launch_bounds__(BNC_THREADS) void bn_fwd_channel_kernel(int B, int C, long L, int relu, int vec, float eps,
                                                                      float momentum, const float *__restrict__ x,
                                                                      const float *__restrict__ gamma,
                                                                      const float *__restrict__ beta, float *__restrict__ y,
                                                                      float *__restrict__ mean_out, float *__restrict__ invstd_out,
                                                                      float *__restrict__ var_out, float *__restrict__ running_mean,
                                                                      float *__restrict__ running_var, long long *__restrict__ tracked)
{
    __shared__ double s_buf[BNC_THREADS / 64][2];
    __shared__ float s_stat[2];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < B; ++b) {
        const float *row = x + ((size_t)b * C + c) * L;
        if (vec) {
            for (long i = threadIdx.x * 4L; i < L; i += BNC_THREADS * 4L) {
                const float4 v = *reinterpret_cast<const float4 *>(row + i);
                s1 += (double)v.x + (double)v.y + ((double)v.z + (double)v.w);
                s2 += (double)v.x * (double)v.x + (double)v.y * (double)v.y + ((double)v.z * (double)v.z + (double)v.w * (double)v.w);
            }
        } else {
            for (long i = threadIdx.x; i < L; i += BNC_THREADS) {
                const float v = row[i];
                s1 += (double)v;
                s2 += (double)v * (double)v;
            }
        }
    }
    block_sum2_f64(s1, s2, s_buf);
    if (threadIdx.x == 0) {
        const double cnt = (double)B * (double)L;
        const double mu = s1 / cnt;
        double var = s2 / cnt - mu * mu;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)eps));
        const float vu = (float)(cnt > 1.0 ? var * cnt / (cnt - 1.0) : var);
        s_stat[0] = mf; s_stat[1] = isf;
        mean_out[c] = mf; invstd_out[c] = isf; var_out[c] = vu;
        if (running_mean && momentum >= 0.f) {
            running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * mf;
            running_var[c] = running_var[c] * (1.f - momentum) + momentum * vu;
            if (c == 0 && tracked) *tracked += 1;
        }
    }
    __syncthreads();
    const float m = s_stat[0], is = s_stat[1], g = gamma[c], bt = beta[c];
    for (int b = 0; b < B; ++b) {
        const float *xr = x + ((size_t)b * C + c) * L;
        float *yr = y + ((size_t)b * C + c) * L;
        if (vec) {
            for (long i = threadIdx.x * 4L; i < L; i += BNC_THREADS * 4L) {
                float4 v = *reinterpret_cast<const float4 *>(xr + i);
                v.x = bn_val(v.x, m, is, g, bt); v.y = bn_val(v.y, m, is, g, bt);
                v.z = bn_val(v.z, m, is, g, bt); v.w = bn_val(v.w, m, is, g, bt);
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4 *>(yr + i) = v;
            }
        } else {
            for (long i = threadIdx.x; i < L; i += BNC_THREADS) {
                const float v = bn_val(xr[i], m, is, g, bt);
                yr[i] = relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
}

__global__ __launch_bounds__(BNC_THREADS) void bn_bwd_channel_kernel(int B, int C, long L, int relu, int vec,
                                                                      const float *__restrict__ x, const float *__restrict__ dy,
                                                                      const float *__restrict__ mean, const float *__restrict__ invstd,
                                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                      float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                      float *__restrict__ dx)
{
    __shared__ double s_buf[BNC_THREADS / 64][2];
    const int c = blockIdx.x;
    const float m = mean[c], is = invstd[c], g = gamma[c], bt = beta[c];
    double sa = 0.0, sb = 0.0;
    auto term = [&](float d, float xv) {
        const float xh = __fmul_rn(__fsub_rn(xv, m), is);
        if (relu && !(__fadd_rn(__fmul_rn(xh, g), bt) > 0.f)) d = 0.f;
        sa += (double)d;
        sb += (double)d * (double)xh;
    };
    for (int b = 0; b < B; ++b) {
        const size_t base = ((size_t)b * C + c) * L;
        if (vec) {
            for (long i = threadIdx.x * 4L; i < L; i += BNC_THREADS * 4L) {
                const float4 d4 = *reinterpret_cast<const float4 *>(dy + base + i);
                const float4 x4 = *reinterpret_cast<const float4 *>(x + base + i);
                term(d4.x, x4.x); term(d4.y, x4.y); term(d4.z, x4.z); term(d4.w, x4.w);
            }
        } else {
            for (long i = threadIdx.x; i < L; i += BNC_THREADS) term(dy[base + i], x[base + i]);
        }
    }
    block_sum2_f64(sa, sb, s_buf);
    const double cnt = (double)B * (double)L;
    const float ma = (float)(sa / cnt), mb = (float)(sb / cnt), gi = __fmul_rn(g, is);
    if (threadIdx.x == 0) { dbeta[c] = (float)sa; dgamma[c] = (float)sb; }
    auto one = [&](float xv, float d) {
        const float xh = __fmul_rn(__fsub_rn(xv, m), is);
        if (relu && !(__fadd_rn(__fmul_rn(xh, g), bt) > 0.f)) d = 0.f;
        return gi * (d - ma - xh * mb);
    };
    for (int b = 0; b < B; ++b) {
        const size_t base = ((size_t)b * C + c) * L;
        if (vec) {
            for (long i = threadIdx.x * 4L; i < L; i += BNC_THREADS * 4L) {
                const float4 d4 = *reinterpret_cast<const float4 *>(dy + base + i);
                const float4 x4 = *reinterpret_cast<const float4 *>(x + base + i);
                float4 o;
                o.x = one(x4.x, d4.x); o.y = one(x4.y, d4.y); o.z = one(x4.z, d4.z); o.w = one(x4.w, d4.w);
                *reinterpret_cast<float4 *>(dx + base + i) = o;
            }
        } else {
            for (long i = threadIdx.x; i < L; i += BNC_THREADS) dx[base + i] = one(x[base + i], dy[base + i]);
        }
    }
}

static bool bn_channel_form(int B, int C, long L) { return C >= 64 && (long)B * L <= BNC_MAX_ELEMS; }

// nn.BatchNorm bookkeeping in ONE launch (torch/nn/modules/batchnorm.py: num_batches_tracked += 1, then the
// exponential -- or, momentum < 0 standing for None, cumulative -- moving average of mean and unbiased variance)
__global__ __launch_bounds__(1024) void bn_running_kernel(int C, float momentum, const float *__restrict__ mean,
                                                          const float *__restrict__ var_unbiased,
                                                          float *__restrict__ running_mean,
                                                          float *__restrict__ running_var, long long *__restrict__ tracked)
{
    const long long nt = *tracked + 1;
    const float f = momentum < 0.f ? 1.f / (float)nt : momentum;
    for (int c = threadIdx.x; c < C; c += 1024) {
        if (momentum < 0.f) {
            running_mean[c] += (mean[c] - running_mean[c]) * f;
            running_var[c] += (var_unbiased[c] - running_var[c]) * f;
        } else {
            running_mean[c] = running_mean[c] * (1.f - f) + f * mean[c];  // running.mul_(1 - m).add_(batch, alpha=m)
            running_var[c] = running_var[c] * (1.f - f) + f * var_unbiased[c];
        }
    }
    __syncthreads();  // every thread has read the old counter
    if (threadIdx.x == 0) *tracked = nt;
}

// sums[c] = {first, second} partial sum of channel c added up over the chunks (one wave per channel): the rank-local
// statistics a cross-rank BatchNorm exchanges.  With out_b/out_a set also stores them as floats (dbeta, dgamma).
__global__ __launch_bounds__(64) void bn_reduce_partials_kernel(int nchunks, const double *__restrict__ partial,
                                                                double *__restrict__ sums, float *__restrict__ out_a,
                                                                float *__restrict__ out_b)
{
    const int c = blockIdx.x;
    double s1, s2;
    bn_partial_sums(partial, c, nchunks, s1, s2);
    if (threadIdx.x == 0) {
        sums[2 * c] = s1; sums[2 * c + 1] = s2;
        if (out_a) { out_a[c] = (float)s1; out_b[c] = (float)s2; }
    }
}

constexpr int BN_MAX_CHUNKS = 64;

}  // namespace amc

using namespace amc;

AMC_API size_t amc3d_bn_workspace_bytes(int C) { return (size_t)C * BN_MAX_CHUNKS * 2 * sizeof(double); }

static BnSplit bn_split(int B, int C, long Lq)
{
    // enough workgroups to fill the chip (>= ~2048) with segments of at least 4096 positions
    BnSplit sp;
    long want = (2048 + C - 1) / C;                  // workgroups per channel
    long cps = (want + B - 1) / B;                   // segments per cloud
    const long cap = Lq / 4096 > 1 ? Lq / 4096 : 1;
    if (cps > cap) cps = cap;
    if (cps < 1) cps = 1;
    sp.seg = ((Lq + cps - 1) / cps + 3) & ~3L;
    sp.cps = (int)((Lq + sp.seg - 1) / sp.seg);
    sp.units = B * sp.cps;
    sp.nchunks = sp.units < BN_MAX_CHUNKS ? sp.units : BN_MAX_CHUNKS;
    return sp;
}

static int aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// statistics of x (B, C, L): mean, invstd = 1/sqrt(var_biased + eps), var_unbiased (for running_var)
AMC_API int amc3d_bn_stats(int B, int C, long L, float eps, const float *x, float *mean, float *invstd,
                           float *var_unbiased, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (!x || !mean || !invstd || !var_unbiased || !workspace || workspace_bytes < amc3d_bn_workspace_bytes(C))
        return bad_arg("amc3d_bn_stats: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const BnSplit sp = bn_split(B, C, L);
    const int nchunks = sp.nchunks;
    const int vec = (L % 4 == 0) && aligned16(x);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nchunks, C), dim3(BN_THREADS), 0, stream, B, C, L, sp, vec, x,
                       (double *)workspace);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(div_up(C, 64)), dim3(64), 0, stream, C, nchunks, (double)B * (double)L,
                       eps, (const double *)workspace, mean, invstd, var_unbiased);
    return launch_status("amc3d_bn_stats");
}

// y = [relu](bn(x)) over (B, C, L)
AMC_API int amc3d_bn_act(int B, int C, long L, int relu, const float *x, const float *mean, const float *invstd,
                         const float *gamma, const float *beta, float *y, void *stream)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (!x || !mean || !invstd || !gamma || !beta || !y) return bad_arg("amc3d_bn_act: null pointer");
    const long per_block = BN_THREADS * 4 * 4;
    const int gx = (int)min((L + per_block - 1) / per_block, (long)4096);
    hipLaunchKernelGGL(bn_act_kernel, dim3(gx, B * C), dim3(BN_THREADS), 0, (hipStream_t)stream, C, L, relu, x, mean,
                       invstd, gamma, beta, y, BnFused{}, (const float *)nullptr);
    return launch_status("amc3d_bn_act");
}

static int launch_bn_max(int B, int C, int M, int K, int relu, const float *x, const float *mean, const float *invstd,
                         const float *gamma, const float *beta, float *y, unsigned char *arg, const BnFused &f,
                         hipStream_t stream)
{
    const int lpr = K / 4;
    if (K % 4 == 0 && (lpr == 2 || lpr == 4 || lpr == 8 || lpr == 16) && aligned16(x)) {
        const int gx = (int)min((long)div_up((long)M * lpr, BN_THREADS * 4), 65535L);
#define AMC_BNMAX(N)                                                                                                 \
    hipLaunchKernelGGL(bn_max_coop_kernel<N>, dim3(gx, B * C), dim3(BN_THREADS), 0, stream, C, M, relu, x, mean, invstd, \
                       gamma, beta, y, arg, f)
        if (lpr == 2) AMC_BNMAX(2); else if (lpr == 4) AMC_BNMAX(4); else if (lpr == 8) AMC_BNMAX(8); else AMC_BNMAX(16);
#undef AMC_BNMAX
        return launch_status("amc3d_bn_max");
    }
    hipLaunchKernelGGL(bn_max_kernel, dim3(div_up(M, BN_THREADS), B * C), dim3(BN_THREADS), 0, stream, C, M, K, relu, x, mean,
                       invstd, gamma, beta, y, arg, f);
    return launch_status("amc3d_bn_max");
}

// y (B, C, M) = max over the K neighbours of [relu](bn(x (B, C, M, K))); arg (B, C, M) bytes, K <= 255
AMC_API int amc3d_bn_max(int B, int C, int M, int K, int relu, const float *x, const float *mean, const float *invstd,
                         const float *gamma, const float *beta, float *y, unsigned char *arg, void *stream)
{
    if (B <= 0 || C <= 0 || M <= 0) return 0;
    if (K <= 0 || K > 255 || !x || !mean || !invstd || !gamma || !beta || !y || !arg)
        return bad_arg("amc3d_bn_max: bad argument (K must be in 1..255)");
    return launch_bn_max(B, C, M, K, relu, x, mean, invstd, gamma, beta, y, arg, BnFused{}, (hipStream_t)stream);
}

// Training-mode BatchNorm forward in two launches: statistics, then normalise [+ReLU] (K == 0: y (B,C,L)) or
// normalise [+ReLU] + max over the K neighbours (K > 0: L = M*K, y (B,C,M), arg (B,C,M)).  Writes mean, invstd,
// var_unbiased (C each) and, when running_mean is given, performs nn.BatchNorm's buffer update
// (momentum < 0 = None: cumulative average).
AMC_API int amc3d_bn_forward(int B, int C, long L, int K, int relu, float eps, float momentum, const float *x,
                             const float *gamma, const float *beta, float *y, unsigned char *arg, float *mean,
                             float *invstd, float *var_unbiased, float *running_mean, float *running_var,
                             long long *num_batches_tracked, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (!x || !gamma || !beta || !y || !mean || !invstd || !var_unbiased || !workspace ||
        workspace_bytes < amc3d_bn_workspace_bytes(C) || K < 0 || K > 255 || (K > 0 && (!arg || L % K != 0)) ||
        (running_mean && (!running_var || !num_batches_tracked)))
        return bad_arg("amc3d_bn_forward: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const BnSplit sp = bn_split(B, C, L);
    const int vec = (L % 4 == 0) && aligned16(x);
    if (K == 0 && bn_channel_form(B, C, L) && !(running_mean && momentum < 0.f)) {
        hipLaunchKernelGGL(bn_fwd_channel_kernel, dim3(C), dim3(BNC_THREADS), 0, stream, B, C, L, relu, vec && aligned16(y), eps,
                           momentum, x, gamma, beta, y, mean, invstd, var_unbiased, running_mean, running_var,
                           num_batches_tracked);
        return launch_status("amc3d_bn_forward");
    }
    hipLaunchKernelGGL(bn_stats_kernel, dim3(sp.nchunks, C), dim3(BN_THREADS), 0, stream, B, C, L, sp, vec, x,
                       (double *)workspace);
    BnFused f{};
    f.partial = (const double *)workspace;
    f.nchunks = sp.nchunks;
    f.count = (double)B * (double)L;
    f.eps = eps;
    f.momentum = momentum;
    f.mean_out = mean; f.invstd_out = invstd; f.var_out = var_unbiased;
    f.running_mean = running_mean; f.running_var = running_var; f.tracked = num_batches_tracked;
    int st;
    if (K == 0) {
        const long per_block = BN_THREADS * 4 * 4;
        const int gx = (int)min((L + per_block - 1) / per_block, (long)4096);
        hipLaunchKernelGGL(bn_act_kernel, dim3(gx, B * C), dim3(BN_THREADS), 0, stream, C, L, relu, x, mean, invstd, gamma,
                           beta, y, f, (const float *)nullptr);
        st = launch_status("amc3d_bn_forward");
    } else {
        st = launch_bn_max(B, C, (int)(L / K), K, relu, x, mean, invstd, gamma, beta, y, arg, f, stream);
    }
    if (st) return st;
    if (running_mean && momentum < 0.f)  // cumulative average: needs the counter before and after -> its own launch
        hipLaunchKernelGGL(bn_running_kernel, dim3(1), dim3(1024), 0, stream, C, momentum, mean, var_unbiased, running_mean,
                           running_var, num_batches_tracked);
    return launch_status("amc3d_bn_forward");
}

// Backward of y = [relu](bn(x)) (arg == NULL, K = 1, dy (B,C,L)) or of y = max_K [relu](bn(x)) (arg given,
// dy (B,C,L/K)): dx (B,C,L), dgamma (C), dbeta (C).  Training-mode BN (batch statistics).
AMC_API int amc3d_bn_backward(int B, int C, long L, int K, int relu, const float *x, const float *dy,
                              const unsigned char *arg, const float *mean, const float *invstd, const float *gamma,
                              const float *beta, float *dx, float *dgamma, float *dbeta, void *workspace,
                              size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    const size_t need = amc3d_bn_workspace_bytes(C) + (size_t)C * 2 * sizeof(float);
    if (!x || !dy || !mean || !invstd || !gamma || !beta || !dx || !dgamma || !dbeta || !workspace ||
        workspace_bytes < need || K <= 0 || (arg && L % K != 0))
        return bad_arg("amc3d_bn_backward: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const int mode = arg ? 1 : 0;
    const BnSplit sp = bn_split(B, C, mode ? L / K : L);
    const int nchunks = sp.nchunks;
    const int vec = (L % 4 == 0) && (K % 4 == 0 || !mode) && aligned16(x) && aligned16(dy) && aligned16(dx) && L < (1L << 31);
    double *partial = (double *)workspace;
    if (mode == 0 && bn_channel_form(B, C, L)) {
        hipLaunchKernelGGL(bn_bwd_channel_kernel, dim3(C), dim3(BNC_THREADS), 0, stream, B, C, L, relu, vec, x, dy, mean, invstd,
                           gamma, beta, dgamma, dbeta, dx);
        return launch_status("amc3d_bn_backward");
    }
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(nchunks, C), dim3(BN_THREADS), 0, stream, mode, B, C, L, K, relu, sp, vec, x,
                       dy, arg, mean, invstd, gamma, beta, partial, (const float *)nullptr);
    const long per_block = BN_THREADS * 16;
    const int gx = (int)min((L + per_block - 1) / per_block, (long)4096);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gx, B * C), dim3(BN_THREADS), 0, stream, mode, C, L, K, relu, vec, x, dy, arg,
                       mean, invstd, gamma, beta, (const double *)partial, nchunks, (double)B * (double)L,
                       (const double *)nullptr, dgamma, dbeta, dx, (const float *)nullptr, (float *)nullptr);
    return launch_status("amc3d_bn_backward");
}

// y = relu(bn(x) + res) with batch statistics: the tail of an InvResMLP block (pointnext_AA.py:296-307: pwconv's last
// Conv1d -> BatchNorm1d without activation, `f += identity`, `self.act(f)`) in the two launches of a plain BatchNorm layer.
AMC_API int amc3d_bn_residual_forward(int B, int C, long L, float eps, float momentum, const float *x, const float *res,
                                      const float *gamma, const float *beta, float *y, float *mean, float *invstd,
                                      float *var_unbiased, float *running_mean, float *running_var,
                                      long long *num_batches_tracked, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (!x || !res || !gamma || !beta || !y || !mean || !invstd || !var_unbiased || !workspace ||
        workspace_bytes < amc3d_bn_workspace_bytes(C) || (running_mean && (!running_var || !num_batches_tracked)))
        return bad_arg("amc3d_bn_residual_forward: bad argument");
    if ((L % 4 == 0) && !(aligned16(x) && aligned16(res) && aligned16(y)))
        return bad_arg("amc3d_bn_residual_forward: x, res and y must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const BnSplit sp = bn_split(B, C, L);
    const int vec = (L % 4 == 0) && aligned16(x);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(sp.nchunks, C), dim3(BN_THREADS), 0, stream, B, C, L, sp, vec, x,
                       (double *)workspace);
    BnFused f{};
    f.partial = (const double *)workspace;
    f.nchunks = sp.nchunks;
    f.count = (double)B * (double)L;
    f.eps = eps;
    f.momentum = momentum;
    f.mean_out = mean; f.invstd_out = invstd; f.var_out = var_unbiased;
    f.running_mean = running_mean; f.running_var = running_var; f.tracked = num_batches_tracked;
    const long per_block = BN_THREADS * 4 * 4;
    const int gx = (int)min((L + per_block - 1) / per_block, (long)4096);
    hipLaunchKernelGGL(bn_act_kernel, dim3(gx, B * C), dim3(BN_THREADS), 0, stream, C, L, 1, x, mean, invstd, gamma, beta, y, f,
                       res);
    if (running_mean && momentum < 0.f)
        hipLaunchKernelGGL(bn_running_kernel, dim3(1), dim3(1024), 0, stream, C, momentum, mean, var_unbiased, running_mean,
                           running_var, num_batches_tracked);
    return launch_status("amc3d_bn_residual_forward");
}

// backward of the same: dq = dy * (y > 0); dres = dq; dx, dgamma, dbeta = BatchNorm backward of dq
AMC_API int amc3d_bn_residual_backward(int B, int C, long L, const float *x, const float *y, const float *dy,
                                       const float *mean, const float *invstd, const float *gamma, const float *beta,
                                       float *dx, float *dres, float *dgamma, float *dbeta, void *workspace,
                                       size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (!x || !y || !dy || !mean || !invstd || !gamma || !beta || !dx || !dres || !dgamma || !dbeta || !workspace ||
        workspace_bytes < amc3d_bn_workspace_bytes(C))
        return bad_arg("amc3d_bn_residual_backward: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const BnSplit sp = bn_split(B, C, L);
    const int vec = (L % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(dy) && aligned16(dx) && aligned16(dres) &&
                    L < (1L << 31);
    double *partial = (double *)workspace;
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(sp.nchunks, C), dim3(BN_THREADS), 0, stream, 0, B, C, L, 1, 1, sp, vec, x, dy,
                       (const unsigned char *)nullptr, mean, invstd, gamma, beta, partial, y);
    const long per_block = BN_THREADS * 16;
    const int gx = (int)min((L + per_block - 1) / per_block, (long)4096);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gx, B * C), dim3(BN_THREADS), 0, stream, 0, C, L, 1, 1, vec, x, dy,
                       (const unsigned char *)nullptr, mean, invstd, gamma, beta, (const double *)partial, sp.nchunks,
                       (double)B * (double)L, (const double *)nullptr, dgamma, dbeta, dx, y, dres);
    return launch_status("amc3d_bn_residual_backward");
}

// y = sigmoid(bn(x)) with batch statistics (the APM towers of AMContrast3D++, openpoints/AMContrast3D/APM/concatenation.py:
// nn.Linear -> nn.BatchNorm1d -> nn.Sigmoid six times per tower) in the two launches of a plain BatchNorm layer, and its backward
// (dq = dy * y (1 - y) from the saved output, then BatchNorm backward) in two
AMC_API int amc3d_bn_sigmoid_forward(int B, int C, long L, float eps, float momentum, const float *x, const float *gamma,
                                     const float *beta, float *y, float *mean, float *invstd, float *var_unbiased,
                                     float *running_mean, float *running_var, long long *num_batches_tracked, void *workspace,
                                     size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (!x || !gamma || !beta || !y || !mean || !invstd || !var_unbiased || !workspace ||
        workspace_bytes < amc3d_bn_workspace_bytes(C) || (running_mean && (!running_var || !num_batches_tracked)))
        return bad_arg("amc3d_bn_sigmoid_forward: bad argument");
    if ((L % 4 == 0) && !(aligned16(x) && aligned16(y))) return bad_arg("amc3d_bn_sigmoid_forward: x and y must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const BnSplit sp = bn_split(B, C, L);
    const int vec = (L % 4 == 0) && aligned16(x);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(sp.nchunks, C), dim3(BN_THREADS), 0, stream, B, C, L, sp, vec, x,
                       (double *)workspace);
    BnFused f{};
    f.partial = (const double *)workspace;
    f.nchunks = sp.nchunks;
    f.count = (double)B * (double)L;
    f.eps = eps;
    f.momentum = momentum;
    f.mean_out = mean; f.invstd_out = invstd; f.var_out = var_unbiased;
    f.running_mean = running_mean; f.running_var = running_var; f.tracked = num_batches_tracked;
    const long per_block = BN_THREADS * 4 * 4;
    const int gx = (int)min((L + per_block - 1) / per_block, (long)4096);
    hipLaunchKernelGGL(bn_act_kernel, dim3(gx, B * C), dim3(BN_THREADS), 0, stream, C, L, 2, x, mean, invstd, gamma, beta, y, f,
                       (const float *)nullptr);
    if (running_mean && momentum < 0.f)
        hipLaunchKernelGGL(bn_running_kernel, dim3(1), dim3(1024), 0, stream, C, momentum, mean, var_unbiased, running_mean,
                           running_var, num_batches_tracked);
    return launch_status("amc3d_bn_sigmoid_forward");
}

AMC_API int amc3d_bn_sigmoid_backward(int B, int C, long L, const float *x, const float *y, const float *dy, const float *mean,
                                      const float *invstd, const float *gamma, const float *beta, float *dx, float *dgamma,
                                      float *dbeta, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (!x || !y || !dy || !mean || !invstd || !gamma || !beta || !dx || !dgamma || !dbeta || !workspace ||
        workspace_bytes < amc3d_bn_workspace_bytes(C))
        return bad_arg("amc3d_bn_sigmoid_backward: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const BnSplit sp = bn_split(B, C, L);
    const int vec = (L % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(dy) && aligned16(dx) && L < (1L << 31);
    double *partial = (double *)workspace;
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(sp.nchunks, C), dim3(BN_THREADS), 0, stream, 0, B, C, L, 1, 2, sp, vec, x, dy,
                       (const unsigned char *)nullptr, mean, invstd, gamma, beta, partial, y);
    const long per_block = BN_THREADS * 16;
    const int gx = (int)min((L + per_block - 1) / per_block, (long)4096);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gx, B * C), dim3(BN_THREADS), 0, stream, 0, C, L, 1, 2, vec, x, dy,
                       (const unsigned char *)nullptr, mean, invstd, gamma, beta, (const double *)partial, sp.nchunks,
                       (double)B * (double)L, (const double *)nullptr, dgamma, dbeta, dx, y, (float *)nullptr);
    return launch_status("amc3d_bn_sigmoid_backward");
}

// running_mean / running_var / num_batches_tracked update of nn.BatchNorm in training mode; momentum < 0 = None
AMC_API int amc3d_bn_update_running(int C, float momentum, const float *mean, const float *var_unbiased,
                                    float *running_mean, float *running_var, long long *num_batches_tracked, void *stream)
{
    if (C <= 0) return 0;
    if (!mean || !var_unbiased || !running_mean || !running_var || !num_batches_tracked)
        return bad_arg("amc3d_bn_update_running: null pointer");
    hipLaunchKernelGGL(bn_running_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, C, momentum, mean, var_unbiased,
                       running_mean, running_var, num_batches_tracked);
    return launch_status("amc3d_bn_update_running");
}

// ---- BatchNorm with statistics over all ranks (torch.nn.SyncBatchNorm, which the reference converts every BN
// layer to whenever world_size > 1: examples/segmentation/main_AA.py:146-148, 820).  The kernels are the ones above;
// the per-channel sums leave the device function between the two launches so that the caller can all-reduce them:
//     forward    amc3d_bn_sums -> all-reduce(sums[2C] ++ count) -> amc3d_bn_forward_synced
//     backward   amc3d_bn_backward_sums -> all-reduce(dsums[2C]) -> amc3d_bn_backward_synced
// Counts are read from device memory (sums_count[2C]), so ranks may hold different numbers of positions and the
// sequence is capturable in a hipGraph.

// sums (2C doubles): {sum x, sum x^2} of channel c over this rank's (B, L)
AMC_API int amc3d_bn_sums(int B, int C, long L, const float *x, double *sums, void *workspace, size_t workspace_bytes,
                          void *stream_)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (!x || !sums || !workspace || workspace_bytes < amc3d_bn_workspace_bytes(C))
        return bad_arg("amc3d_bn_sums: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const BnSplit sp = bn_split(B, C, L);
    const int vec = (L % 4 == 0) && aligned16(x);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(sp.nchunks, C), dim3(BN_THREADS), 0, stream, B, C, L, sp, vec, x,
                       (double *)workspace);
    hipLaunchKernelGGL(bn_reduce_partials_kernel, dim3(C), dim3(64), 0, stream, sp.nchunks, (const double *)workspace, sums,
                       (float *)nullptr, (float *)nullptr);
    return launch_status("amc3d_bn_sums");
}

// second half of amc3d_bn_forward from global statistics: sums_count = 2C sums followed by the global element count
AMC_API int amc3d_bn_forward_synced(int B, int C, long L, int K, int relu, float eps, float momentum, const float *x,
                                    const double *sums_count, const float *gamma, const float *beta, float *y,
                                    unsigned char *arg, float *mean, float *invstd, float *var_unbiased,
                                    float *running_mean, float *running_var, long long *num_batches_tracked,
                                    void *stream_)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (!x || !sums_count || !gamma || !beta || !y || !mean || !invstd || !var_unbiased || K < 0 || K > 255 ||
        (K > 0 && (!arg || L % K != 0)) || (running_mean && (!running_var || !num_batches_tracked)))
        return bad_arg("amc3d_bn_forward_synced: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    BnFused f{};
    f.partial = sums_count;
    f.nchunks = 1;
    f.count_dev = sums_count + 2 * (size_t)C;
    f.eps = eps;
    f.momentum = momentum;
    f.mean_out = mean; f.invstd_out = invstd; f.var_out = var_unbiased;
    f.running_mean = running_mean; f.running_var = running_var; f.tracked = num_batches_tracked;
    int st;
    if (K == 0) {
        const long per_block = BN_THREADS * 4 * 4;
        const int gx = (int)min((L + per_block - 1) / per_block, (long)4096);
        hipLaunchKernelGGL(bn_act_kernel, dim3(gx, B * C), dim3(BN_THREADS), 0, stream, C, L, relu, x, mean, invstd, gamma,
                           beta, y, f, (const float *)nullptr);
        st = launch_status("amc3d_bn_forward_synced");
    } else {
        st = launch_bn_max(B, C, (int)(L / K), K, relu, x, mean, invstd, gamma, beta, y, arg, f, stream);
    }
    if (st) return st;
    if (running_mean && momentum < 0.f)
        hipLaunchKernelGGL(bn_running_kernel, dim3(1), dim3(1024), 0, stream, C, momentum, mean, var_unbiased, running_mean,
                           running_var, num_batches_tracked);
    return launch_status("amc3d_bn_forward_synced");
}

// dsums (2C doubles): {sum dq, sum dq * xhat} over this rank; dbeta / dgamma (C floats) = the same, rank-local
// (torch's SyncBatchNorm leaves the parameter gradients local; the gradient all-reduce averages them)
AMC_API int amc3d_bn_backward_sums(int B, int C, long L, int K, int relu, const float *x, const float *dy,
                                   const unsigned char *arg, const float *mean, const float *invstd, const float *gamma,
                                   const float *beta, double *dsums, float *dgamma, float *dbeta, void *workspace,
                                   size_t workspace_bytes, void *stream_)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (!x || !dy || !mean || !invstd || !gamma || !beta || !dsums || !dgamma || !dbeta || !workspace ||
        workspace_bytes < amc3d_bn_workspace_bytes(C) || K <= 0 || (arg && L % K != 0))
        return bad_arg("amc3d_bn_backward_sums: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const int mode = arg ? 1 : 0;
    const BnSplit sp = bn_split(B, C, mode ? L / K : L);
    const int vec = (L % 4 == 0) && (K % 4 == 0 || !mode) && aligned16(x) && aligned16(dy) && L < (1L << 31);
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(sp.nchunks, C), dim3(BN_THREADS), 0, stream, mode, B, C, L, K, relu, sp, vec,
                       x, dy, arg, mean, invstd, gamma, beta, (double *)workspace, (const float *)nullptr);
    hipLaunchKernelGGL(bn_reduce_partials_kernel, dim3(C), dim3(64), 0, stream, sp.nchunks, (const double *)workspace, dsums,
                       dbeta, dgamma);
    return launch_status("amc3d_bn_backward_sums");
}

// dx from the global {sum dq, sum dq * xhat} (dsums, 2C) and the global element count (*count, device memory)
AMC_API int amc3d_bn_backward_synced(int B, int C, long L, int K, int relu, const float *x, const float *dy,
                                     const unsigned char *arg, const float *mean, const float *invstd,
                                     const float *gamma, const float *beta, const double *dsums, const double *count,
                                     float *dx, void *stream_)
{
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (!x || !dy || !mean || !invstd || !gamma || !beta || !dsums || !count || !dx || K <= 0 || (arg && L % K != 0))
        return bad_arg("amc3d_bn_backward_synced: bad argument");
    const int mode = arg ? 1 : 0;
    const int vec = (L % 4 == 0) && (K % 4 == 0 || !mode) && aligned16(x) && aligned16(dy) && aligned16(dx) && L < (1L << 31);
    const long per_block = BN_THREADS * 16;
    const int gx = (int)min((L + per_block - 1) / per_block, (long)4096);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gx, B * C), dim3(BN_THREADS), 0, (hipStream_t)stream_, mode, C, L, K, relu,
                       vec, x, dy, arg, mean, invstd, gamma, beta, dsums, 1, 0.0, count, (float *)nullptr, (float *)nullptr,
                       dx, (const float *)nullptr, (float *)nullptr);
    return launch_status("amc3d_bn_backward_synced");
}
