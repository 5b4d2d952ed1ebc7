// Reverse adjacency (CSR) of a neighbourhood query, and what it replaces float atomics with (gfx950).
//
// The backward of a gathered layer sums, for every SOURCE point n, the gradients of all positions (m, k) that gathered
// it: idx[b, m, k] == n.  The reference does that with float atomics (group_points_grad_kernel_fast,
// group_points_gpu.cu:34-51), and so did this library; on MI355X float atomics run at ~1.3 TB/s of added bytes and stall
// the issuing wave, 3x slower than reading the same bytes.  The edges depend on coordinates only, so the geometry plan
// sorts them once per batch by target (stable LSD radix sort of (b*n + idx, position) pairs: hipCUB / rocPRIM device
// primitive) and every backward pass becomes a gather over contiguous edge lists -- no atomics, a fixed summation order,
// bit-reproducible results:
//   amc3d_group_csr            idx (b,m,k) -> rev_start (b*n + 1), rev_edge (b*m*k) positions ordered by (target, position)
//   amc3d_group_moments_csr    the geometry moments of lagg.hip from the lists (no scattered atomics, exact in-degree)
//   amc3d_grouped_conv_bn_backward_csr   the collapse pass of the first SetAbstraction layer as a gather
#include "cub_kernel_memset.h"  // hipCUB with its memsets as kernels (graph-safe)

#include <stdlib.h>

#include "common.h"

namespace amc {

constexpr double CSR_FX_D = 68719476736.0;   // = LAGG_FX_D (2^36)
constexpr double CSR_FX_M = 1073741824.0;    // = LAGG_FX_M (2^30)

static size_t csr_align(size_t v) { return (v + 255) & ~(size_t)255; }

static int csr_bits(long v)
{
    int b = 1;
    while ((1L << b) < v) ++b;
    return b;
}

static size_t csr_sort_temp(long e)
{
    size_t t = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t, (const unsigned *)nullptr, (unsigned *)nullptr, (const int *)nullptr,
                                             (int *)nullptr, (int)e);
    return csr_align(t);
}

__global__ void csr_keys_kernel(int n, long P, long E, const int *__restrict__ idx, unsigned *__restrict__ key, int *__restrict__ val)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const long b = e / P;
    const int id = idx[e];
    key[e] = (unsigned)(b * n + (id >= 0 && id < n ? id : 0));
    val[e] = (int)(e - b * P);
}

// ---- the same lists by a counting sort (AMC3D_CSR_COUNTING_SORT=1; 0.43 ms per step against the radix sort's 0.36) ---------
// Written when the radix sort turned out to be unsafe inside captured graphs (its hipMemsetAsync calls: cub_kernel_memset.h, which
// is what fixed it); kept as the alternative that needs no library sort.
__global__ void csr_degree_kernel(int n, long P, long E, const int *__restrict__ idx, int *__restrict__ deg)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const long b = e / P;
    const int id = idx[e];
    atomicAdd(deg + b * n + (id >= 0 && id < n ? id : 0), 1);
}

__global__ void csr_fill_kernel(int n, long P, long E, const int *__restrict__ idx, const int *__restrict__ rev_start,
                                int *__restrict__ cursor, int *__restrict__ rev_edge)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const long b = e / P;
    const int id = idx[e];
    const long g = b * n + (id >= 0 && id < n ? id : 0);
    rev_edge[rev_start[g] + atomicAdd(cursor + g, 1)] = (int)(e - b * P);
}

// every list in ascending order of its positions (distinct): lists of up to CSR_SORT_SHORT entries by insertion in place,
// longer ones (hubs of a ball query) by ranks -- a wave per list, through `temp` -- so that the gathers that walk the lists
// sum in a fixed order whatever order the fill's atomics produced
constexpr int CSR_SORT_SHORT = 16;
__global__ void csr_order_kernel(long G, const int *__restrict__ rev_start, int *__restrict__ rev_edge)
{
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int e0 = rev_start[g], d = rev_start[g + 1] - e0;
    if (d < 2 || d > CSR_SORT_SHORT) return;
    // the list in registers, padded with INT_MAX, through a bitonic network (static indices: no scratch, no dependent round trips
    // to global memory -- an in-place insertion sort took 250-340 us per stage: a chain of up to d^2 / 2 loads and stores)
    int r[CSR_SORT_SHORT];
#pragma unroll
    for (int i = 0; i < CSR_SORT_SHORT; ++i) r[i] = i < d ? rev_edge[e0 + i] : 0x7fffffff;
#pragma unroll
    for (int k = 2; k <= CSR_SORT_SHORT; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < CSR_SORT_SHORT; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const int lo = min(r[i], r[l]), hi = max(r[i], r[l]);
                    r[i] = up ? lo : hi;
                    r[l] = up ? hi : lo;
                }
            }
#pragma unroll
    for (int i = 0; i < CSR_SORT_SHORT; ++i)
        if (i < d) rev_edge[e0 + i] = r[i];
}

__global__ __launch_bounds__(256) void csr_order_long_kernel(long G, const int *__restrict__ rev_start, int *__restrict__ rev_edge,
                                                             int *__restrict__ temp)
{
    const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= G) return;
    const int e0 = rev_start[g], d = rev_start[g + 1] - e0;
    if (d <= CSR_SORT_SHORT) return;  // (wave-uniform)
    if (d <= 64) {  // one entry per lane, ranks by broadcasts inside the wave: no round trip through memory
        const int v = lane < d ? rev_edge[e0 + lane] : 0x7fffffff;
        int r = 0;
        for (int y = 0; y < d; ++y) r += __shfl(v, y, 64) < v;
        if (lane < d) rev_edge[e0 + r] = v;  // (every lane has loaded before any stores: the shuffles above are the barrier)
        return;
    }
    const int *l = rev_edge + e0;
    for (int x = lane; x < d; x += 64) {  // rank = number of smaller entries (the positions of a list are distinct)
        const int v = l[x];
        int r = 0;
        for (int y = 0; y < d; ++y) r += l[y] < v;
        temp[e0 + r] = v;
    }
    // (the wave reads back what its own lanes stored: the stores have reached L2 after the fence, and nothing of temp is in this CU's L1)
    __threadfence();
    for (int x = lane; x < d; x += 64) rev_edge[e0 + x] = temp[e0 + x];
}

// rev_start[g] = first sorted edge with key >= g (g = 0 .. G inclusive): binary search, no atomics
__global__ void csr_start_kernel(long G, long E, const unsigned *__restrict__ skey, int *__restrict__ rev_start)
{
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > G) return;
    long lo = 0, hi = E;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if ((long)skey[mid] < g) lo = mid + 1; else hi = mid;
    }
    rev_start[g] = (int)lo;
}

// per source point: in-degree and fixed-point dp sum, in list order (deterministic); a lane per point
__global__ void csr_moments_kernel(int n, long P, long G, const int *__restrict__ rev_start, const int *__restrict__ rev_edge,
                                   const float *__restrict__ dp, int *__restrict__ cnt, long long *__restrict__ dfx)
{
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const long b = g / n;
    const int s = rev_start[g], e = rev_start[g + 1];
    const float *d0 = dp + b * 3 * P, *d1 = d0 + P, *d2 = d1 + P;
    double x = 0.0, y = 0.0, z = 0.0;
    for (int i = s; i < e; i += 4) {  // four edges' loads in flight (the walk is a chain of dependent round trips otherwise);
        int p[4];                     // the sums keep the list order
        float vx[4], vy[4], vz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = rev_edge[min(i + u, e - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) { vx[u] = d0[p[u]]; vy[u] = d1[p[u]]; vz[u] = d2[p[u]]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u < e) { x += (double)vx[u]; y += (double)vy[u]; z += (double)vz[u]; }
    }
    cnt[g] = e - s;
    dfx[g * 3 + 0] = __double2ll_rn(x * CSR_FX_D);
    dfx[g * 3 + 1] = __double2ll_rn(y * CSR_FX_D);
    dfx[g * 3 + 2] = __double2ll_rn(z * CSR_FX_D);
}

// global sums of dp and dp dp^T over all positions: block partials in double, fixed-point integer atomics (order-free)
__global__ __launch_bounds__(256) void csr_global_moments_kernel(long P, const float *__restrict__ dp, unsigned long long *__restrict__ mom)
{
    __shared__ double red[4][9];
    const int b = blockIdx.y;
    const float *d0 = dp + (size_t)b * 3 * P, *d1 = d0 + P, *d2 = d1 + P;
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        const double dx = d0[p], dy = d1[p], dz = d2[p];
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
        atomicAdd(mom + threadIdx.x, (unsigned long long)__double2ll_rn(v * CSR_FX_M));
    }
}

__device__ __forceinline__ float csr_bn(float x, float mean, float invstd, float gamma, float beta)
{
    return __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(x, mean), invstd), gamma), beta);
}

// Collapse as a gather: a group of CT lanes owns one source point g and walks its edge list; lane = channel.
//   q[g][c] = sum_e d(e, c),  d = dx1_pm[position e][c] * [relu: bn(G[g][c] + W_dp[c] . dp_e) > 0]
//   partial sums per channel {sum d, sum d xhat, sum d dp_j}.  dx1_pm is the POSITION-major gradient (b, P, C).
// grid (point groups, channel chunks of 64, 1); a workgroup takes PTS consecutive source points per wave-group
// Workgroup shape (BS threads, MINW = the compiler's waves-per-SIMD target).  The software pipeline below wants ~142 registers;
// a 512-thread workgroup is two waves per SIMD, so at 142 registers only ONE fits a CU (two would need four waves per SIMD =
// 128 registers): 8 waves per CU for a kernel that is a chain of memory round trips -- found at the end of round 3 by listing
// registers x workgroup size for every kernel (the occupancy the compiler prints, 3, is per wave, not per workgroup).
// AMC3D_CSR_SHAPE (measured on PointNeXt-S' four launches, real room-like neighbourhoods, 4096 workgroups; SA1 / SA2-4 avg):
//   0: this kernel, 512 threads, 142 registers -> 8 waves per CU                                   210 / 61 us  (step 6.10 ms)
//   (512 threads capped at 128 registers -- 15 spilled, 16 waves per CU -- and the shfl kernel capped at 96: slower, removed)
//   2: 256 threads, 142 registers, three per CU -> 12 waves per CU                                  (synthetic lists: 182 -> 128)
//   3: csr_collapse_shfl_kernel: records distributed over the lanes, 104 registers -> 16 waves     164 / 46 us  (step 5.985)
//   5: csr_collapse_stream_kernel: the group's edge range as one stream                            153 / 55 us  (step 6.008)
//   6 (default): the stream kernel below 64 channels, the shfl kernel from 64 up                    153 / 46 us  (step 5.97)
template <int CT, int CSR_BS, int MINW>
__global__ __launch_bounds__(CSR_BS, MINW) void csr_collapse_kernel(int C, int n, long P, long G, int relu, const float *__restrict__ dx1_pm,
                                                           const float *__restrict__ g_pm, const int *__restrict__ rev_start,
                                                           const int *__restrict__ rev_edge, const float *__restrict__ dp,
                                                           const float *__restrict__ w_dp, const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float *__restrict__ Q,
                                                           double *__restrict__ partial, int pts_per_group,
                                                           const float4 *__restrict__ rev_dp)
{
    constexpr int GROUPS = CSR_BS / CT;  // point groups per workgroup
    __shared__ double red[GROUPS][CT][5];
    const int c0 = blockIdx.y * 64;
    const int cl = threadIdx.x % CT, grp = threadIdx.x / CT;
    const int c = c0 + cl;
    const float w0 = w_dp[c * 3 + 0], w1 = w_dp[c * 3 + 1], w2 = w_dp[c * 3 + 2];
    const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    const long g_begin = ((long)blockIdx.x * GROUPS + grp) * pts_per_group;
    if (rev_dp) {
        // Software pipeline over the points of this lane group: a point costs three DEPENDENT round trips (list bounds -> edge
        // records -> gradient rows); here the rows of point i, the records of point i + 1 and the bounds of point i + 2 are in
        // flight together (the first eight edges of a list -- the lists have ~8; longer ones finish in the plain loop below).
        const long E = G > 0 ? (long)rev_start[G] : 0;
        auto bounds = [&](long g_, int &s_, int &e_) {
            const bool in = g_ < G && g_ < g_begin + pts_per_group;
            s_ = in ? rev_start[g_] : 0;
            e_ = in ? rev_start[g_ + 1] : 0;
        };
        auto records = [&](int s_, int e_, float4 *r_) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                long j = min(s_ + u, e_ - 1);
                j = j < 0 ? 0 : (j >= E ? E - 1 : j);  // (an empty list reads a valid record it does not use)
                r_[u] = rev_dp[j];
            }
        };
        int s0, e0_, s1, e1_, s2, e2_;
        float4 rec[8], recn[8];
        bounds(g_begin, s0, e0_);
        bounds(g_begin + 1, s1, e1_);
        records(s0, e0_, rec);
        float gv = g_begin < G ? g_pm[g_begin * C + c] : 0.f;
        for (int i = 0; i < pts_per_group; ++i) {
            const long g = g_begin + i;
            if (g >= G) break;
            const int b = (int)(g / n);
            const float *xrow = dx1_pm + (long)b * P * C + c;
            float d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) d[u] = xrow[(long)__float_as_int(rec[u].w) * C];   // rows of point i
            records(s1, e1_, recn);                                                          // records of point i + 1
            bounds(g + 2, s2, e2_);                                                          // bounds of point i + 2
            const float gvn = (g + 1 < G) ? g_pm[(g + 1) * C + c] : 0.f;
            float q = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float y = __fmaf_rn(w0, rec[u].x, __fmaf_rn(w1, rec[u].y, __fmaf_rn(w2, rec[u].z, gv)));
                const float xh = __fmul_rn(__fsub_rn(y, mu), is);
                float dv = d[u];
                if (s0 + u >= e0_ || (relu && !(__fadd_rn(__fmul_rn(xh, ga), be) > 0.f))) dv = 0.f;
                q += dv;
                q1 = __fmaf_rn(dv, xh, q1); q2 = __fmaf_rn(dv, rec[u].x, q2); q3 = __fmaf_rn(dv, rec[u].y, q3); q4 = __fmaf_rn(dv, rec[u].z, q4);
            }
            for (int j0 = s0 + 8; j0 < e0_; j0 += 8) {  // the rest of a long list
                float4 r[8];
                float dd[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) r[u] = rev_dp[min(j0 + u, e0_ - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) dd[u] = xrow[(long)__float_as_int(r[u].w) * C];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float y = __fmaf_rn(w0, r[u].x, __fmaf_rn(w1, r[u].y, __fmaf_rn(w2, r[u].z, gv)));
                    const float xh = __fmul_rn(__fsub_rn(y, mu), is);
                    float dv = dd[u];
                    if (j0 + u >= e0_ || (relu && !(__fadd_rn(__fmul_rn(xh, ga), be) > 0.f))) dv = 0.f;
                    q += dv;
                    q1 = __fmaf_rn(dv, xh, q1); q2 = __fmaf_rn(dv, r[u].x, q2); q3 = __fmaf_rn(dv, r[u].y, q3); q4 = __fmaf_rn(dv, r[u].z, q4);
                }
            }
            a0 += (double)q; a1 += (double)q1; a2 += (double)q2; a3 += (double)q3; a4 += (double)q4;
            Q[g * C + c] = q;
            s0 = s1; e0_ = e1_; s1 = s2; e1_ = e2_; gv = gvn;
#pragma unroll
            for (int u = 0; u < 8; ++u) rec[u] = recn[u];
        }
    } else
    for (int i = 0; i < pts_per_group; ++i) {
        const long g = g_begin + i;
        if (g >= G) break;
        const long b = g / n;
        const float gv = g_pm[g * C + c];
        const float *d0 = dp + b * 3 * P, *d1 = d0 + P, *d2 = d1 + P;
        const float *xrow = dx1_pm + b * P * C + c;
        const int s = rev_start[g], e = rev_start[g + 1];
        float q = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f;  // this point's sums in fp32 (a list has ~8 edges), fp64 across points
        for (int j0 = s; j0 < e; j0 += 8) {  // eight edges at a time: their loads are independent, one latency for all
            int p[8];
            float e0[8], e1[8], e2[8], d[8];
            if (rev_dp) {  // (dp, position) of the edges in list order: one 16-byte stream instead of an id + three gathers
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 ed = rev_dp[min(j0 + u, e - 1)];
                    e0[u] = ed.x; e1[u] = ed.y; e2[u] = ed.z; p[u] = __float_as_int(ed.w);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) d[u] = xrow[(long)p[u] * C];
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = rev_edge[min(j0 + u, e - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    e0[u] = d0[p[u]]; e1[u] = d1[p[u]]; e2[u] = d2[p[u]];
                    d[u] = xrow[(long)p[u] * C];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float y = __fmaf_rn(w0, e0[u], __fmaf_rn(w1, e1[u], __fmaf_rn(w2, e2[u], gv)));
                const float xh = __fmul_rn(__fsub_rn(y, mu), is);
                float dv = d[u];
                if (j0 + u >= e || (relu && !(__fadd_rn(__fmul_rn(xh, ga), be) > 0.f))) dv = 0.f;
                q += dv;
                q1 = __fmaf_rn(dv, xh, q1); q2 = __fmaf_rn(dv, e0[u], q2); q3 = __fmaf_rn(dv, e1[u], q3); q4 = __fmaf_rn(dv, e2[u], q4);
            }
        }
        a0 += (double)q; a1 += (double)q1; a2 += (double)q2; a3 += (double)q3; a4 += (double)q4;
        Q[g * C + c] = q;
    }
    red[grp][cl][0] = a0; red[grp][cl][1] = a1; red[grp][cl][2] = a2; red[grp][cl][3] = a3; red[grp][cl][4] = a4;
    __syncthreads();
    for (int t = threadIdx.x; t < CT * 5; t += CSR_BS) {
        const int k = t / 5, v = t - k * 5;
        double sum = 0.0;
        for (int gq = 0; gq < GROUPS; ++gq) sum += red[gq][k][v];
        partial[((size_t)blockIdx.x * C + c0 + k) * 5 + v] = sum;
    }
}


// The same pass with the edge records DISTRIBUTED over the lanes of a group instead of replicated in every lane: the 32 (16, 64)
// lanes of a group all held the same eight float4 records of this point and of the next one -- 64 registers of the 142.  Here
// lane l keeps record (l & 7) of each, and the group reads a record's fields by a lane shuffle when it uses them (32 shuffles
// per point on the otherwise idle LDS pipe): ~90 registers, five waves per SIMD instead of three for a kernel whose time is
// round trips per wave.  Same sums in the same order as csr_collapse_kernel (bit-identical Q and partials for an equal shape).
template <int CT>
__global__ __launch_bounds__(256) void csr_collapse_shfl_kernel(int C, int n, long P, long G, int relu, const float *__restrict__ dx1_pm,
                                                                     const float *__restrict__ g_pm, const int *__restrict__ rev_start,
                                                                     const float *__restrict__ w_dp, const float *__restrict__ mean,
                                                                     const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                                     const float *__restrict__ beta, float *__restrict__ Q,
                                                                     double *__restrict__ partial, int pts_per_group,
                                                                     const float4 *__restrict__ rev_dp)
{
    constexpr int BS = 256, GROUPS = BS / CT;
    __shared__ double red[GROUPS][CT][5];
    const int c0 = blockIdx.y * 64;
    const int cl = threadIdx.x % CT, grp = threadIdx.x / CT;
    const int c = c0 + cl;
    const int slot = cl & 7;  // the record of a batch of eight that this lane keeps
    const float w0 = w_dp[c * 3 + 0], w1 = w_dp[c * 3 + 1], w2 = w_dp[c * 3 + 2];
    const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    const long g_begin = ((long)blockIdx.x * GROUPS + grp) * pts_per_group;
    const long E = G > 0 ? (long)rev_start[G] : 0;
    auto bounds = [&](long g_, int &s_, int &e_) {
        const bool in = g_ < G && g_ < g_begin + pts_per_group;
        s_ = in ? rev_start[g_] : 0;
        e_ = in ? rev_start[g_ + 1] : 0;
    };
    auto record = [&](int j0, int e_) {  // record j0 + slot of a list that ends at e_ (clamped: unused slots read a valid record)
        long j = min(j0 + slot, e_ - 1);
        j = j < 0 ? 0 : (j >= E ? E - 1 : j);
        return E > 0 ? rev_dp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    int s0, e0_, s1, e1_, s2, e2_;
    bounds(g_begin, s0, e0_);
    bounds(g_begin + 1, s1, e1_);
    float4 rec = record(s0, e0_);
    float gv = g_begin < G ? g_pm[g_begin * C + c] : 0.f;
    for (int i = 0; i < pts_per_group; ++i) {
        const long g = g_begin + i;
        if (g >= G) break;  // uniform over the group (g does not depend on the lane), and a group never straddles two waves
        const int b = (int)(g / n);
        const float *xrow = dx1_pm + (long)b * P * C + c;
        float d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = xrow[(long)__shfl(__float_as_int(rec.w), u, CT) * C];   // rows of point i
        const float4 recn = record(s1, e1_);                                                          // records of point i + 1
        bounds(g + 2, s2, e2_);                                                                       // bounds of point i + 2
        const float gvn = (g + 1 < G) ? g_pm[(g + 1) * C + c] : 0.f;
        float q = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f;
        auto edge = [&](float rx, float ry, float rz, float dv, bool live) {
            const float y = __fmaf_rn(w0, rx, __fmaf_rn(w1, ry, __fmaf_rn(w2, rz, gv)));
            const float xh = __fmul_rn(__fsub_rn(y, mu), is);
            if (!live || (relu && !(__fadd_rn(__fmul_rn(xh, ga), be) > 0.f))) dv = 0.f;
            q += dv;
            q1 = __fmaf_rn(dv, xh, q1); q2 = __fmaf_rn(dv, rx, q2); q3 = __fmaf_rn(dv, ry, q3); q4 = __fmaf_rn(dv, rz, q4);
        };
#pragma unroll
        for (int u = 0; u < 8; ++u)
            edge(__shfl(rec.x, u, CT), __shfl(rec.y, u, CT), __shfl(rec.z, u, CT), d[u], s0 + u < e0_);
        for (int j0 = s0 + 8; j0 < e0_; j0 += 8) {  // the rest of a long list
            const float4 r = record(j0, e0_);
            float dd[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) dd[u] = xrow[(long)__shfl(__float_as_int(r.w), u, CT) * C];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                edge(__shfl(r.x, u, CT), __shfl(r.y, u, CT), __shfl(r.z, u, CT), dd[u], j0 + u < e0_);
        }
        a0 += (double)q; a1 += (double)q1; a2 += (double)q2; a3 += (double)q3; a4 += (double)q4;
        Q[g * C + c] = q;
        s0 = s1; e0_ = e1_; s1 = s2; e1_ = e2_; gv = gvn;
        rec = recn;
    }
    red[grp][cl][0] = a0; red[grp][cl][1] = a1; red[grp][cl][2] = a2; red[grp][cl][3] = a3; red[grp][cl][4] = a4;
    __syncthreads();
    for (int t = threadIdx.x; t < CT * 5; t += BS) {
        const int k = t / 5, v = t - k * 5;
        double sum = 0.0;
        for (int gq = 0; gq < GROUPS; ++gq) sum += red[gq][k][v];
        partial[((size_t)blockIdx.x * C + c0 + k) * 5 + v] = sum;
    }
}


// The same pass as a STREAM over the edges of a lane group.  The lists of a group's consecutive points are one contiguous range
// of rev_dp, so the group walks that range in chunks of eight edges whatever the lengths of the lists: a ball query pads a
// sparse neighbourhood with repeats of its first point, which makes hubs (lists of 20-200 edges among lists of ~8), and the
// point-by-point kernels above serve everything past a list's first eight edges with two dependent round trips per eight
// edges (SA1 of PointNeXt-S on room-like clouds: 250 us against 110 us on lists of even length, tools/csr_bench.py).
// Here chunk k is multiplied while the rows of chunk k + 1 and the records of chunk k + 2 are in flight; which point an edge
// belongs to is a walk over the group's list ends (LDS), and a point's sums are closed when the walk leaves it.  The cloud of
// edge j is j / P (every position of a cloud appears in exactly one list), tracked by comparison.
// Sums, their order and the partials' layout are those of csr_collapse_kernel.  pts_per_group <= CSR_STREAM_PMAX.
constexpr int CSR_STREAM_PMAX = 16;
template <int CT>
__global__ __launch_bounds__(256) void csr_collapse_stream_kernel(int C, long P, long G, int relu, const float *__restrict__ dx1_pm,
                                                                  const float *__restrict__ g_pm, const int *__restrict__ rev_start,
                                                                  const float *__restrict__ w_dp, const float *__restrict__ mean,
                                                                  const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, float *__restrict__ Q,
                                                                  double *__restrict__ partial, int pts_per_group,
                                                                  const float4 *__restrict__ rev_dp)
{
    constexpr int BS = 256, GROUPS = BS / CT;
    __shared__ double red[GROUPS][CT][5];
    __shared__ float s_gv[GROUPS][CSR_STREAM_PMAX][CT];
    __shared__ int s_bnd[GROUPS][CSR_STREAM_PMAX + 1];
    const int c0 = blockIdx.y * 64;
    const int cl = threadIdx.x % CT, grp = threadIdx.x / CT;
    const int c = c0 + cl;
    const int slot = cl & 7;
    const float w0 = w_dp[c * 3 + 0], w1 = w_dp[c * 3 + 1], w2 = w_dp[c * 3 + 2];
    const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    const long g_begin = ((long)blockIdx.x * GROUPS + grp) * pts_per_group;
    const int npts = (int)max(0L, min((long)pts_per_group, G - g_begin));
    const long E = G > 0 ? (long)rev_start[G] : 0;
    if (npts > 0)
        for (int i = cl; i <= npts; i += CT) s_bnd[grp][i] = rev_start[g_begin + i];
    for (int i = 0; i < npts; ++i) s_gv[grp][i][cl] = g_pm[(g_begin + i) * C + c];
    __syncthreads();
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    if (npts > 0) {  // (uniform over the group; a group never straddles two waves)
        const int e_first = s_bnd[grp][0], e_last = s_bnd[grp][npts];
        const int nchunks = (e_last - e_first + 7) >> 3;
        auto record = [&](int k_) {  // record slot of chunk k_ (clamped: a slot past the range reads a valid record it does not use)
            long j = min((long)e_first + 8L * k_ + slot, (long)e_last - 1);
            j = j < 0 ? 0 : (j >= E ? E - 1 : j);
            return E > 0 ? rev_dp[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        // cloud of the edges, at the row-issue stage (one chunk ahead of the sums)
        // (a group of trailing points without edges starts at e_first = E: it has no chunk and must not form an address there)
        const long cloud0 = (long)e_first / P < (E - 1) / P ? (long)e_first / P : (E - 1) / P;
        long cloud_end = (cloud0 + 1) * P;               // first edge of the next cloud
        const float *xbase = dx1_pm + cloud0 * P * C + c;
        auto rows = [&](const float4 &r_, int k_, float *d_) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long j = (long)e_first + 8L * k_ + u;
                while (j >= cloud_end && j < e_last) { cloud_end += P; xbase += P * C; }
                d_[u] = xbase[(long)__shfl(__float_as_int(r_.w), u, CT) * C];
            }
        };
        // the point whose sums are open, at the summing stage
        int pi = 0, cur_end = s_bnd[grp][1];
        float gv = s_gv[grp][0][cl];
        float q = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f;
        auto close_point = [&]() {
            a0 += (double)q; a1 += (double)q1; a2 += (double)q2; a3 += (double)q3; a4 += (double)q4;
            Q[(g_begin + pi) * C + c] = q;
            q = q1 = q2 = q3 = q4 = 0.f;
            ++pi;
            if (pi < npts) { gv = s_gv[grp][pi][cl]; cur_end = s_bnd[grp][pi + 1]; }
        };
        float4 rec0 = record(0), rec1 = record(1);
        float d0[8], d1[8];
        if (nchunks > 0) rows(rec0, 0, d0);
        for (int k = 0; k < nchunks; ++k) {
            const float4 rec2 = record(k + 2);
            if (k + 1 < nchunks) rows(rec1, k + 1, d1);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = e_first + 8 * k + u;
                const bool live = j < e_last;
                while (live && j >= cur_end) close_point();  // (also steps over points with no edges: their Q is 0)
                const float rx = __shfl(rec0.x, u, CT), ry = __shfl(rec0.y, u, CT), rz = __shfl(rec0.z, u, CT);
                const float y = __fmaf_rn(w0, rx, __fmaf_rn(w1, ry, __fmaf_rn(w2, rz, gv)));
                const float xh = __fmul_rn(__fsub_rn(y, mu), is);
                float dv = d0[u];
                if (!live || (relu && !(__fadd_rn(__fmul_rn(xh, ga), be) > 0.f))) dv = 0.f;
                q += dv;
                q1 = __fmaf_rn(dv, xh, q1); q2 = __fmaf_rn(dv, rx, q2); q3 = __fmaf_rn(dv, ry, q3); q4 = __fmaf_rn(dv, rz, q4);
            }
            rec0 = rec1; rec1 = rec2;
#pragma unroll
            for (int u = 0; u < 8; ++u) d0[u] = d1[u];
        }
        while (pi < npts) close_point();  // the open point and the empty ones behind it
    }
    red[grp][cl][0] = a0; red[grp][cl][1] = a1; red[grp][cl][2] = a2; red[grp][cl][3] = a3; red[grp][cl][4] = a4;
    __syncthreads();
    for (int t = threadIdx.x; t < CT * 5; t += BS) {
        const int k = t / 5, v = t - k * 5;
        double sum = 0.0;
        for (int gq = 0; gq < GROUPS; ++gq) sum += red[gq][k][v];
        partial[((size_t)blockIdx.x * C + c0 + k) * 5 + v] = sum;
    }
}


// ---- reverse lists of the loss stages' k-NN edges (anchor i, neighbour slot j) restricted to the selected anchors -------
// Counting sort (in-degree by integer atomics, exclusive scan, cursor fill), then every short list is put in ascending
// order so the gather that uses it sums in a fixed order.  E = (selected anchors) * k, a device-side quantity: the launch
// covers m * k slots and the surplus threads leave at once.
__global__ void nbr_degree_kernel(int m, int k, int nbr_stride, const int *__restrict__ nbr, const int *__restrict__ sel,
                                  int *__restrict__ deg)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long w = t / k;
    if (w >= sel[0]) return;
    const int i = sel[1 + w], j = (int)(t - w * k);
    const int nb = nbr[(size_t)i * nbr_stride + j];
    if (nb >= 0 && nb < m) atomicAdd(deg + nb, 1);
}

__global__ void nbr_fill_kernel(int m, int k, int nbr_stride, const int *__restrict__ nbr, const int *__restrict__ sel,
                                const int *__restrict__ rev_start, int *__restrict__ cursor, int *__restrict__ rev_edge)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long w = t / k;
    if (w >= sel[0]) return;
    const int i = sel[1 + w], j = (int)(t - w * k);
    const int nb = nbr[(size_t)i * nbr_stride + j];
    if (nb >= 0 && nb < m) rev_edge[rev_start[nb] + atomicAdd(cursor + nb, 1)] = i * k + j;
}

// ---- mutual edges of the loss stages' k-NN graph ----------------------------------------------------------------------------
// mutual[i*k + s] = how many times anchor i stands in the list of its s-th neighbour x = nbr[i][s] (0 or 1 in a k-NN graph;
// 91 % of the edges of the 24-NN graph of an S3DIS-like batch are mutual, measured) -- counted at the FIRST slot that holds x
// when a degenerate list names x more than once, 0 at the others, so that a walk over N(i) meets every incoming edge once.
// The loss backward needs, for every point n, the sum over the anchors i that list n; for a mutual edge n meets i while it
// walks its OWN list, so only the non-mutual edges of the selected anchors (0 < a <= 1) need reverse lists: a tenth of the
// integer atomics of nbr_degree / nbr_fill over all edges.  One thread per edge; x's list is one or two cache lines.
// One wave per point i: lane s holds x_s = nbr[i][s]; the lists of two neighbours are read per step, lanes 0-31 the list of
// x_t, lanes 32-63 that of x_{t+1} -- whole 92-byte rows per half-wave instead of k scattered 4-byte loads per edge thread --
// and ALL steps' loads are issued before the first compare (the first version waited for each row in turn: 12 dependent
// round trips per wave, 267 us for the 192000 points of the finest stage).  Counting is a ballot per step, no shuffles.
// Bit 0x80 of mutual[] marks the edges the reverse lists take (non-mutual, anchor selected): the fill pass reads bytes only.
__global__ __launch_bounds__(256) void nbr_mutual_kernel(int m, int k, int nbr_stride, const int *__restrict__ nbr,
                                                         const float *__restrict__ a, unsigned char *__restrict__ mutual,
                                                         int *__restrict__ deg)
{
    constexpr int STEPS = 32;  // k <= 64: two neighbours per step
    const int lane = threadIdx.x & 63, half = lane >> 5, sub = lane & 31;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    const float ai = a[i];
    const bool selected = 0.f < ai && ai <= 1.f;
    const int x = lane < k ? nbr[(size_t)i * nbr_stride + lane] : -1;
    const bool valid = x >= 0 && x < m;
    int v0[STEPS], v1[STEPS];  // entries sub and sub + 32 of the list of this half-wave's neighbour at every step
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        v0[st] = -1; v1[st] = -1;
        if (2 * st < k) {  // wave-uniform
            const int xt = __shfl(x, 2 * st + half, 64);
            if (xt >= 0 && xt < m) {
                if (sub < k) v0[st] = nbr[(size_t)xt * nbr_stride + sub];
                if (sub + 32 < k) v1[st] = nbr[(size_t)xt * nbr_stride + sub + 32];
            }
        }
    }
    int count = 0;      // lane s: occurrences of i in the list of x_s
    bool first = true;  // lane s: no earlier slot of this list names x_s (degenerate lists only)
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        if (2 * st >= k) break;  // wave-uniform
        const unsigned long long hit = __ballot(v0[st] == i) , hit2 = __ballot(v1[st] == i);
        const int lo = __popc((unsigned)hit) + __popc((unsigned)hit2);
        const int hi = __popc((unsigned)(hit >> 32)) + __popc((unsigned)(hit2 >> 32));
        if (lane == 2 * st) count = lo;
        if (lane == 2 * st + 1) count = hi;
        const int xa = __shfl(x, 2 * st, 64), xb = __shfl(x, 2 * st + 1, 64);
        first &= !(lane > 2 * st && xa == x) && !(lane > 2 * st + 1 && xb == x);
    }
    if (lane < k) {
        const bool listed = valid && count == 0 && selected;
        mutual[(size_t)i * k + lane] = (unsigned char)((first ? (count < 127 ? count : 127) : 0) | (listed ? 0x80 : 0));
        if (listed) atomicAdd(deg + x, 1);
    }
}

// The same bytes from the squared distances of the search that made the lists (amc3d_knnquery's dist2, same strides): in a
// k-NN graph i is in the list of x exactly when d(i,x) is below x's largest kept distance -- ONE 4-byte gather per edge from a
// compact, L2-resident array instead of x's whole list (the row-scan kernel above is bound by the rate of 128-byte lines out
// of the Infinity Cache: 213 us for the 4.4 M edges of the finest stage).  dist2_ref(q, p) is bitwise symmetric in its
// arguments, so both ends compare the same number.  Edges the comparison cannot decide -- equality with the bound (the
// reference breaks such ties by scan order), zero distances (coincident points: column 0 of the search need not be the point
// itself), unfilled slots of segments shorter than k (1e10) -- are decided as above, by scanning x's list.  dist2 may hold the
// distances or their squares (the Python wrapper returns roots, pointops.py:55): any monotone image decides the same edges.
__global__ void nbr_radius_kernel(int m, int k, int stride, const float *__restrict__ dist2, float *__restrict__ rk)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) rk[i] = dist2[(size_t)i * stride + k - 1];
}

__global__ __launch_bounds__(256) void nbr_mutual_dist_kernel(int m, int k, int nbr_stride, const int *__restrict__ nbr,
                                                              const float *__restrict__ dist2, const float *__restrict__ rk,
                                                              const float *__restrict__ a, unsigned char *__restrict__ mutual,
                                                              int *__restrict__ deg)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)m * k) return;
    const int i = (int)(t / k), s = (int)(t - (long)i * k);
    const int *li = nbr + (size_t)i * nbr_stride;
    const int x = li[s];
    const bool valid = x >= 0 && x < m;
    const float d = dist2[(size_t)i * nbr_stride + s];
    int count = 0;
    bool first = true;
    if (valid) {
        const float r = rk[x];
        if (d > 0.f && d < 9.0e4f && r < 9.0e4f && d != r) {  // (the placeholder of an unfilled slot is 1e10, or its root 1e5)
            count = d < r ? 1 : 0;
        } else {  // undecidable by distance: scan (rare)
            const int *lx = nbr + (size_t)x * nbr_stride;
            for (int j = 0; j < k; ++j) count += lx[j] == i;
            for (int j = 0; j < s; ++j) first &= li[j] != x;
        }
    }
    const float ai = a[i];
    const bool listed = valid && count == 0 && 0.f < ai && ai <= 1.f;
    mutual[t] = (unsigned char)((first ? (count < 127 ? count : 127) : 0) | (listed ? 0x80 : 0));
    if (listed) atomicAdd(deg + x, 1);
}

__global__ void nbr_fill_nonmutual_kernel(int m, int k, int nbr_stride, const int *__restrict__ nbr,
                                          const unsigned char *__restrict__ mutual, const int *__restrict__ rev_start,
                                          int *__restrict__ cursor, int *__restrict__ rev_edge)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)m * k || !(mutual[t] & 0x80)) return;
    const int i = (int)(t / k), s = (int)(t - (long)i * k);
    const int x = nbr[(size_t)i * nbr_stride + s];
    rev_edge[rev_start[x] + atomicAdd(cursor + x, 1)] = (int)t;
}

constexpr int NBR_SORT_MAX = 96;  // longer lists (degenerate clouds) keep their fill order
__global__ void nbr_order_kernel(int m, const int *__restrict__ rev_start, int *__restrict__ rev_edge)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= m) return;
    const int e0 = rev_start[n], d = rev_start[n + 1] - e0;
    if (d < 2 || d > NBR_SORT_MAX) return;
    int *l = rev_edge + e0;
    for (int x = 1; x < d; ++x) {  // insertion sort: the lists hold ~k entries
        const int v = l[x];
        int y = x - 1;
        while (y >= 0 && l[y] > v) { l[y + 1] = l[y]; --y; }
        l[y + 1] = v;
    }
}
}  // namespace amc

using namespace amc;

AMC_API size_t amc3d_group_csr_workspace_bytes(int b, int npoints, int nsample)
{
    const long E = (long)b * npoints * nsample;
    if (E <= 0) return 0;
    return 2 * csr_align((size_t)E * 4) + csr_align((size_t)E * 4) + csr_sort_temp(E) + 256;  // keys, sorted keys, values, temp
}

// rev_start (b*n + 1) int32: rev_edge[rev_start[b*n + j] .. rev_start[b*n + j + 1]) are the positions p = m*nsample + k of
// batch b whose neighbour index is j, in ascending p; rev_edge (b*npoints*nsample) int32
AMC_API int amc3d_group_csr(int b, int n, int npoints, int nsample, const int *idx, int *rev_start, int *rev_edge, void *workspace,
                            size_t workspace_bytes, void *stream_)
{
    const long P = (long)npoints * nsample, E = (long)b * P, G = (long)b * n;
    if (E <= 0 || n <= 0) return 0;
    if (!idx || !rev_start || !rev_edge || !workspace || workspace_bytes < amc3d_group_csr_workspace_bytes(b, npoints, nsample) ||
        E >= (1L << 31) || G >= (1L << 31))
        return bad_arg("amc3d_group_csr: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    char *w = (char *)workspace;
    {   // counting sort on request, wherever the workspace holds its arrays: G + 1 <~ E
        size_t scan_temp = 0;
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_temp, (const int *)nullptr, (int *)nullptr, (int)(G + 1));
        const size_t need = 2 * csr_align((size_t)(G + 1) * 4) + csr_align(scan_temp) + csr_align((size_t)E * 4);
        static const bool counting = getenv("AMC3D_CSR_COUNTING_SORT") != nullptr;
        if (counting && need <= workspace_bytes) {
            int *deg = (int *)w; w += csr_align((size_t)(G + 1) * 4);
            int *cursor = (int *)w; w += csr_align((size_t)(G + 1) * 4);
            void *st = (void *)w; w += csr_align(scan_temp);
            int *tmp = (int *)w;
            if (int s0 = fill_i32(deg, 0, (size_t)(cursor - deg) + G + 1, stream)) return s0;  // deg and cursor, one launch
            hipLaunchKernelGGL(csr_degree_kernel, dim3(div_up(E, 256)), dim3(256), 0, stream, n, P, E, idx, deg);
            const hipError_t es = hipcub::DeviceScan::ExclusiveSum(st, scan_temp, (const int *)deg, rev_start, (int)(G + 1), stream);
            if (es != hipSuccess) { set_error("amc3d_group_csr: scan: %s", hipGetErrorString(es)); return (int)es; }
            hipLaunchKernelGGL(csr_fill_kernel, dim3(div_up(E, 256)), dim3(256), 0, stream, n, P, E, idx, (const int *)rev_start, cursor,
                               rev_edge);
            hipLaunchKernelGGL(csr_order_kernel, dim3(div_up(G, 256)), dim3(256), 0, stream, G, (const int *)rev_start, rev_edge);
            hipLaunchKernelGGL(csr_order_long_kernel, dim3(div_up(G, 4)), dim3(256), 0, stream, G, (const int *)rev_start, rev_edge, tmp);
            return launch_status("amc3d_group_csr");
        }
    }
    unsigned *key = (unsigned *)w; w += csr_align((size_t)E * 4);
    unsigned *skey = (unsigned *)w; w += csr_align((size_t)E * 4);
    int *val = (int *)w; w += csr_align((size_t)E * 4);
    size_t temp = csr_sort_temp(E);
    hipLaunchKernelGGL(csr_keys_kernel, dim3(div_up(E, 256)), dim3(256), 0, stream, n, P, E, idx, key, val);
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(w, temp, (const unsigned *)key, skey, (const int *)val, rev_edge, (int)E, 0,
                                                            csr_bits(G), stream);
    if (e != hipSuccess) { set_error("amc3d_group_csr: radix sort: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(csr_start_kernel, dim3(div_up(G + 1, 256)), dim3(256), 0, stream, G, E, (const unsigned *)skey, rev_start);
    return launch_status("amc3d_group_csr");
}

// Loss stage: rev (m + 1 + m*k) int32 = [rev_start (m+1) | rev_edge]: rev_edge[rev_start[n] .. rev_start[n+1]) are the
// positions i*k + j, ascending, of the SELECTED anchors i (list sel of amc3d_select_anchors) whose neighbour j is n.
namespace amc {
__global__ void csr_edge_dp_kernel(long P, long E, const int *__restrict__ rev_edge, const float *__restrict__ dp,
                                   float4 *__restrict__ rev_dp)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const long b = e / P;  // every position is one edge: cloud b's lists fill [b * P, (b + 1) * P)
    const int p = rev_edge[e];
    const float *d = dp + (size_t)b * 3 * P;
    rev_dp[e] = make_float4(d[p], d[P + p], d[2 * P + p], __int_as_float(p));
}
}  // namespace amc

// rev_dp (b*npoints*nsample, 4) fp32: for every edge of the reverse lists of amc3d_group_csr, in list order, the relative
// position dp (b,3,npoints,nsample) of its position p and p itself (its int32 bits in the fourth float) -- what the collapse
// of amc3d_grouped_conv_bn_backward_csr then reads as ONE 16-byte stream instead of an edge id and three 4-byte gathers
AMC_API int amc3d_group_csr_dp(int b, int npoints, int nsample, const int *rev_edge, const float *dp, float *rev_dp, void *stream)
{
    const long P = (long)npoints * nsample, E = (long)b * P;
    if (E <= 0) return 0;
    if (!rev_edge || !dp || !rev_dp || ((uintptr_t)rev_dp & 15) || E >= (1L << 31)) return bad_arg("amc3d_group_csr_dp: bad argument");
    hipLaunchKernelGGL(csr_edge_dp_kernel, dim3(div_up(E, 256)), dim3(256), 0, (hipStream_t)stream, P, E, rev_edge, dp, (float4 *)rev_dp);
    return launch_status("amc3d_group_csr_dp");
}

AMC_API size_t amc3d_contrast_csr_workspace_bytes(int m)
{
    if (m <= 0) return 0;
    size_t t = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t, (const int *)nullptr, (int *)nullptr, m + 1);
    return 2 * csr_align((size_t)(m + 1) * 4) + csr_align(t) + 256;
}

AMC_API int amc3d_contrast_csr(int m, int k, int nbr_stride, const int *nbr, const int *sel, int *rev, void *workspace,
                               size_t workspace_bytes, void *stream_)
{
    if (m <= 0) return 0;
    if (k <= 0 || nbr_stride < k || !nbr || !sel || !rev || !workspace || (long)m * k >= (1L << 31) ||
        workspace_bytes < amc3d_contrast_csr_workspace_bytes(m))
        return bad_arg("amc3d_contrast_csr: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    char *w = (char *)workspace;
    int *deg = (int *)w; w += csr_align((size_t)(m + 1) * 4);
    int *cursor = (int *)w; w += csr_align((size_t)(m + 1) * 4);
    size_t temp = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, temp, (const int *)nullptr, (int *)nullptr, m + 1);
    if (int st = fill_i32(deg, 0, (size_t)(cursor - deg) + m + 1, stream)) return st;  // deg and cursor, one launch
    const long slots = (long)m * k;
    hipLaunchKernelGGL(nbr_degree_kernel, dim3(div_up(slots, 256)), dim3(256), 0, stream, m, k, nbr_stride, nbr, sel, deg);
    const hipError_t e = hipcub::DeviceScan::ExclusiveSum(w, temp, (const int *)deg, rev, m + 1, stream);
    if (e != hipSuccess) { set_error("amc3d_contrast_csr: scan: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(nbr_fill_kernel, dim3(div_up(slots, 256)), dim3(256), 0, stream, m, k, nbr_stride, nbr, sel,
                       (const int *)rev, cursor, rev + m + 1);
    hipLaunchKernelGGL(nbr_order_kernel, dim3(div_up(m, 256)), dim3(256), 0, stream, m, (const int *)rev, rev + m + 1);
    return launch_status("amc3d_contrast_csr");
}

// mutual (m*k) bytes and rev = [rev_start (m+1) | rev_edge (<= m*k)]: the positions i*k + s, ascending per row, of the
// NON-mutual edges of the selected anchors (0 < a[i] <= 1) that point at the row.  Workspace as amc3d_contrast_csr.
AMC_API size_t amc3d_contrast_mutual_workspace_bytes(int m) { return amc3d_contrast_csr_workspace_bytes(m) + csr_align((size_t)(m > 0 ? m : 0) * 4); }

// dist2: NULL, or the squared distances of the search that produced nbr (same row stride, pointing at the same first column):
// the fast path of nbr_mutual_dist_kernel.  Only for lists that ARE the k nearest of their row (amc3d_knnquery, self search).
AMC_API int amc3d_contrast_mutual(int m, int k, int nbr_stride, const int *nbr, const float *dist2, const float *a,
                                  unsigned char *mutual, int *rev, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (m <= 0) return 0;
    if (k <= 0 || k > 64 || nbr_stride < k || !nbr || !a || !mutual || !rev || !workspace || (long)m * k >= (1L << 31) ||
        workspace_bytes < amc3d_contrast_mutual_workspace_bytes(m))
        return bad_arg("amc3d_contrast_mutual: bad argument (k must be in 1..64)");
    hipStream_t stream = (hipStream_t)stream_;
    char *w = (char *)workspace;
    int *deg = (int *)w; w += csr_align((size_t)(m + 1) * 4);
    int *cursor = (int *)w; w += csr_align((size_t)(m + 1) * 4);
    size_t temp = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, temp, (const int *)nullptr, (int *)nullptr, m + 1);
    if (int st = fill_i32(deg, 0, (size_t)(cursor - deg) + m + 1, stream)) return st;  // deg and cursor, one launch
    const long slots = (long)m * k;
    if (dist2) {
        float *rk = (float *)((char *)workspace + amc3d_contrast_csr_workspace_bytes(m));
        hipLaunchKernelGGL(nbr_radius_kernel, dim3(div_up(m, 256)), dim3(256), 0, stream, m, k, nbr_stride, dist2, rk);
        hipLaunchKernelGGL(nbr_mutual_dist_kernel, dim3(div_up(slots, 256)), dim3(256), 0, stream, m, k, nbr_stride, nbr, dist2,
                           (const float *)rk, a, mutual, deg);
    } else {
        hipLaunchKernelGGL(nbr_mutual_kernel, dim3(div_up(m, 4)), dim3(256), 0, stream, m, k, nbr_stride, nbr, a, mutual, deg);
    }
    const hipError_t e = hipcub::DeviceScan::ExclusiveSum(w, temp, (const int *)deg, rev, m + 1, stream);
    if (e != hipSuccess) { set_error("amc3d_contrast_mutual: scan: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(nbr_fill_nonmutual_kernel, dim3(div_up(slots, 256)), dim3(256), 0, stream, m, k, nbr_stride, nbr,
                       (const unsigned char *)mutual, (const int *)rev, cursor, rev + m + 1);
    hipLaunchKernelGGL(nbr_order_kernel, dim3(div_up(m, 256)), dim3(256), 0, stream, m, (const int *)rev, rev + m + 1);
    return launch_status("amc3d_contrast_mutual");
}

// the moments buffer of amc3d_group_moments (same layout) from the reverse lists: exact in-degree, dp sums in list order
AMC_API int amc3d_group_moments_csr(int b, int n, int npoints, int nsample, const int *rev_start, const int *rev_edge,
                                    const float *dp, void *moments, size_t moments_bytes, void *stream_)
{
    if (b <= 0 || n <= 0) return 0;
    if (!rev_start || !rev_edge || !dp || !moments || moments_bytes < amc3d_group_moments_bytes(b, n) || npoints <= 0 || nsample <= 0)
        return bad_arg("amc3d_group_moments_csr: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    const long P = (long)npoints * nsample, G = (long)b * n;
    char *p = (char *)moments;
    unsigned long long *mom = (unsigned long long *)p;
    int *cnt = (int *)(p + 16 * 8);
    long long *dfx = (long long *)(p + 16 * 8 + (size_t)G * 4 + ((size_t)G & 1) * 4);
    if (int st = fill_i32((int *)moments, 0, 32, stream)) return st;  // the 16 global 64-bit words
    hipLaunchKernelGGL(csr_moments_kernel, dim3(div_up(G, 256)), dim3(256), 0, stream, n, P, G, rev_start, rev_edge, dp, cnt, dfx);
    const int blocks = (int)(div_up(P, 256 * 8) < 512 ? div_up(P, 256 * 8) : 512);
    hipLaunchKernelGGL(csr_global_moments_kernel, dim3(blocks, b), dim3(256), 0, stream, P, dp, mom);
    return launch_status("amc3d_group_moments_csr");
}

static int csr_shape()
{
    static const int v = getenv("AMC3D_CSR_SHAPE") ? atoi(getenv("AMC3D_CSR_SHAPE")) : 6;
    return v;
}
static int csr_bs() { return csr_shape() >= 2 ? 256 : 512; }

static int csr_pts_per_group(long G, int groups_per_wg)
{
    // ~4096 workgroups (round 3; 2048 of 1024 threads before): a group walks its points one after the other, so what counts is
    // FEW points per group -- 3-6 at SA1 instead of 12 -- while the number of partial sums (one per workgroup) stays where the
    // finalize kernel reads them quickly (2048: 13 us per finalize instead of 19, but the step is 0.05 ms slower; 8192: slower)
    static const long wgs = getenv("AMC3D_CSR_WGS") ? atol(getenv("AMC3D_CSR_WGS")) : 4096L;
    long per = (G + wgs * groups_per_wg - 1) / (wgs * groups_per_wg);
    return (int)(per < 1 ? 1 : per);
}

namespace amc {
size_t csr_partials(int b, int cout, int n)
{
    const int ct = cout < 64 ? cout : 64;
    const int groups = csr_bs() / ct;
    const long G = (long)b * n;
    const int per = csr_pts_per_group(G, groups);
    return (size_t)div_up(G, (long)groups * per);
}
}  // namespace amc

// stage 1 of amc3d_grouped_conv_bn_backward without atomics: Q (b,n,cout) and the per-workgroup partial sums
// [nparts][cout][5] doubles from the POSITION-major gradient dx1_pm (b, npoints*nsample, cout); the caller finishes with the
// finalize / apply kernels of lagg.hip (amc3d_grouped_conv_bn_backward does all of it when given rev lists)
namespace amc {
int csr_collapse(int b, int cout, int n, int npoints, int nsample, int relu, const float *dx1_pm, const float *g_pm,
                 const int *rev_start, const int *rev_edge, const float *rev_dp, const float *dp, const float *w_dp,
                 const float *mean, const float *invstd, const float *gamma, const float *beta, float *Q, double *partial,
                 int *nparts, hipStream_t stream)
{
    const long P = (long)npoints * nsample, G = (long)b * n;
    const int ct = cout < 64 ? cout : 64;
    const int bs = csr_bs(), shape = csr_shape();
    const int groups = bs / ct;
    const int per = csr_pts_per_group(G, groups);
    const int wgs = div_up(G, (long)groups * per);
    *nparts = wgs;
#define AMC_CSR_(CTV, BS, MINW)                                                                                               \
    hipLaunchKernelGGL((csr_collapse_kernel<CTV, BS, MINW>), dim3(wgs, cout / ct), dim3(BS), 0, stream, cout, n, P, G, relu,    \
                       dx1_pm, g_pm, rev_start, rev_edge, dp, w_dp, mean, invstd, gamma, beta, Q, partial, per,                \
                       (const float4 *)rev_dp)
#define AMC_CSR(CTV)                                                                                                          \
    do {                                                                                                                      \
        if ((shape == 5 || (shape == 6 && CTV < 64)) && rev_dp && per <= CSR_STREAM_PMAX)                                     \
            hipLaunchKernelGGL((csr_collapse_stream_kernel<CTV>), dim3(wgs, cout / ct), dim3(256), 0, stream, cout, P, G, relu, \
                               dx1_pm, g_pm, rev_start, w_dp, mean, invstd, gamma, beta, Q, partial, per,                      \
                               (const float4 *)rev_dp);                                                                       \
        else if (shape >= 3 && rev_dp)                                                                                        \
            hipLaunchKernelGGL((csr_collapse_shfl_kernel<CTV>), dim3(wgs, cout / ct), dim3(256), 0, stream, cout, n, P, G,  \
                               relu, dx1_pm, g_pm, rev_start, w_dp, mean, invstd, gamma, beta, Q, partial, per,                \
                               (const float4 *)rev_dp);                                                                       \
        else if (shape >= 2) AMC_CSR_(CTV, 256, 1);                                                                           \
        else AMC_CSR_(CTV, 512, 1);                                                                                           \
    } while (0)
    switch (ct) { case 8: AMC_CSR(8); break; case 16: AMC_CSR(16); break; case 32: AMC_CSR(32); break; default: AMC_CSR(64); }
#undef AMC_CSR
#undef AMC_CSR_
    return launch_status("csr_collapse");
}
}  // namespace amc
