// Furthest point sampling for gfx950.
//
// Reference: pointnet2_batch/src/sampling_gpu.cu:100-216 (+ __update :93-98,
// block size cuda_utils.h:10-14).  One workgroup per cloud, idxs[0] = 0, then
// m-1 dependent iterations: update every point's running minimum distance to
// the selected set, take the arg-max.
//
// The reference keeps the running minimum in global memory and re-reads xyz from
// global every iteration.  Here the workgroup has exactly the reference's block
// size RB = opt_n_threads(n) and thread t owns the same points k = t + RB*j,
// but x/y/z/min-dist of its points live in VGPRs for the whole kernel
// (PPT = ceil(n/RB) of each), so an iteration touches no memory except the
// broadcast of the winner.  The arg-max is a DPP/bpermute wave reduction plus a
// <=16-entry LDS step instead of a 10-level __syncthreads tree.
//
// Tie rule (bit-exact with the reference): inside a thread the first strict
// maximum in ascending j wins (same scan order); across threads the reference's
// tree `dists_i[t] = v2 > v1 ? i2 : i1` over strides RB/2 ... 1 keeps, among
// equal values, the thread whose bit-reversed id (log2 RB bits) is smallest.
// The reduction here orders candidates by (value desc, bitrev(tid) asc).
#include "common.h"

namespace amc {

struct Cand {
    float v;
    int key;  // bit-reversed thread id: smaller wins on equal v
    int k;    // point index
};

__device__ __forceinline__ Cand better(Cand a, Cand b)
{
    const bool take_b = (b.v > a.v) || (b.v == a.v && b.key < a.key);
    return take_b ? b : a;
}

__device__ __forceinline__ Cand shfl_xor(Cand c, int m)
{
    Cand r;
    r.v = __shfl_xor(c.v, m, 64);
    r.key = __shfl_xor(c.key, m, 64);
    r.k = __shfl_xor(c.k, m, 64);
    return r;
}

template <int PPT>
__global__ __launch_bounds__(1024) void fps_kernel(int n, int m, int log2rb,
                                                   const float *__restrict__ dataset,
                                                   float *__restrict__ temp, int *__restrict__ idxs)
{
    __shared__ float s_v[2][16];
    __shared__ int s_key[2][16], s_k[2][16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = (blockDim.x + 63) >> 6;
    const int rb = 1 << log2rb;  // reference block size; blockDim.x = max(rb, 64)
    const float *pts = dataset + (size_t)blockIdx.x * n * 3;
    int *out = idxs + (size_t)blockIdx.x * m;

    float px[PPT], py[PPT], pz[PPT], pt[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int k = tid + rb * j;
        const bool ok = tid < rb && k < n;
        px[j] = ok ? pts[(size_t)k * 3 + 0] : 0.f;
        py[j] = ok ? pts[(size_t)k * 3 + 1] : 0.f;
        pz[j] = ok ? pts[(size_t)k * 3 + 2] : 0.f;
        // running minimum distance: the caller's fill value (1e10, subsample.py:95)
        pt[j] = ok ? (temp ? temp[(size_t)blockIdx.x * n + k] : 1e10f) : -2.f;
    }
    const int key = tid < rb ? (int)(__brev((unsigned)tid) >> (32 - max(log2rb, 1))) : 0x7fffffff;

    float x1 = pts[0], y1 = pts[1], z1 = pts[2];  // old = 0
    if (tid == 0) out[0] = 0;

    for (int it = 1; it < m; ++it) {
        // ---- update running minima, per-thread first strict maximum -------------
        float best = -1.f;
        int bestj = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float d = dist2_ref(px[j], py[j], pz[j], x1, y1, z1);
            // slots past the end hold -2 and never win (min keeps -2 < best = -1)
            const float d2 = fminf(d, pt[j]);
            pt[j] = d2;
            bestj = d2 > best ? j : bestj;
            best = fmaxf(best, d2);
            // keep the unrolled bodies in order: interleaving them costs ~1.5 VGPRs per
            // point and spills at 24 points per thread (96 of the 128 VGPRs are the cloud)
            if (PPT >= 16) __builtin_amdgcn_sched_barrier(0);
        }
        Cand c{best, key, tid + rb * bestj};
        // ---- wave arg-max --------------------------------------------------------
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) c = better(c, shfl_xor(c, s));
        if (nwaves > 1) {
            // double-buffered by iteration parity: one barrier per iteration is enough
            const int buf = it & 1;
            if (lane == 0) { s_v[buf][wave] = c.v; s_key[buf][wave] = c.key; s_k[buf][wave] = c.k; }
            __syncthreads();
            Cand w{-3.f, 0x7fffffff, 0};
            if (lane < nwaves) { w.v = s_v[buf][lane]; w.key = s_key[buf][lane]; w.k = s_k[buf][lane]; }
#pragma unroll
            for (int s = 8; s >= 1; s >>= 1) w = better(w, shfl_xor(w, s));
            c = w;  // lanes 0..15 of every wave now agree on the block winner
        }
        // every wave knows the winner: fetch its coordinates with a wave-uniform
        // (scalar) load from L2 -- no second barrier, no LDS broadcast
        const int old = __builtin_amdgcn_readfirstlane(c.k);
        x1 = pts[(size_t)old * 3 + 0]; y1 = pts[(size_t)old * 3 + 1]; z1 = pts[(size_t)old * 3 + 2];
        if (tid == 0) out[it] = old;
    }

    if (temp) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int k = tid + rb * j;
            if (tid < rb && k < n) temp[(size_t)blockIdx.x * n + k] = pt[j];
        }
    }
}

// Large clouds (more than 24 points per reference thread): same algorithm with the
// running minimum in global memory, as the reference does.  Not on the benchmark
// path (n <= 24576 there); kept so the entry point has no size limit.
__global__ __launch_bounds__(1024) void fps_kernel_large(int n, int m, const float *__restrict__ dataset,
                                                         float *__restrict__ temp, int *__restrict__ idxs)
{
    __shared__ float s_v[16];
    __shared__ int s_key[16], s_k[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *pts = dataset + (size_t)blockIdx.x * n * 3;
    float *tmp = temp + (size_t)blockIdx.x * n;
    int *out = idxs + (size_t)blockIdx.x * m;
    const int key = (int)(__brev((unsigned)tid) >> 22);
    int old = 0;
    if (tid == 0) out[0] = 0;
    for (int it = 1; it < m; ++it) {
        const float x1 = pts[(size_t)old * 3], y1 = pts[(size_t)old * 3 + 1], z1 = pts[(size_t)old * 3 + 2];
        float best = -1.f;
        int besti = 0;
        for (int k = tid; k < n; k += 1024) {
            const float d = dist2_ref(pts[(size_t)k * 3], pts[(size_t)k * 3 + 1], pts[(size_t)k * 3 + 2], x1, y1, z1);
            const float d2 = fminf(d, tmp[k]);
            tmp[k] = d2;
            besti = d2 > best ? k : besti;
            best = fmaxf(best, d2);
        }
        Cand c{best, key, besti};
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) c = better(c, shfl_xor(c, s));
        __syncthreads();  // previous iteration's readers are done with s_*
        if (lane == 0) { s_v[wave] = c.v; s_key[wave] = c.key; s_k[wave] = c.k; }
        __syncthreads();
        Cand w{-3.f, 0x7fffffff, 0};
        if (lane < 16) { w.v = s_v[lane]; w.key = s_key[lane]; w.k = s_k[lane]; }
#pragma unroll
        for (int s = 8; s >= 1; s >>= 1) w = better(w, shfl_xor(w, s));
        old = __builtin_amdgcn_readfirstlane(w.k);
        if (tid == 0) out[it] = old;
    }
}

template <int PPT>
static int launch_fps(int b, int n, int m, int log2rb, const float *dataset, float *temp, int *idxs,
                      hipStream_t stream)
{
    const int threads = max(1 << log2rb, 64);
    hipLaunchKernelGGL(fps_kernel<PPT>, dim3(b), dim3(threads), 0, stream, n, m, log2rb, dataset, temp, idxs);
    return launch_status("amc3d_furthest_point_sampling");
}

}  // namespace amc

using namespace amc;

AMC_API int amc3d_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp, int *idxs,
                                          void *stream_)
{
    if (b <= 0 || m <= 0) return 0;  // sampling_gpu.cu:108 `if (m <= 0) return;`
    if (n <= 0 || !dataset || !idxs) return bad_arg("amc3d_furthest_point_sampling: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    // cuda_utils.h:10-14: largest power of two <= n, capped at 1024
    int log2rb = 0;
    while ((2 << log2rb) <= n && log2rb < 10) ++log2rb;
    const int rb = 1 << log2rb;
    const int ppt = (n + rb - 1) / rb;
    if (ppt <= 1) return launch_fps<1>(b, n, m, log2rb, dataset, temp, idxs, stream);
    if (ppt <= 2) return launch_fps<2>(b, n, m, log2rb, dataset, temp, idxs, stream);
    if (ppt <= 4) return launch_fps<4>(b, n, m, log2rb, dataset, temp, idxs, stream);
    if (ppt <= 6) return launch_fps<6>(b, n, m, log2rb, dataset, temp, idxs, stream);
    if (ppt <= 8) return launch_fps<8>(b, n, m, log2rb, dataset, temp, idxs, stream);
    if (ppt <= 12) return launch_fps<12>(b, n, m, log2rb, dataset, temp, idxs, stream);
    if (ppt <= 16) return launch_fps<16>(b, n, m, log2rb, dataset, temp, idxs, stream);
    if (ppt <= 24) return launch_fps<24>(b, n, m, log2rb, dataset, temp, idxs, stream);
    // 24 points x 4 VGPRs is what fits the 128-VGPR budget of a 1024-thread workgroup
    if (!temp) return bad_arg("amc3d_furthest_point_sampling: n > 24576 needs the temp buffer");
    hipLaunchKernelGGL(fps_kernel_large, dim3(b), dim3(1024), 0, stream, n, m, dataset, temp, idxs);
    return launch_status("amc3d_furthest_point_sampling");
}
