// k-nearest-neighbour query over "offset" segments for gfx950.
//
// Reference: pointops/src/knnquery/knnquery_cuda_kernel.cu:65-108 -- one thread
// per query scans its whole segment in index order, keeping a max-heap of
// `nsample` (<= 100) entries in per-thread scratch: a candidate replaces the
// root iff d2 < root (strict), then `reheap` (:21-36) sifts down; `heap_sort`
// (:39-48) finally emits ascending distance.  How equal distances are ordered
// (and which of several equal maxima is evicted) is a property of that heap
// history, so "bit-exact indices" means reproducing it.
//
// knn_exact_kernel: one WAVEFRONT per query replays exactly that heap, but scans
// 64 candidates per step: lanes compute d2 for 64 consecutive indices, a ballot
// selects those below the current root, and only those are fed -- in ascending
// index order, re-checked against the updated root -- to the heap, which lives
// in LDS.  A candidate that fails `d2 < root` at ballot time can never pass
// later (the root only decreases), so the heap sees the same insert sequence as
// the reference's sequential scan.  The 16 waves of a workgroup share LDS tiles
// of the support cloud.
#include "common.h"

namespace amc {

constexpr int KNN_TILE = 1024;
constexpr int KNN_WAVES = 16;
constexpr int KNN_MAXK = 100;  // reference: float best_dist[100] (knnquery_cuda_kernel.cu:86)

// knnquery_cuda_kernel.cu:21-36
__device__ __forceinline__ void reheap(float *dist, int *idx, int k)
{
    int root = 0;
    int child = 1;
    while (child < k) {
        if (child + 1 < k && dist[child + 1] > dist[child]) child++;
        if (dist[root] > dist[child]) return;
        const float tf = dist[root]; dist[root] = dist[child]; dist[child] = tf;
        const int ti = idx[root]; idx[root] = idx[child]; idx[child] = ti;
        root = child;
        child = root * 2 + 1;
    }
}

// queries: either all of 0..m-1 (qlist == nullptr) or the first *qcount entries of qlist
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_exact_kernel(
    int m, int nsample, int nbatch, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const int *__restrict__ offset, const int *__restrict__ new_offset, int *__restrict__ idx,
    float *__restrict__ dist2, const int *__restrict__ qlist, const int *__restrict__ qcount)
{
    __shared__ float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
    __shared__ float hd[KNN_WAVES][KNN_MAXK];
    __shared__ int hi[KNN_WAVES][KNN_MAXK];
    __shared__ int s_lo[KNN_WAVES], s_hi[KNN_WAVES];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = qlist ? *qcount : m;

    for (int base = blockIdx.x * KNN_WAVES; base < total; base += gridDim.x * KNN_WAVES) {
        const int qi = base + wave;
        const bool live = qi < total;
        const int pt = live ? (qlist ? qlist[qi] : qi) : 0;
        // knnquery_cuda_kernel.cu:51-62,74-80: segment of this query
        int start = 0, end = 0;
        if (live) {
            int bt = 0;
            while (bt < nbatch - 1 && !(pt < new_offset[bt])) bt++;
            start = bt == 0 ? 0 : offset[bt - 1];
            end = offset[bt];
        }
        const float qx = new_xyz[(size_t)pt * 3], qy = new_xyz[(size_t)pt * 3 + 1], qz = new_xyz[(size_t)pt * 3 + 2];
        float *mydist = hd[wave];
        int *myidx = hi[wave];
        for (int i = lane; i < nsample; i += 64) { mydist[i] = 1e10f; myidx[i] = start; }
        float root = 1e10f;

        __syncthreads();  // previous round's readers of s_lo/s_hi and tiles are done
        if (lane == 0) { s_lo[wave] = live ? start : 0x7fffffff; s_hi[wave] = live ? end : 0; }
        __syncthreads();
        int lo = 0x7fffffff, hi_ = 0;
        for (int w = 0; w < KNN_WAVES; ++w) { lo = min(lo, s_lo[w]); hi_ = max(hi_, s_hi[w]); }

        for (int t0 = lo; t0 < hi_; t0 += KNN_TILE) {
            const int tn = min(KNN_TILE, hi_ - t0);
            __syncthreads();
            for (int i = threadIdx.x; i < tn * 3; i += KNN_WAVES * 64) {
                const float v = xyz[(size_t)t0 * 3 + i];
                const int p = i / 3, c = i - p * 3;
                (c == 0 ? sx : c == 1 ? sy : sz)[p] = v;
            }
            __syncthreads();
            if (!live || t0 >= end || t0 + tn <= start) continue;  // wave-uniform
            for (int k0 = 0; k0 < tn; k0 += 64) {
                const int kl = k0 + lane, gi = t0 + kl;
                const bool valid = kl < tn && gi >= start && gi < end;
                const float d2 = dist2_ref(qx, qy, qz, sx[kl < tn ? kl : 0], sy[kl < tn ? kl : 0], sz[kl < tn ? kl : 0]);
                unsigned long long mask = __ballot(valid && d2 < root);
                while (mask) {
                    const int bpos = (int)__builtin_ctzll(mask);
                    mask &= mask - 1;
                    const float cd = __shfl(d2, bpos, 64);
                    if (cd < root) {  // knnquery_cuda_kernel.cu:97-101
                        mydist[0] = cd;
                        myidx[0] = t0 + k0 + bpos;
                        reheap(mydist, myidx, nsample);
                        root = mydist[0];
                    }
                }
            }
        }
        if (live) {
            // knnquery_cuda_kernel.cu:39-48
            for (int i = nsample - 1; i > 0; i--) {
                const float tf = mydist[0]; mydist[0] = mydist[i]; mydist[i] = tf;
                const int ti = myidx[0]; myidx[0] = myidx[i]; myidx[i] = ti;
                reheap(mydist, myidx, i);
            }
            for (int i = lane; i < nsample; i += 64) {
                idx[(size_t)pt * nsample + i] = myidx[i];
                dist2[(size_t)pt * nsample + i] = mydist[i];
            }
        }
    }
}

}  // namespace amc

using namespace amc;

AMC_API size_t amc3d_knnquery_workspace_bytes(int n, int m, int nsample, int nbatch)
{
    (void)n; (void)m; (void)nsample; (void)nbatch;
    return 256;
}

AMC_API int amc3d_knnquery(int m, int nsample, int n, int nbatch, const float *xyz, const float *new_xyz,
                           const int *offset, const int *new_offset, int *idx, float *dist2, void *workspace,
                           size_t workspace_bytes, void *stream)
{
    (void)workspace; (void)workspace_bytes; (void)n;
    if (m <= 0) return 0;
    if (nsample <= 0 || nsample > KNN_MAXK) return bad_arg("amc3d_knnquery: nsample must be in 1..100");
    if (nbatch <= 0 || !xyz || !new_xyz || !offset || !new_offset || !idx || !dist2)
        return bad_arg("amc3d_knnquery: bad argument");
    const int blocks = min(div_up(m, KNN_WAVES), 256 * 64);
    hipLaunchKernelGGL(knn_exact_kernel, dim3(blocks), dim3(KNN_WAVES * 64), 0, (hipStream_t)stream, m, nsample,
                       nbatch, xyz, new_xyz, offset, new_offset, idx, dist2, (const int *)nullptr,
                       (const int *)nullptr);
    return launch_status("amc3d_knnquery");
}
