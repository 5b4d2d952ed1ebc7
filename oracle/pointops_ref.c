/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.
 *
 * Plain-C CPU restatement of the reference's native point operators (the
 * reference ships them as CUDA only and has no CPU path).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Every function follows the sequential semantics of ONE thread (or one
 * thread block, for FPS) of the reference kernel it cites; the outer loops that
 * the reference distributes over the grid are independent and are run under
 * OpenMP here.  Distance expressions are evaluated exactly as written in the
 * reference source, left to right, in fp32 and WITHOUT fused multiply-add
 * (build with -ffp-contract=off): ((dx*dx + dy*dy) + dz*dz).
 *
 * Parity status: the reference holds no golden vectors for these operators
 * (SURVEY.md section 4), and its CUDA sources cannot be built in this image
 * (no nvcc).  The restatement is pinned by (a) independent brute-force
 * formulations in tests/test_oracle_ops.py and (b) end-to-end fixtures made by
 * running the reference's own Python layer on top of this library
 * (oracle/gen_golden.py -> tests/golden/).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <omp.h>

#define REF_API __attribute__((visibility("default")))

/* number of OpenMP threads the restatement uses (bench.py's cpu_baseline states it) */
REF_API void ref_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
REF_API int ref_get_threads(void) { return omp_get_max_threads(); }

/* ------------------------------------------------------------------------ *
 * ball query -- openpoints/cpp/pointnet2_batch/src/ball_query_gpu.cu:15-51
 * first `nsample` in-radius support indices in ascending index order; on the
 * first hit all slots are pre-filled with that index; rows without a hit keep
 * whatever the caller put there (the Python wrapper zero-fills, group.py:194).
 * ------------------------------------------------------------------------ */
REF_API void ref_ball_query(int b, int n, int m, float radius, int nsample,
                            const float *new_xyz, const float *xyz, int *idx)
{
    const float radius2 = radius * radius;
#pragma omp parallel for schedule(static)
    for (long q = 0; q < (long)b * m; ++q) {
        const int bs = (int)(q / m);
        const float *c = new_xyz + q * 3;
        const float *s = xyz + (long)bs * n * 3;
        int *out = idx + q * nsample;
        const float new_x = c[0], new_y = c[1], new_z = c[2];
        int cnt = 0;
        for (int k = 0; k < n; ++k) {
            const float x = s[k * 3 + 0], y = s[k * 3 + 1], z = s[k * 3 + 2];
            const float d2 = (new_x - x) * (new_x - x) + (new_y - y) * (new_y - y) +
                             (new_z - z) * (new_z - z);
            if (d2 < radius2) {
                if (cnt == 0)
                    for (int l = 0; l < nsample; ++l) out[l] = k;
                out[cnt] = k;
                ++cnt;
                if (cnt >= nsample) break;
            }
        }
    }
}

/* ------------------------------------------------------------------------ *
 * grouping -- openpoints/cpp/pointnet2_batch/src/group_points_gpu.cu:53-72
 * out[b,c,p,s] = points[b,c,idx[b,p,s]]
 * ------------------------------------------------------------------------ */
REF_API void ref_group_points(int b, int c, int n, int npoints, int nsample,
                              const float *points, const int *idx, float *out)
{
#pragma omp parallel for schedule(static)
    for (long bc = 0; bc < (long)b * c; ++bc) {
        const int bs = (int)(bc / c);
        const float *src = points + bc * n;
        const int *ii = idx + (long)bs * npoints * nsample;
        float *dst = out + bc * (long)npoints * nsample;
        for (long t = 0; t < (long)npoints * nsample; ++t) dst[t] = src[ii[t]];
    }
}

/* group_points_gpu.cu:14-31 -- the reference scatters with atomicAdd, i.e. in an
 * unspecified order; this restatement adds in ascending (p,s) order. */
REF_API void ref_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                   const float *grad_out, const int *idx, float *grad_points)
{
#pragma omp parallel for schedule(static)
    for (long bc = 0; bc < (long)b * c; ++bc) {
        const int bs = (int)(bc / c);
        const float *g = grad_out + bc * (long)npoints * nsample;
        const int *ii = idx + (long)bs * npoints * nsample;
        float *dst = grad_points + bc * n;
        for (long t = 0; t < (long)npoints * nsample; ++t) dst[ii[t]] += g[t];
    }
}

/* gather -- openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu:15-31, 53-70 */
REF_API void ref_gather_points(int b, int c, int n, int npoints,
                               const float *points, const int *idx, float *out)
{
#pragma omp parallel for schedule(static)
    for (long bc = 0; bc < (long)b * c; ++bc) {
        const int bs = (int)(bc / c);
        for (int p = 0; p < npoints; ++p)
            out[bc * npoints + p] = points[bc * n + idx[(long)bs * npoints + p]];
    }
}

REF_API void ref_gather_points_grad(int b, int c, int n, int npoints,
                                    const float *grad_out, const int *idx, float *grad_points)
{
#pragma omp parallel for schedule(static)
    for (long bc = 0; bc < (long)b * c; ++bc) {
        const int bs = (int)(bc / c);
        for (int p = 0; p < npoints; ++p)
            grad_points[bc * n + idx[(long)bs * npoints + p]] += grad_out[bc * npoints + p];
    }
}

/* ------------------------------------------------------------------------ *
 * furthest point sampling
 *   kernel    openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu:100-216
 *   __update  sampling_gpu.cu:93-98
 *   block     openpoints/cpp/pointnet2_batch/src/cuda_utils.h:10-14
 * One thread block per cloud.  Thread `tid` scans k = tid, tid+block, ... and
 * keeps the first strict maximum; the shared-memory tree then keeps the lower
 * slot on equal values.  Both are restated literally so that ties resolve as
 * in the reference for the block size the reference would launch.
 * ------------------------------------------------------------------------ */
REF_API int ref_fps_block_size(int work_size)
{
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

REF_API void ref_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                         float *temp, int *idxs)
{
    if (m <= 0) return;
    const int block = ref_fps_block_size(n);
#pragma omp parallel for schedule(dynamic, 1)
    for (int bs = 0; bs < b; ++bs) {
        const float *pts = dataset + (long)bs * n * 3;
        float *tmp = temp + (long)bs * n;
        int *out = idxs + (long)bs * m;
        float dists[1024];
        int dists_i[1024];
        int old = 0;
        out[0] = old;
        for (int j = 1; j < m; ++j) {
            const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
            for (int tid = 0; tid < block; ++tid) {
                int besti = 0;
                float best = -1;
                for (int k = tid; k < n; k += block) {
                    const float x2 = pts[k * 3 + 0], y2 = pts[k * 3 + 1], z2 = pts[k * 3 + 2];
                    const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) +
                                    (z2 - z1) * (z2 - z1);
                    const float d2 = d < tmp[k] ? d : tmp[k]; /* min(d, temp[k]) */
                    tmp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best;
                dists_i[tid] = besti;
            }
            for (int s = block >> 1; s >= 1; s >>= 1) {
                for (int tid = 0; tid < s; ++tid) {
                    const float v1 = dists[tid], v2 = dists[tid + s];
                    const int i1 = dists_i[tid], i2 = dists_i[tid + s];
                    dists[tid] = v1 > v2 ? v1 : v2; /* max(v1, v2) */
                    dists_i[tid] = v2 > v1 ? i2 : i1;
                }
            }
            old = dists_i[0];
            out[j] = old;
        }
    }
}

/* ------------------------------------------------------------------------ *
 * three nearest neighbours -- pointnet2_batch/src/interpolate_gpu.cu:16-59
 * best distances are held in double (initial 1e40) and compared with the fp32
 * candidate by strict '<'; output is the squared distance cast back to fp32.
 * ------------------------------------------------------------------------ */
REF_API void ref_three_nn(int b, int n, int m, const float *unknown, const float *known,
                          float *dist2, int *idx)
{
#pragma omp parallel for schedule(static)
    for (long q = 0; q < (long)b * n; ++q) {
        const int bs = (int)(q / n);
        const float *u = unknown + q * 3;
        const float *kn = known + (long)bs * m * 3;
        const float ux = u[0], uy = u[1], uz = u[2];
        double best1 = 1e40, best2 = 1e40, best3 = 1e40;
        int besti1 = 0, besti2 = 0, besti3 = 0;
        for (int k = 0; k < m; ++k) {
            const float x = kn[k * 3 + 0], y = kn[k * 3 + 1], z = kn[k * 3 + 2];
            const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
            if (d < best1) {
                best3 = best2; besti3 = besti2;
                best2 = best1; besti2 = besti1;
                best1 = d; besti1 = k;
            } else if (d < best2) {
                best3 = best2; besti3 = besti2;
                best2 = d; besti2 = k;
            } else if (d < best3) {
                best3 = d; besti3 = k;
            }
        }
        dist2[q * 3 + 0] = (float)best1; dist2[q * 3 + 1] = (float)best2; dist2[q * 3 + 2] = (float)best3;
        idx[q * 3 + 0] = besti1; idx[q * 3 + 1] = besti2; idx[q * 3 + 2] = besti3;
    }
}

/* interpolate_gpu.cu:84-104 -- out = w0*p[i0] + w1*p[i1] + w2*p[i2], left to right */
REF_API void ref_three_interpolate(int b, int c, int m, int n, const float *points,
                                   const int *idx, const float *weight, float *out)
{
#pragma omp parallel for schedule(static)
    for (long bc = 0; bc < (long)b * c; ++bc) {
        const int bs = (int)(bc / c);
        const float *src = points + bc * m;
        const int *ii = idx + (long)bs * n * 3;
        const float *w = weight + (long)bs * n * 3;
        float *dst = out + bc * n;
        for (int p = 0; p < n; ++p)
            dst[p] = w[p * 3 + 0] * src[ii[p * 3 + 0]] + w[p * 3 + 1] * src[ii[p * 3 + 1]] +
                     w[p * 3 + 2] * src[ii[p * 3 + 2]];
    }
}

/* interpolate_gpu.cu:127-149 -- atomicAdd in the reference; ascending p here */
REF_API void ref_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                        const int *idx, const float *weight, float *grad_points)
{
#pragma omp parallel for schedule(static)
    for (long bc = 0; bc < (long)b * c; ++bc) {
        const int bs = (int)(bc / c);
        const float *g = grad_out + bc * n;
        const int *ii = idx + (long)bs * n * 3;
        const float *w = weight + (long)bs * n * 3;
        float *dst = grad_points + bc * m;
        for (int p = 0; p < n; ++p) {
            dst[ii[p * 3 + 0]] += g[p] * w[p * 3 + 0];
            dst[ii[p * 3 + 1]] += g[p] * w[p * 3 + 1];
            dst[ii[p * 3 + 2]] += g[p] * w[p * 3 + 2];
        }
    }
}

/* ------------------------------------------------------------------------ *
 * k nearest neighbours over "offset" segments
 *   kernel     openpoints/cpp/pointops/src/knnquery/knnquery_cuda_kernel.cu:65-108
 *   reheap     :21-36      heap_sort :39-48      get_bt_idx :51-62
 * Per query: max-heap of `nsample` (<= 100) entries initialised (1e10, start);
 * a candidate replaces the root iff d2 < root (strict), then sift-down; the
 * heap is finally heap-sorted into ascending distance.
 * ------------------------------------------------------------------------ */
static void knn_reheap(float *dist, int *idx, int k)
{
    int root = 0;
    int child = root * 2 + 1;
    while (child < k) {
        if (child + 1 < k && dist[child + 1] > dist[child]) child++;
        if (dist[root] > dist[child]) return;
        float tf = dist[root]; dist[root] = dist[child]; dist[child] = tf;
        int ti = idx[root]; idx[root] = idx[child]; idx[child] = ti;
        root = child;
        child = root * 2 + 1;
    }
}

static void knn_heap_sort(float *dist, int *idx, int k)
{
    for (int i = k - 1; i > 0; i--) {
        float tf = dist[0]; dist[0] = dist[i]; dist[i] = tf;
        int ti = idx[0]; idx[0] = idx[i]; idx[i] = ti;
        knn_reheap(dist, idx, i);
    }
}

REF_API void ref_knnquery(int m, int nsample, const float *xyz, const float *new_xyz,
                          const int *offset, const int *new_offset, int *idx, float *dist2)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int pt = 0; pt < m; ++pt) {
        int bt = 0;
        while (!(pt < new_offset[bt])) bt++;
        const int start = bt == 0 ? 0 : offset[bt - 1];
        const int end = offset[bt];
        const float new_x = new_xyz[pt * 3 + 0], new_y = new_xyz[pt * 3 + 1], new_z = new_xyz[pt * 3 + 2];
        float best_dist[100];
        int best_idx[100];
        for (int i = 0; i < nsample; i++) { best_dist[i] = 1e10; best_idx[i] = start; }
        for (int i = start; i < end; i++) {
            const float x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
            const float d2 = (new_x - x) * (new_x - x) + (new_y - y) * (new_y - y) +
                             (new_z - z) * (new_z - z);
            if (d2 < best_dist[0]) {
                best_dist[0] = d2;
                best_idx[0] = i;
                knn_reheap(best_dist, best_idx, nsample);
            }
        }
        knn_heap_sort(best_dist, best_idx, nsample);
        for (int i = 0; i < nsample; i++) {
            idx[(long)pt * nsample + i] = best_idx[i];
            dist2[(long)pt * nsample + i] = best_dist[i];
        }
    }
}
