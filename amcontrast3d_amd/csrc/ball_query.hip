// Ball query for gfx950.
//
// Reference semantics (pointnet2_batch/src/ball_query_gpu.cu:15-51): per query the
// first `nsample` support indices, in ascending index order, whose squared
// distance is < radius^2; remaining slots repeat the first hit; a query without
// a hit yields zeros (caller zero-fill, models/layers/group.py:194).
//
// Mapping: the reference gives one thread a whole O(n) scan with per-hit
// scattered stores.  Here one wavefront owns QPW queries and tests 64
// consecutive support points per step (one per lane): __ballot gives the hit
// mask, mbcnt its rank, so hits land in their ordered slot with no
// serialisation and index order holds by construction.  The support cloud is
// staged through LDS in SoA tiles by the whole workgroup (coalesced dword
// loads of the AoS xyz array, conflict-free ds_read_b32 afterwards) and shared by
// the 4 waves x QPW queries of the block, so each support byte is fetched from
// L2 once per 16 queries.  A wave stops testing a query once it has nsample
// hits; the block leaves the scan when all its queries are full.
#include "common.h"

namespace amc {

constexpr int BQ_TILE = 1024;  // support points per LDS tile
constexpr int BQ_WAVES = 4;
constexpr int BQ_QPW = 4;  // queries per wave

__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(
    int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx)
{
    __shared__ float sx[BQ_TILE], sy[BQ_TILE], sz[BQ_TILE];
    __shared__ int s_active;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bs = blockIdx.y;
    const int q0 = (blockIdx.x * BQ_WAVES + wave) * BQ_QPW;
    const float *S = xyz + (size_t)bs * n * 3;

    float qx[BQ_QPW], qy[BQ_QPW], qz[BQ_QPW];
    int cnt[BQ_QPW], first[BQ_QPW];
#pragma unroll
    for (int i = 0; i < BQ_QPW; ++i) {
        const int q = q0 + i;
        const bool ok = q < m;
        const float *c = new_xyz + ((size_t)bs * m + (ok ? q : 0)) * 3;
        qx[i] = c[0]; qy[i] = c[1]; qz[i] = c[2];
        cnt[i] = ok ? 0 : nsample;  // out-of-range queries are "full" from the start
        first[i] = 0;
    }

    for (int t0 = 0; t0 < n; t0 += BQ_TILE) {
        const int tn = min(BQ_TILE, n - t0);
        __syncthreads();  // previous tile fully consumed
        if (threadIdx.x == 0) s_active = 0;
        // coalesced AoS -> SoA staging: flat dword i of the tile is coord (i%3) of point i/3
        for (int i = threadIdx.x; i < tn * 3; i += BQ_WAVES * 64) {
            const float v = S[(size_t)t0 * 3 + i];
            const int p = i / 3, c = i - p * 3;
            (c == 0 ? sx : c == 1 ? sy : sz)[p] = v;
        }
        __syncthreads();

        bool wave_active = false;
#pragma unroll
        for (int i = 0; i < BQ_QPW; ++i) wave_active |= cnt[i] < nsample;
        if (wave_active) {
            for (int k0 = 0; k0 < tn; k0 += 64) {
                const int kl = k0 + lane;
                const bool valid = kl < tn;
                const float x = sx[valid ? kl : 0], y = sy[valid ? kl : 0], z = sz[valid ? kl : 0];
                bool any_left = false;
#pragma unroll
                for (int i = 0; i < BQ_QPW; ++i) {
                    if (cnt[i] < nsample) {  // wave-uniform
                        const float d2 = dist2_ref(qx[i], qy[i], qz[i], x, y, z);
                        const bool hit = valid && d2 < radius2;
                        const unsigned long long mask = __ballot(hit);
                        if (mask) {
                            const int pos = cnt[i] + mbcnt(mask);
                            if (hit && pos < nsample)
                                idx[((size_t)bs * m + q0 + i) * nsample + pos] = t0 + kl;
                            if (cnt[i] == 0) first[i] = t0 + k0 + (int)__builtin_ctzll(mask);
                            cnt[i] += (int)__popcll(mask);
                        }
                        any_left |= cnt[i] < nsample;
                    }
                }
                if (!any_left) { wave_active = false; break; }
            }
        }
        if (wave_active && lane == 0) s_active = 1;
        __syncthreads();
        if (!s_active) break;  // every query of the block is full (block-uniform)
    }

    // pad the tail of each row with the first hit (zeros when there was none)
#pragma unroll
    for (int i = 0; i < BQ_QPW; ++i) {
        const int q = q0 + i;
        if (q < m) {
            const int have = min(cnt[i], nsample);
            for (int l = have + lane; l < nsample; l += 64)
                idx[((size_t)bs * m + q) * nsample + l] = first[i];
        }
    }
}

// out[b,c,p,s] = points[b,c,idx[b,p,s]] -- group_points_gpu.cu:53-72.
// One thread per (b, p*nsample+s) reads its index once and walks the channels:
// stores are coalesced per channel plane, loads are a gather inside one plane.
__global__ void group_points_kernel(int c, int n, int ps, const float *__restrict__ points,
                                    const int *__restrict__ idx, float *__restrict__ out)
{
    const int bs = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ps) return;
    const int id = idx[(size_t)bs * ps + t];
    const float *src = points + (size_t)bs * c * n + id;
    float *dst = out + (size_t)bs * c * ps + t;
    for (int ch = 0; ch < c; ++ch) dst[(size_t)ch * ps] = src[(size_t)ch * n];
}

// grad_points[b,c,idx[b,p,s]] += grad_out[b,c,p,s] -- group_points_gpu.cu:14-31
__global__ void group_points_grad_kernel(int c, int n, int ps, const float *__restrict__ grad_out,
                                         const int *__restrict__ idx, float *__restrict__ grad_points)
{
    const int bs = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ps) return;
    const int id = idx[(size_t)bs * ps + t];
    const float *src = grad_out + (size_t)bs * c * ps + t;
    float *dst = grad_points + (size_t)bs * c * n + id;
    for (int ch = 0; ch < c; ++ch) atomicAdd(dst + (size_t)ch * n, src[(size_t)ch * ps]);
}

}  // namespace amc

using namespace amc;

namespace amc {
bool grid_search_pays(int b, int n, int m);
size_t grid_search_workspace_bytes(int b, int n, int m);
int ball_query_grid(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx,
                    void *workspace, hipStream_t stream);
}

AMC_API size_t amc3d_grid_search_workspace_bytes(int b, int n_support, int m_queries)
{
    if (b <= 0 || n_support <= 0 || m_queries <= 0) return 0;
    return grid_search_workspace_bytes(b, n_support, m_queries);
}

AMC_API int amc3d_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                             const float *xyz, int *idx, void *workspace, size_t workspace_bytes, void *stream)
{
    if (b <= 0 || m <= 0) return 0;
    if (n < 0 || nsample <= 0 || !new_xyz || !xyz || !idx) return bad_arg("amc3d_ball_query: bad argument");
    if (workspace && n > 0 && nsample <= 64 && radius > 0.f && grid_search_pays(b, n, m) &&
        workspace_bytes >= grid_search_workspace_bytes(b, n, m))
        return ball_query_grid(b, n, m, radius, nsample, new_xyz, xyz, idx, workspace, (hipStream_t)stream);
    dim3 grid(div_up(m, BQ_WAVES * BQ_QPW), b);
    const float radius2 = radius * radius;  // ball_query_gpu.cu:31
    hipLaunchKernelGGL(ball_query_kernel, grid, dim3(BQ_WAVES * 64), 0, (hipStream_t)stream, n, m, radius2,
                       nsample, new_xyz, xyz, idx);
    return launch_status("amc3d_ball_query");
}

AMC_API int amc3d_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                               const int *idx, float *out, void *stream)
{
    const long ps = (long)npoints * nsample;
    if (b <= 0 || c <= 0 || ps <= 0) return 0;
    if (!points || !idx || !out) return bad_arg("amc3d_group_points: null pointer");
    hipLaunchKernelGGL(group_points_kernel, dim3(div_up(ps, 256), b), dim3(256), 0, (hipStream_t)stream, c, n,
                       (int)ps, points, idx, out);
    return launch_status("amc3d_group_points");
}

namespace amc {
int scatter_add_pm(int fan, int b, int c, int n, long entries, const float *grad_out, const int *idx,
                   const float *weight, float *grad_points, float *scratch, hipStream_t stream, const char *what);
}

AMC_API size_t amc3d_scatter_workspace_bytes(int b, int c, int n) { return (size_t)b * c * n * sizeof(float); }

AMC_API int amc3d_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                    const int *idx, float *grad_points, void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    const long ps = (long)npoints * nsample;
    if (b <= 0 || c <= 0 || ps <= 0) return 0;
    if (!grad_out || !idx || !grad_points) return bad_arg("amc3d_group_points_grad: null pointer");
    if (workspace && workspace_bytes >= amc3d_scatter_workspace_bytes(b, c, n) && c >= 8)
        return scatter_add_pm(1, b, c, n, ps, grad_out, idx, nullptr, grad_points, (float *)workspace,
                              (hipStream_t)stream, "amc3d_group_points_grad");
    hipLaunchKernelGGL(group_points_grad_kernel, dim3(div_up(ps, 256), b), dim3(256), 0, (hipStream_t)stream, c,
                       n, (int)ps, grad_out, idx, grad_points);
    return launch_status("amc3d_group_points_grad");
}

// gather == grouping with nsample = 1 (sampling_gpu.cu:15-31, 53-70)
AMC_API int amc3d_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                                float *out, void *stream)
{
    return amc3d_group_points(b, c, n, npoints, 1, points, idx, out, stream);
}

AMC_API int amc3d_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                                     float *grad_points, void *stream)
{
    return amc3d_group_points_grad(b, c, n, npoints, 1, grad_out, idx, grad_points, nullptr, 0, stream);
}
