// Gradient-norm clipping + AdamW over every parameter tensor of the model as two launches (three for large models; gfx950).
//
// Reference: the trainer's update is  torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.grad_norm_clip, norm_type=2)
// followed by  optimizer.step()  with torch.optim.AdamW (examples/segmentation/main_AA.py:586-592, optimizer from
// openpoints/optim/optim_factory.py:160-230, cfgs/s3dis/default.yaml:64-69).  PointNeXt-S has 0.8 M parameters in ~150 tensors:
// as torch kernels that is a multi-tensor norm, its clean-up, a reciprocal / clamp / multiply chain and three fused-AdamW
// launches (the multi-tensor argument block holds ~50 tensors) -- 0.11 ms of a 8.3 ms step, all of it launch overhead, and
// serial: it sits between the backward of one step and the forward of the next.
//
// Here the tensors are described once by a table in device memory (pointers, sizes, per-tensor weight decay and learning
// rate) and a block map (which 1024-element chunk of which tensor a workgroup handles), so one launch covers them all:
//   adamw_norm_kernel    per-chunk sum of squares of the gradients (double), advances the tensors' step counters
//   adamw_update_kernel  every workgroup sums the chunk partials in the same fixed order -> total norm -> clip coefficient
//                        (max_norm / (norm + 1e-6), capped at 1: clip_grad_norm_'s formula), then AdamW in the order torch's
//                        fused kernel evaluates it (decoupled decay, lerp form of the first moment, bias corrections from
//                        the device step counters).  The clipped gradient is used, not written back.
// Deterministic (no atomics), capture-safe (no allocation, no sync; the learning rate lives in the table).
#include "common.h"

namespace amc {

struct AdamwTensor {  // 56 bytes; mirrored by amc3d_adamw_tensor in include/amc3d.h
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    float *step;  // this parameter's step count (torch keeps one per parameter: a parameter without a gradient skips a step)
    long long numel;
    float weight_decay;
    float lr;
};

constexpr int ADAMW_CHUNK = 1024;  // elements per workgroup: 256 threads x 4

__global__ __launch_bounds__(256) void adamw_norm_kernel(const AdamwTensor *__restrict__ table, const int *__restrict__ block_map,
                                                         double *__restrict__ partial)
{
    __shared__ double s_sum[4];
    const AdamwTensor t = table[block_map[2 * blockIdx.x]];
    const long long e0 = (long long)block_map[2 * blockIdx.x + 1] * ADAMW_CHUNK;
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long e = e0 + threadIdx.x + u * 256;
        if (e < t.numel) { const float g = t.grad[e]; acc += (double)g * (double)g; }
    }
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
        if (block_map[2 * blockIdx.x + 1] == 0) t.step[0] += 1.f;  // first chunk of a tensor: the update kernel reads the new count
    }
}

// Large models (PointNeXt-XL: 41 M parameters = 40 k chunks): every workgroup of the update kernel summing every chunk partial is
// nblocks^2 reads (12.8 GB from L2, 1.09 ms measured).  One launch in between folds groups of ADAMW_FOLD partials into the first
// slot of their group, in place and in a fixed order; the update kernel then reads one value per group.
constexpr int ADAMW_FOLD = 1024;

__global__ __launch_bounds__(256) void adamw_fold_kernel(int nblocks, double *__restrict__ partial)
{
    __shared__ double s_sum[4];
    const int i0 = blockIdx.x * ADAMW_FOLD;
    double acc = 0.0;
    for (int i = i0 + threadIdx.x; i < min(i0 + ADAMW_FOLD, nblocks); i += 256) acc += partial[i];
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[i0] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
}

__global__ __launch_bounds__(256) void adamw_update_kernel(const AdamwTensor *__restrict__ table, const int *__restrict__ block_map,
                                                           int nblocks, int stride, const double *__restrict__ partial,
                                                           double beta1, double beta2, float eps, float max_norm,
                                                           float *__restrict__ total_norm)
{
    __shared__ double s_sum[4];
    // total gradient norm: every workgroup adds the same partials in the same order (thread-strided, then a fixed tree)
    double acc = 0.0;
    for (int i = threadIdx.x * stride; i < nblocks; i += 256 * stride) acc += partial[i];
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    const float norm = (float)sqrt(((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3]);
    if (blockIdx.x == 0 && threadIdx.x == 0 && total_norm) total_norm[0] = norm;
    float clip = 1.f;
    if (max_norm > 0.f) clip = fminf(max_norm / (norm + 1e-6f), 1.f);  // clip_grad_norm_: coefficient clamped to 1

    const AdamwTensor t = table[block_map[2 * blockIdx.x]];
    const long long e0 = (long long)block_map[2 * blockIdx.x + 1] * ADAMW_CHUNK;
    // hyper-parameters are doubles on the host (python floats): 1 - beta, beta^step and lr * weight_decay are formed in double
    // and rounded once, as torch's fused kernel does (1 - 0.999f is 1.3e-5 away from 0.001)
    const double st = (double)t.step[0];
    const float bc1 = (float)(1.0 - pow(beta1, st)), bc2 = (float)(1.0 - pow(beta2, st));
    const float step_size = t.lr / bc1, bc2_sqrt = sqrtf(bc2);
    const float lr_wd = (float)((double)t.lr * (double)t.weight_decay);
    const float b2 = (float)beta2, omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long e = e0 + threadIdx.x + u * 256;
        if (e < t.numel) {
            const float g = t.grad[e] * clip;
            float p = t.param[e];
            p -= lr_wd * p;
            float m = t.exp_avg[e], v = t.exp_avg_sq[e];
            m = m + omb1 * (g - m);
            v = b2 * v + omb2 * g * g;
            const float denom = sqrtf(v) / bc2_sqrt + eps;
            p -= step_size * (m / denom);
            t.param[e] = p;
            t.exp_avg[e] = m;
            t.exp_avg_sq[e] = v;
        }
    }
}

}  // namespace amc

using namespace amc;

AMC_API int amc3d_adamw_chunk(void) { return ADAMW_CHUNK; }

AMC_API int amc3d_adamw_step(const void *table, const int *block_map, int nblocks, double beta1, double beta2, float eps,
                             float max_grad_norm, double *partial, float *total_norm, void *stream_)
{
    if (nblocks <= 0) return 0;
    if (!table || !block_map || !partial || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps > 0.f))
        return bad_arg("amc3d_adamw_step: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    static_assert(sizeof(AdamwTensor) == 56, "amc3d_adamw_tensor layout");
    hipLaunchKernelGGL(adamw_norm_kernel, dim3(nblocks), dim3(256), 0, stream, (const AdamwTensor *)table, block_map, partial);
    const int stride = nblocks > 2 * ADAMW_FOLD ? ADAMW_FOLD : 1;
    if (stride > 1) hipLaunchKernelGGL(adamw_fold_kernel, dim3(div_up(nblocks, ADAMW_FOLD)), dim3(256), 0, stream, nblocks, partial);
    hipLaunchKernelGGL(adamw_update_kernel, dim3(nblocks), dim3(256), 0, stream, (const AdamwTensor *)table, block_map, nblocks,
                       stride, (const double *)partial, beta1, beta2, eps, max_grad_norm, total_norm);
    return launch_status("amc3d_adamw_step");
}
