// Optimizer tail of the contrastive step: clip_grad_norm_ + AdamW on the adapter parameters
// (scripts/train_contrast.py:453-465, AdamW(lr=2e-4, eps=1e-6, betas=(0.9,0.999), wd=0.01) :621-626).
// Two launches per tensor, no host sync and no atomics: (1) 256 per-block partial sums of g^2,
// (2) the update kernel, where every block re-adds all partials in a fixed order (deterministic
// total norm), derives the clip coefficient on the device and applies torch's AdamW update.
// The update also refreshes the `shadow` copy (bf16 / GEMM row stride) the next forward reads.
#include <math.h>

#include "common.h"
#include "kernels.h"

namespace p2t {

constexpr int kNormBlocks = 256;

__global__ void __launch_bounds__(256) sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            float v[4];
            load4(g + i, v);
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        } else {
            for (int j = 0; j < 4 && i + j < n; ++j) s += g[i + j] * g[i + j];
        }
    }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

template <typename TS>
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, TS* __restrict__ shadow, int64_t cols,
                                                    int64_t shadow_ld, const float* __restrict__ partial, int n_partial,
                                                    float decay, float omb1, float beta2, float omb2, float eps, float step,
                                                    float bc2_sqrt, float max_norm, float* __restrict__ grad_norm_out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[i];
    s = block_sum<4>(s, red);
    const float total = sqrtf(s);
    if (blockIdx.x == 0 && threadIdx.x == 0 && grad_norm_out) grad_norm_out[0] = total;
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total + 1e-6), clamped to 1
    const float coef = fminf(max_norm / (total + 1e-6f), 1.0f);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * coef;
        float pi = p[i] * decay;                                          // param.mul_(1 - lr * wd)
        const float mi = m[i] + (gi - m[i]) * omb1;                       // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * beta2 + omb2 * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= step * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (shadow) shadow[(i / cols) * shadow_ld + (i % cols)] = from_f32<TS>(pi);
    }
}

}  // namespace p2t

using namespace p2t;

extern "C" int p2t_clip_adamw_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                                   float* const* exp_avg_sq, const int64_t* numel, void* const* shadow, const int64_t* cols,
                                   const int64_t* shadow_ld, int shadow_dtype, int step, double lr, double beta1, double beta2,
                                   double eps, double weight_decay, double max_norm, float* grad_norm_out, float* scratch,
                                   p2t_stream stream) {
    P2T_REQUIRE(n_tensors > 0 && n_tensors <= 64 && params && grads && exp_avg && exp_avg_sq && numel && scratch && step >= 1,
                "p2t_clip_adamw_step: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    for (int t = 0; t < n_tensors; ++t) {
        sumsq_partial_kernel<<<kNormBlocks, 256, 0, s>>>(grads[t], numel[t], scratch + (int64_t)t * kNormBlocks);
        P2T_LAUNCH_CHECK();
    }
    // scalar prep in double, as torch.optim.AdamW does in Python
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float decay = (float)(1.0 - lr * weight_decay), omb1 = (float)(1.0 - beta1), b2 = (float)beta2;
    const float omb2 = (float)(1.0 - beta2), epsf = (float)eps, stepsz = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float mn = (max_norm > 0.0 && max_norm < 1e30) ? (float)max_norm : INFINITY;
    for (int t = 0; t < n_tensors; ++t) {
        const int64_t n = numel[t];
        int grid = (int)ceil_div(n, 256 * 8);
        grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
        void* sh = shadow ? shadow[t] : nullptr;
        const int64_t c = (sh && cols) ? cols[t] : 1, ld = (sh && shadow_ld) ? shadow_ld[t] : 1;
        float* gn = t == 0 ? grad_norm_out : nullptr;
        if (sh && shadow_dtype == P2T_BF16)
            adamw_kernel<bf16_t><<<grid, 256, 0, s>>>(params[t], grads[t], exp_avg[t], exp_avg_sq[t], n, (bf16_t*)sh, c, ld, scratch,
                                                      n_tensors * kNormBlocks, decay, omb1, b2, omb2, epsf, stepsz, bc2_sqrt, mn, gn);
        else
            adamw_kernel<float><<<grid, 256, 0, s>>>(params[t], grads[t], exp_avg[t], exp_avg_sq[t], n, (float*)sh, c, ld, scratch,
                                                     n_tensors * kNormBlocks, decay, omb1, b2, omb2, epsf, stepsz, bc2_sqrt, mn, gn);
        P2T_LAUNCH_CHECK();
    }
    return P2T_OK;
}
