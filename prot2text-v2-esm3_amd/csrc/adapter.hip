// ModalityAdapter forward / backward  (reference models/modeling_esm2llama_instruct.py:45-68):
//     y = normalize( drop( gelu( fc2( drop( gelu( fc1(x) ) ) ) ) ), p=2, dim=-1 )
// Forward = two GEMMs with the bias+erf-GELU(+dropout) epilogue and a row L2-normalise.
// Backward (the only backward on the contrastive path: both towers are frozen,
// scripts/train_contrast.py:186-187) produces fp32 gradients of fc1/fc2 weight and bias:
//     dg2 = (dy - y <dy, y>) / ||g2||          dz2 = dg2 * mask2 * gelu'(z2)
//     dW2 = dz2^T h1,  db2 = colsum(dz2)        dh1 = dz2 W2
//     dz1 = dh1 * mask1 * gelu'(z1)             dW1 = dz1^T x,  db1 = colsum(dz1)
// The weight-gradient products contract over the token axis; they run on the same NT GEMM kernel
// after an LDS-tiled transpose of the two operands (token axis made contiguous).
#include "common.h"
#include "epilogue.h"
#include "kernels.h"

namespace p2t {

constexpr uint64_t kSeed2 = 0x632BE59BD9B4E019ull;      // decorrelates the second dropout mask

// one wave per row: dot = <dy, y>, then dz2 (two passes over the row, second one L1/L2 resident)
template <typename T>
__global__ void __launch_bounds__(256) adapter_dz2_kernel(const T* __restrict__ g2, const T* __restrict__ z2,
                                                          const float* __restrict__ inv_norm, const float* __restrict__ dy,
                                                          T* __restrict__ dz2, int64_t ld, int64_t M, int D, float drop_p,
                                                          float drop_scale, uint64_t drop_seed) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float inv = inv_norm[m];
    const T* gr = g2 + m * ld;
    const float* dr = dy + m * (int64_t)D;
    float dot = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        float g[4], d4[4];
        load4(gr + c, g);
        load4(dr + c, d4);
        dot += (g[0] * d4[0] + g[1] * d4[1] + g[2] * d4[2] + g[3] * d4[3]) * inv;
    }
    dot = wave_sum(dot);
    for (int c = lane * 4; c < ld; c += 256) {
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            float g[4], d4[4], z[4];
            load4(gr + c, g);
            load4(dr + c, d4);
            load4(z2 + m * ld + c, z);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = (d4[j] - g[j] * inv * dot) * inv;          // d / d g2 (post-dropout activation)
                if (drop_p > 0.f) v = dropout_keep(drop_seed, m * (int64_t)D + c + j, drop_p) ? v * drop_scale : 0.f;
                r[j] = v * gelu_erf_grad(z[j]);
            }
        }
        store4(dz2 + m * ld + c, r);
    }
}

int launch_adapter_dz2(const void* g2, const void* z2, const float* inv_norm, const float* dy, void* dz2, int64_t ld, int64_t M,
                       int D, int dtype, float drop_p, uint64_t drop_seed, hipStream_t s) {
    const dim3 grid((unsigned)ceil_div(M, 4));
    const float sc = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    if (dtype == P2T_BF16)
        adapter_dz2_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)g2, (const bf16_t*)z2, inv_norm, dy, (bf16_t*)dz2, ld, M, D, drop_p, sc, drop_seed);
    else
        adapter_dz2_kernel<float><<<grid, 256, 0, s>>>((const float*)g2, (const float*)z2, inv_norm, dy, (float*)dz2, ld, M, D, drop_p, sc, drop_seed);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

}  // namespace p2t

using namespace p2t;

static int check_adapter(const p2t_adapter_config* c) {
    P2T_REQUIRE(c && c->input_dim > 0 && c->intermediate_dim > 0 && c->output_dim > 0, "adapter: bad config");
    P2T_REQUIRE(c->input_dim % 16 == 0 && c->intermediate_dim % 16 == 0 && c->output_dim % 16 == 0,
                "adapter: input_dim, intermediate_dim and output_dim must be multiples of 16");
    P2T_REQUIRE(c->dropout_p >= 0.f && c->dropout_p < 1.f, "adapter: dropout_p out of range");
    return P2T_OK;
}

extern "C" int p2t_adapter_forward(const p2t_adapter_config* cfg, const p2t_adapter_weights* w, const void* x, int64_t ld_x,
                                   int64_t M, void* y, const p2t_adapter_saved* save, p2t_stream stream) {
    P2T_TRY(check_adapter(cfg));
    P2T_REQUIRE(w && x && y && save && save->h1 && save->g2, "p2t_adapter_forward: h1 and g2 buffers are required");
    hipStream_t s = (hipStream_t)stream;
    const int dt = cfg->dtype;
    const int64_t I = cfg->intermediate_dim, O = cfg->output_dim, K1 = round_up(cfg->input_dim, 64);
    const int64_t ld1 = round_up(I, 64), ld2 = round_up(O, 64);
    P2T_REQUIRE(ld_x >= K1, "p2t_adapter_forward: ld_x=%lld must cover input_dim rounded up to 64 (zero padded)", (long long)ld_x);
    GemmArgs a{x, ld_x, w->fc1_w, K1, w->fc1_b, save->h1, ld1, save->z1, M, I, K1, dt, dt, P2T_EPI_GELU, 0, -1, -1,
               cfg->dropout_p, cfg->dropout_seed, 0};
    P2T_TRY(gemm_nt(a, s));
    GemmArgs b{save->h1, ld1, w->fc2_w, ld1, w->fc2_b, save->g2, ld2, save->z2, M, O, ld1, dt, dt, P2T_EPI_GELU, 0, -1, -1,
               cfg->dropout_p, cfg->dropout_seed ^ kSeed2, 0};
    P2T_TRY(gemm_nt(b, s));
    return launch_l2norm(save->g2, dt, ld2, y, dt, ld2, save->inv_norm, M, O, 1e-12f, s);
}

extern "C" size_t p2t_adapter_backward_workspace_bytes(const p2t_adapter_config* cfg, int64_t M) {
    if (!cfg || M < 0) return 0;
    const size_t e = dtype_size(cfg->dtype);
    const int64_t Mp = round_up(M, 64), I = cfg->intermediate_dim, O = cfg->output_dim, X = cfg->input_dim;
    const int64_t ld1 = round_up(I, 64), ld2 = round_up(O, 64);
    size_t n = 0;
    n += (size_t)M * ld2 * e + 256;        // dz2
    n += (size_t)O * Mp * e + 256;         // dz2^T
    n += (size_t)I * Mp * e + 256;         // h1^T  (reused for dz1^T)
    n += (size_t)I * ld2 * e + 256;        // W2^T
    n += (size_t)M * ld1 * e + 256;        // dz1
    n += (size_t)X * Mp * e + 256;         // x^T
    n += colsum_scratch_bytes(O > I ? O : I) + 256;
    return n + 1024;
}

extern "C" int p2t_adapter_backward(const p2t_adapter_config* cfg, const p2t_adapter_weights* w, const void* x, int64_t ld_x,
                                    int64_t M, const p2t_adapter_saved* saved, const float* dy, float* d_fc1_w, float* d_fc1_b,
                                    float* d_fc2_w, float* d_fc2_b, int accumulate, void* workspace, size_t workspace_bytes,
                                    p2t_stream stream) {
    P2T_TRY(check_adapter(cfg));
    P2T_REQUIRE(w && x && saved && saved->z1 && saved->h1 && saved->z2 && saved->g2 && saved->inv_norm && dy && d_fc1_w &&
                    d_fc1_b && d_fc2_w && d_fc2_b && workspace,
                "p2t_adapter_backward: null argument (all saved activations are required)");
    P2T_REQUIRE(workspace_bytes >= p2t_adapter_backward_workspace_bytes(cfg, M), "p2t_adapter_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int dt = cfg->dtype;
    const size_t e = dtype_size(dt);
    const int64_t Mp = round_up(M, 64), I = cfg->intermediate_dim, O = cfg->output_dim, X = cfg->input_dim;
    const int64_t ld1 = round_up(I, 64), ld2 = round_up(O, 64);
    Arena ar(workspace, workspace_bytes);
    void* dz2 = ar.take((size_t)M * ld2 * e);
    void* dz2T = ar.take((size_t)O * Mp * e);
    void* h1T = ar.take((size_t)I * Mp * e);
    void* w2T = ar.take((size_t)I * ld2 * e);
    void* dz1 = ar.take((size_t)M * ld1 * e);
    void* xT = ar.take((size_t)X * Mp * e);
    float* cs_scratch = (float*)ar.take(colsum_scratch_bytes(O > I ? O : I));
    P2T_REQUIRE(!ar.overflow, "p2t_adapter_backward: workspace overflow");

    P2T_TRY(launch_adapter_dz2(saved->g2, saved->z2, saved->inv_norm, dy, dz2, ld2, M, (int)O, dt, cfg->dropout_p,
                               cfg->dropout_seed ^ kSeed2, s));
    P2T_TRY(launch_colsum(dz2, dt, M, O, ld2, d_fc2_b, accumulate, cs_scratch, s));
    P2T_TRY(p2t_transpose(dz2, M, O, ld2, dz2T, Mp, dt, stream));
    P2T_TRY(p2t_transpose(saved->h1, M, I, ld1, h1T, Mp, dt, stream));
    {   // dW2 [O, I] = dz2^T [O, M] . (h1^T [I, M])^T
        GemmArgs g{dz2T, Mp, h1T, Mp, nullptr, d_fc2_w, I, nullptr, O, I, Mp, dt, P2T_F32, P2T_EPI_STORE_F32, accumulate, -1, -1, 0.f, 0, 0};
        P2T_TRY(gemm_nt(g, s));
    }
    P2T_TRY(p2t_transpose(w->fc2_w, O, I, ld1, w2T, ld2, dt, stream));
    {   // dz1 [M, I] = (dz2 [M, O] . (W2^T [I, O])^T) * mask1 * gelu'(z1)
        GemmArgs g{dz2, ld2, w2T, ld2, nullptr, dz1, ld1, saved->z1, M, I, ld2, dt, dt, P2T_EPI_GELU_BWD, 0, -1, -1,
                   cfg->dropout_p, cfg->dropout_seed, 0};
        P2T_TRY(gemm_nt(g, s));
    }
    P2T_TRY(launch_colsum(dz1, dt, M, I, ld1, d_fc1_b, accumulate, cs_scratch, s));
    void* dz1T = h1T;   // h1^T is dead after dW2
    P2T_TRY(p2t_transpose(dz1, M, I, ld1, dz1T, Mp, dt, stream));
    P2T_TRY(p2t_transpose(x, M, X, ld_x, xT, Mp, dt, stream));
    {   // dW1 [I, X] = dz1^T [I, M] . (x^T [X, M])^T
        GemmArgs g{dz1T, Mp, xT, Mp, nullptr, d_fc1_w, X, nullptr, I, X, Mp, dt, P2T_F32, P2T_EPI_STORE_F32, accumulate, -1, -1, 0.f, 0, 0};
        P2T_TRY(gemm_nt(g, s));
    }
    return P2T_OK;
}
