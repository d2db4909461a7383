// Exact-fp32 attention (parity mode and fallback): one wavefront per query row, online softmax
// over 64-key chunks.  Phase 1 puts keys on lanes (scores, running max / sum), phase 2 puts the
// head channels on lanes and accumulates P.V from V ([keys][dp] row-major: coalesced across lanes).
//   ESM:   bidirectional + key padding, scale 1.0 (q pre-scaled), modeling_esm.py:292-317
//   Llama: causal + key padding, GQA, fp32 softmax, modeling_llama.py:191-213
// Rows with no visible key produce zeros (the reference would average all keys; not reachable with
// the right-padded batch contract).
#include "common.h"
#include "kernels.h"

namespace p2t {

template <typename T>
__global__ void __launch_bounds__(256) attn_simple_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                          const T* __restrict__ v, const uint8_t* __restrict__ key_mask,
                                                          const int32_t* __restrict__ kv_end, T* __restrict__ out,
                                                          int64_t ld_out, int seq, int nh, int nkv, int d, int dp,
                                                          float scale, int causal, int out_cols, float* __restrict__ lse) {
    __shared__ float s_q[4][128];
    __shared__ float s_p[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
    if (i >= seq) return;
    const int hk = h / (nh / nkv);
    const T* qrow = q + ((int64_t)(b * nh + h) * seq + i) * dp;
    const T* kbase = k + ((int64_t)(b * nkv + hk) * seq) * dp;
    const T* vbase = v + ((int64_t)(b * nkv + hk) * seq) * dp;
    for (int c = lane; c < dp; c += 64) s_q[w][c] = to_f32(qrow[c]);
    int end = kv_end[b];
    if (causal) end = min(end, i + 1);
    float m = -INFINITY, l = 0.f, acc0 = 0.f, acc1 = 0.f;
    for (int j0 = 0; j0 < end; j0 += 64) {
        const int j = j0 + lane;
        float s = -INFINITY;
        if (j < end && key_mask[(int64_t)b * seq + j]) {
            const T* kr = kbase + (int64_t)j * dp;
            float dot = 0.f;
            for (int c = 0; c < d; c += 4) {
                float kv[4];
                load4(kr + c, kv);
                dot = fmaf(s_q[w][c], kv[0], dot);
                dot = fmaf(s_q[w][c + 1], kv[1], dot);
                dot = fmaf(s_q[w][c + 2], kv[2], dot);
                dot = fmaf(s_q[w][c + 3], kv[3], dot);
            }
            s = dot * scale;
        }
        const float m_new = fmaxf(m, wave_max(s));
        float p = 0.f, alpha = 0.f;
        if (m_new > -INFINITY) {
            p = expf(s - m_new);                       // s = -inf -> 0
            alpha = expf(m - m_new);                   // m = -inf -> 0
        }
        l = l * alpha + wave_sum(p);
        m = m_new;
        s_p[w][lane] = p;                              // same-wave LDS exchange: program order suffices
        acc0 *= alpha;
        acc1 *= alpha;
        const int nj = min(64, end - j0);
        if (lane < d) {
            const T* vr = vbase + (int64_t)j0 * dp + lane;
            for (int jj = 0; jj < nj; ++jj) acc0 = fmaf(s_p[w][jj], to_f32(vr[(int64_t)jj * dp]), acc0);
        }
        if (lane + 64 < d) {
            const T* vr = vbase + (int64_t)j0 * dp + lane + 64;
            for (int jj = 0; jj < nj; ++jj) acc1 = fmaf(s_p[w][jj], to_f32(vr[(int64_t)jj * dp]), acc1);
        }
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (lse && lane == 0) lse[(int64_t)(b * nh + h) * seq + i] = l > 0.f ? m + logf(l) : INFINITY;      // kernels.h attention(): +inf = no visible key
    T* orow = out + ((int64_t)b * seq + i) * ld_out + h * d;
    if (lane < d) orow[lane] = from_f32<T>(acc0 * inv);
    if (lane + 64 < d) orow[lane + 64] = from_f32<T>(acc1 * inv);
    if (h == nh - 1)                                   // zero the K padding of the o-proj GEMM
        for (int c = nh * d + lane; c < out_cols; c += 64) out[((int64_t)b * seq + i) * ld_out + c] = from_f32<T>(0.f);
}

int launch_attn_simple(const void* q, const void* k, const void* v, const uint8_t* key_mask, const int32_t* kv_end,
                       void* out, int64_t ld_out, int B, int T, int nh, int nkv, int d, int dp, float scale,
                       int causal, int dtype, float* lse, hipStream_t s) {
    P2T_REQUIRE(d % 4 == 0 && d <= 128 && dp <= 128 && nh % nkv == 0, "attention: head_dim %d / heads %d/%d unsupported", d, nh, nkv);
    const dim3 grid((unsigned)ceil_div(T, 4), (unsigned)nh, (unsigned)B);
    const int out_cols = (int)(round_up((int64_t)nh * d, 64) < ld_out ? round_up((int64_t)nh * d, 64) : ld_out);
    if (dtype == P2T_BF16)
        attn_simple_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_mask, kv_end,
                                                        (bf16_t*)out, ld_out, T, nh, nkv, d, dp, scale, causal, out_cols, lse);
    else
        attn_simple_kernel<float><<<grid, 256, 0, s>>>((const float*)q, (const float*)k, (const float*)v, key_mask, kv_end,
                                                       (float*)out, ld_out, T, nh, nkv, d, dp, scale, causal, out_cols, lse);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

}  // namespace p2t
