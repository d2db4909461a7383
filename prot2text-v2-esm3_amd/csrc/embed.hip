// Mask preparation, embedding gathers, rotary tables and the QKV post-pass (head split, query
// scale, rotary, V transpose).  All HBM-bound.
//   ESM embeddings: HF EsmEmbeddings.forward, transformers/models/esm/modeling_esm.py:224-271
//   Llama embed:    HF LlamaModel.forward, transformers/models/llama/modeling_llama.py:381
//   rotary:         modeling_esm.py:48-79,141-160 / modeling_llama.py:108-160 (rotate-half layout)
#include "common.h"
#include "kernels.h"

namespace p2t {

// One block per batch row: key_mask u8, kv_info[b] = 1 + last valid key, kv_info[B+b] = prefix flag,
// ESM token-dropout scale.
__global__ void __launch_bounds__(256) mask_prepare_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask,
                                                           int T, int mask_id, int token_dropout,
                                                           uint8_t* __restrict__ key_mask, int32_t* __restrict__ kv_info,
                                                           float* __restrict__ emb_scale) {
    __shared__ int s_end[4], s_cnt[4], s_nmask[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int end = 0, cnt = 0, nm = 0;
    for (int t = threadIdx.x; t < T; t += 256) {
        const int v = mask[(int64_t)b * T + t] != 0;
        key_mask[(int64_t)b * T + t] = (uint8_t)v;
        if (v) end = t + 1;
        cnt += (int)mask[(int64_t)b * T + t];
        if (ids) nm += (ids[(int64_t)b * T + t] == mask_id);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        end = max(end, __shfl_xor(end, o, 64));
        cnt += __shfl_xor(cnt, o, 64);
        nm += __shfl_xor(nm, o, 64);
    }
    if (lane == 0) { s_end[w] = end; s_cnt[w] = cnt; s_nmask[w] = nm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        end = max(max(s_end[0], s_end[1]), max(s_end[2], s_end[3]));
        cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        nm = s_nmask[0] + s_nmask[1] + s_nmask[2] + s_nmask[3];
        kv_info[b] = end;                                   // 1 + last valid key
        kv_info[gridDim.x + b] = (cnt == end);              // mask is a plain prefix (right-padded contract)
        if (emb_scale) {
            // (1 - 0.15*0.8) / (1 - observed mask ratio), modeling_esm.py:256-262
            const float ratio = (float)nm / (float)cnt;
            emb_scale[b * 2 + 0] = token_dropout ? 0.88f : 1.0f;
            emb_scale[b * 2 + 1] = token_dropout ? (1.0f - ratio) : 1.0f;
        }
    }
}

// x[m, :] = mask[m] * ((id == <mask> ? 0 : table[id]) * 0.88 / (1 - ratio_b)); one wave per token.
template <typename T>
__global__ void __launch_bounds__(256) esm_embed_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask,
                                                        const T* __restrict__ table, const float* __restrict__ emb_scale,
                                                        int seq, int H, int vocab, int mask_id, int token_dropout,
                                                        float* __restrict__ x, int64_t M) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int b = (int)(m / seq);
    int64_t id = ids[m];
    const float mk = (float)mask[m];
    const bool zero = (token_dropout && id == mask_id) || id < 0 || id >= vocab;
    if (zero) id = 0;
    const float mul = emb_scale[b * 2 + 0], den = emb_scale[b * 2 + 1];
    const T* row = table + id * H;
    for (int c = lane * 4; c < H; c += 256) {
        float v[4];
        load4(row + c, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float e = zero ? 0.f : v[j];
            if (token_dropout) e = __fdiv_rn(__fmul_rn(e, mul), den);
            v[j] = e * mk;
        }
        store4(x + m * H + c, v);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) llama_embed_kernel(const int64_t* __restrict__ ids, const T* __restrict__ table,
                                                          int H, int vocab, float* __restrict__ x, int64_t M) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    int64_t id = ids[m];
    if (id < 0 || id >= vocab) id = 0;
    const T* row = table + id * H;
    for (int c = lane * 4; c < H; c += 256) {
        float v[4];
        load4(row + c, v);
        store4(x + m * H + c, v);
    }
}

// inv_freq[j] = 1 / theta^(2j/d), optionally with the llama3 wavelength-dependent scaling
// (transformers/modeling_rope_utils.py:580-662), evaluated in fp32 like the reference does.
__global__ void inv_freq_kernel(float* __restrict__ inv_freq, int half, float theta, int llama3, float factor,
                                float low_ff, float high_ff, float orig_max_pos) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= half) return;
    float inv = 1.0f / powf(theta, (float)(2 * j) / (float)(2 * half));
    if (llama3) {
        const float low_wl = orig_max_pos / low_ff, high_wl = orig_max_pos / high_ff;
        const float wavelen = 6.283185307179586f / inv;
        float inv_l = wavelen > low_wl ? inv / factor : inv;
        const float smooth = (orig_max_pos / wavelen - low_ff) / (high_ff - low_ff);
        const float smoothed = (1.0f - smooth) * inv_l / factor + smooth * inv_l;
        const bool medium = !(wavelen < high_wl) && !(wavelen > low_wl);
        inv = medium ? smoothed : inv_l;
    }
    inv_freq[j] = inv;
}

// cs[t, 0:half] = cos(t * inv_freq), cs[t, half:2*half] = sin(...)
__global__ void __launch_bounds__(256) rope_table_kernel(const float* __restrict__ inv_freq, int T, int half,
                                                         float* __restrict__ cs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * half) return;
    const int t = i / half, j = i % half;
    const float a = __fmul_rn((float)t, inv_freq[j]);
    cs[(int64_t)t * 2 * half + j] = cosf(a);
    cs[(int64_t)t * 2 * half + half + j] = sinf(a);
}

// grid (ceil(T/64), nh + 2*nkv, B).  qkv row m: [q heads | k heads | v heads], row stride ldq.
// q: [B, nh, T, dp]  k: [B, nkv, T, dp] (scaled / rotated, zero padded to dp)
// v: [B, nkv, T, dp] (row-major like k: the attention kernels transpose on the fly)
template <typename T>
__global__ void __launch_bounds__(256) qkv_post_kernel(const T* __restrict__ qkv, int64_t ldq, const float* __restrict__ cs,
                                                       T* __restrict__ q, T* __restrict__ k, T* __restrict__ v, int seq,
                                                       int nh, int nkv, int d, int dp, float q_scale) {
    const int b = blockIdx.z, hh = blockIdx.y, t0 = blockIdx.x * 64;
    const int half = d / 2;
    if (hh < nh + nkv) {
        const bool is_q = hh < nh;
        const int head = is_q ? hh : hh - nh;
        const int col0 = is_q ? head * d : nh * d + head * d;
        T* dst = is_q ? q + ((int64_t)(b * nh + head) * seq) * dp : k + ((int64_t)(b * nkv + head) * seq) * dp;
        const float sc = is_q ? q_scale : 1.0f;
        const int hp = dp / 2;                       // loop over (token, j in [0, dp/2))
        for (int i = threadIdx.x; i < 64 * hp; i += 256) {
            const int tl = i / hp, j = i % hp, t = t0 + tl;
            if (t >= seq) continue;
            float o1 = 0.f, o2 = 0.f;
            const T* src = qkv + ((int64_t)b * seq + t) * ldq + col0;
            if (j < half) {
                const float x1 = to_f32(src[j]) * sc, x2 = to_f32(src[j + half]) * sc;
                const float c = cs[(int64_t)t * d + j], s = cs[(int64_t)t * d + half + j];
                o1 = x1 * c - x2 * s;
                o2 = x2 * c + x1 * s;
            }
            // element j -> column j, element j+half -> column j+half; columns >= d are zero
            T* row = dst + (int64_t)t * dp;
            if (j < half) {
                row[j] = from_f32<T>(o1);
                row[j + half] = from_f32<T>(o2);
            }
            // zero pad: columns d .. dp-1 (covered by j in [half, hp) mapped to d + 2*(j-half) + {0,1})
            if (j >= half) {
                const int c0 = d + 2 * (j - half);
                if (c0 < dp) row[c0] = from_f32<T>(0.f);
                if (c0 + 1 < dp) row[c0 + 1] = from_f32<T>(0.f);
            }
        }
    } else {
        const int head = hh - nh - nkv;
        const int col0 = (nh + nkv) * d + head * d;
        T* dst = v + ((int64_t)(b * nkv + head) * seq) * dp;
        for (int i = threadIdx.x; i < 64 * dp; i += 256) {
            const int tl = i / dp, c = i % dp, t = t0 + tl;
            if (t < seq) dst[(int64_t)t * dp + c] = c < d ? qkv[((int64_t)b * seq + t) * ldq + col0 + c] : from_f32<T>(0.f);
        }
    }
}

// Qwen3: head split with RMSNorm over head_dim on every query / key head (f32 statistics, eps inside the root, weight last:
// HF Qwen3RMSNorm = LlamaRMSNorm) followed by the rotation; value heads are copied.  One wave per (token, head): lane j < d/2
// holds the rotary pair (j, j + d/2).  Restates HF Qwen3Attention.forward up to the attention call.
template <typename T>
__global__ void __launch_bounds__(256) qk_norm_rope_kernel(const T* __restrict__ qkv, int64_t ldq, const float* __restrict__ cs,
                                                           const float* __restrict__ qw, const float* __restrict__ kw, float eps,
                                                           T* __restrict__ q, T* __restrict__ k, T* __restrict__ v, int64_t rows, int seq,
                                                           int nh, int nkv, int d, int dp, float q_scale) {
    const int lane = threadIdx.x & 63, heads = nh + 2 * nkv, half = d / 2;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t bt = row / heads;
    const int hh = (int)(row - bt * heads);
    const int b = (int)(bt / seq), t = (int)(bt - (int64_t)b * seq);
    const T* src = qkv + bt * ldq + (int64_t)hh * d;
    float x1 = 0.f, x2 = 0.f;
    if (lane < half) { x1 = to_f32(src[lane]); x2 = to_f32(src[lane + half]); }
    T* dst;
    if (hh < nh + nkv) {
        const bool is_q = hh < nh;
        const float* w = is_q ? qw : kw;
        const float ss = wave_sum(x1 * x1 + x2 * x2);
        const float rstd = rsqrtf(ss / (float)d + eps);
        dst = is_q ? q + (((int64_t)b * nh + hh) * seq + t) * dp : k + (((int64_t)b * nkv + (hh - nh)) * seq + t) * dp;
        if (lane < half) {
            const float a1 = w[lane] * (x1 * rstd), a2 = w[lane + half] * (x2 * rstd);
            const float c = cs[(int64_t)t * d + lane], s = cs[(int64_t)t * d + half + lane];
            const float qs = is_q ? q_scale : 1.0f;              // (the towers fold the softmax scale into q: kernels.h attention())
            dst[lane] = from_f32<T>((a1 * c - a2 * s) * qs);
            dst[lane + half] = from_f32<T>((a2 * c + a1 * s) * qs);
        }
    } else {
        dst = v + (((int64_t)b * nkv + (hh - nh - nkv)) * seq + t) * dp;
        if (lane < half) { dst[lane] = from_f32<T>(x1); dst[lane + half] = from_f32<T>(x2); }
    }
    for (int c = d + lane; c < dp; c += 64) dst[c] = from_f32<T>(0.f);
}

int launch_qk_norm_rope(const void* qkv, int64_t ldq, const float* cs, const float* q_norm_w, const float* k_norm_w, float eps, void* q,
                        void* k, void* v, int B, int T, int nh, int nkv, int d, int dp, float q_scale, int dtype, hipStream_t s) {
    P2T_REQUIRE(d % 2 == 0 && d <= 128 && dp >= d && dp <= 128 && q_norm_w && k_norm_w, "qk_norm_rope: head_dim %d (padded %d) unsupported", d, dp);
    const int64_t rows = (int64_t)B * T * (nh + 2 * nkv);
    const dim3 grid((unsigned)ceil_div(rows, 4));
    if (dtype == P2T_BF16)
        qk_norm_rope_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)qkv, ldq, cs, q_norm_w, k_norm_w, eps, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v,
                                                          rows, T, nh, nkv, d, dp, q_scale);
    else
        qk_norm_rope_kernel<float><<<grid, 256, 0, s>>>((const float*)qkv, ldq, cs, q_norm_w, k_norm_w, eps, (float*)q, (float*)k, (float*)v, rows,
                                                         T, nh, nkv, d, dp, q_scale);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

int launch_mask_prepare(const int64_t* ids, const int64_t* mask, int B, int T, int mask_id, int token_dropout,
                        uint8_t* key_mask, int32_t* kv_info, float* emb_scale, hipStream_t s) {
    mask_prepare_kernel<<<B, 256, 0, s>>>(ids, mask, T, mask_id, token_dropout, key_mask, kv_info, emb_scale);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

int launch_esm_embed(const int64_t* ids, const int64_t* mask, const void* table, int dtype, const float* emb_scale, int T,
                     int H, int vocab, int mask_id, int token_dropout, float* x, int64_t M, hipStream_t s) {
    const dim3 grid((unsigned)ceil_div(M, 4));
    if (dtype == P2T_BF16)
        esm_embed_kernel<bf16_t><<<grid, 256, 0, s>>>(ids, mask, (const bf16_t*)table, emb_scale, T, H, vocab, mask_id, token_dropout, x, M);
    else
        esm_embed_kernel<float><<<grid, 256, 0, s>>>(ids, mask, (const float*)table, emb_scale, T, H, vocab, mask_id, token_dropout, x, M);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

int launch_llama_embed(const int64_t* ids, const void* table, int dtype, int H, int vocab, float* x, int64_t M,
                       hipStream_t s) {
    const dim3 grid((unsigned)ceil_div(M, 4));
    if (dtype == P2T_BF16)
        llama_embed_kernel<bf16_t><<<grid, 256, 0, s>>>(ids, (const bf16_t*)table, H, vocab, x, M);
    else
        llama_embed_kernel<float><<<grid, 256, 0, s>>>(ids, (const float*)table, H, vocab, x, M);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

int launch_inv_freq(float* inv_freq, int half, float theta, int llama3, float factor, float low_ff, float high_ff,
                    float orig_max_pos, hipStream_t s) {
    inv_freq_kernel<<<ceil_div(half, 64), 64, 0, s>>>(inv_freq, half, theta, llama3, factor, low_ff, high_ff, orig_max_pos);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

int launch_rope_table(const float* inv_freq, int T, int half, float* cs, hipStream_t s) {
    rope_table_kernel<<<ceil_div((int64_t)T * half, 256), 256, 0, s>>>(inv_freq, T, half, cs);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

int launch_qkv_post(const void* qkv, int64_t ldq, const float* cs, void* q, void* k, void* v, int B, int T, int nh,
                    int nkv, int d, int dp, float q_scale, int dtype, hipStream_t s) {
    P2T_REQUIRE(d % 2 == 0 && d <= 128 && dp >= d && dp <= 128 && dp % 2 == 0, "qkv_post: head_dim %d (padded %d) unsupported", d, dp);
    const dim3 grid((unsigned)ceil_div(T, 64), (unsigned)(nh + 2 * nkv), (unsigned)B);
    if (dtype == P2T_BF16)
        qkv_post_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)qkv, ldq, cs, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v, T, nh, nkv, d, dp, q_scale);
    else
        qkv_post_kernel<float><<<grid, 256, 0, s>>>((const float*)qkv, ldq, cs, (float*)q, (float*)k, (float*)v, T, nh, nkv, d, dp, q_scale);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

}  // namespace p2t
