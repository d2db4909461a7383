// Generation with a KV cache: what `llama_decoder.generate(inputs_embeds=..., attention_mask=...)` runs under
// Esm2LlamaInstructForCausalLM.generate (reference models/modeling_esm2llama_instruct.py:217-251; HF GenerationMixin + the cached
// LlamaAttention path, transformers/models/llama/modeling_llama.py).
//
// Layout (sized for 288 GB of HBM: nothing is paged, nothing is re-laid-out per step):
//   * prompts are COMPACTED first (p2t_compact_rows): the valid tokens of every row move to the front in order, so positions are
//     0..len-1 -- exactly HF's `position_ids = cumsum(attention_mask) - 1` on the tokens under the mask, for left padding, right
//     padding or holes alike -- and every row's keys are a prefix of its cache row;
//   * the cache has two segments: the PROMPT segment [layer][B0][kv_head][Tp][dp] (keys, post-rotation) + the same transposed
//     [layer][B0][kv_head][dp][Tp] (values: the decode kernel reads 8 consecutive keys of one feature with one 16-byte load), written
//     once by the prefill; and the GENERATED segment [layer][BB][kv_head][G][dp] / [..][dp][G] with BB = B0 * group rows (group =
//     beams per prompt: all beams of a prompt share its prompt segment, beam re-ordering touches the generated segment only);
//   * every row generates in lock-step: ONE device counter `step` = generated tokens already in the cache.  A decode step reads it on
//     the device (so the whole step replays as a HIP graph without a host round trip), appends at index `step`, attends to
//     prompt_len[b0] + step + 1 keys and increments it last.
#include "common.h"
#include "kernels.h"

namespace p2t {
namespace {

// ---------------------------------------------------------------------------------------------
// compaction of the prompt rows
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) compact_index_kernel(const int64_t* __restrict__ mask, int T, int32_t* __restrict__ dst,
                                                            int64_t* __restrict__ out_mask, int32_t* __restrict__ lens) {
    __shared__ int cnt[256];
    __shared__ int total;
    const int b = blockIdx.x, tid = threadIdx.x, per = (T + 255) / 256;
    const int t0 = min(T, tid * per), t1 = min(T, t0 + per);
    const int64_t* m = mask + (int64_t)b * T;
    int c = 0;
    for (int t = t0; t < t1; ++t) c += m[t] != 0;
    cnt[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int a = 0;
        for (int i = 0; i < 256; ++i) { const int v = cnt[i]; cnt[i] = a; a += v; }
        total = a;
        lens[b] = a;
    }
    __syncthreads();
    int a = cnt[tid];
    for (int t = t0; t < t1; ++t) dst[(int64_t)b * T + t] = m[t] != 0 ? a++ : -1;
    const int n = total;
    for (int t = tid; t < T; t += 256) out_mask[(int64_t)b * T + t] = t < n ? 1 : 0;
}

__global__ void __launch_bounds__(256) compact_copy_kernel(const float* __restrict__ x, const int32_t* __restrict__ dst, int T, int H,
                                                           float* __restrict__ out) {
    const int t = blockIdx.x, b = blockIdx.y;
    const int to = dst[(int64_t)b * T + t];
    if (to < 0) return;
    const float4* s = reinterpret_cast<const float4*>(x + ((int64_t)b * T + t) * H);
    float4* d = reinterpret_cast<float4*>(out + ((int64_t)b * T + to) * H);
    for (int c = threadIdx.x; c < H / 4; c += 256) d[c] = s[c];
}

// last valid row of every prompt: out[b] = x[b, len[b] - 1]
__global__ void __launch_bounds__(256) gather_last_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens, int T, int H,
                                                          float* __restrict__ out) {
    const int b = blockIdx.x;
    const int t = max(0, min(T, lens[b]) - 1);
    const float* s = x + ((int64_t)b * T + t) * H;
    for (int c = threadIdx.x; c < H; c += 256) out[(int64_t)b * H + c] = s[c];
}

// v [BH][seq][dp] -> vt [BH][dp][Tp] (columns 0..seq-1)
template <typename T>
__global__ void __launch_bounds__(256) v_transpose_store_kernel(const T* __restrict__ v, T* __restrict__ vt, int seq, int dp, int Tp) {
    __shared__ T tile[64][130];
    const int bh = blockIdx.y, t0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * dp; i += 256) {
        const int tl = i / dp, c = i - tl * dp;
        tile[tl][c] = t0 + tl < seq ? v[((int64_t)bh * seq + t0 + tl) * dp + c] : from_f32<T>(0.f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * dp; i += 256) {
        const int c = i >> 6, tl = i & 63;
        if (t0 + tl < Tp) vt[((int64_t)bh * dp + c) * Tp + t0 + tl] = tile[tl][c];
    }
}

// ---------------------------------------------------------------------------------------------
// one new token per row: rotation at its position + append to the generated segment
// ---------------------------------------------------------------------------------------------
// One wave per (row, head slot): lane j < d/2 holds the rotary pair (j, j + d/2).  qkv: the f32 result of the QKV projection in the
// ROW ORDER OF qkv_w (for head_dim 128 without q/k norm: the packed order of p2t_llama_layer).  `round_first`: the prefill of this
// model rounds the projection to the model dtype before the rotation (its unfused path), so the step does too.
template <typename T>
__global__ void __launch_bounds__(256) rope_append_kernel(const float* __restrict__ qkv, int64_t ldq, const float* __restrict__ inv_freq,
                                                          const float* __restrict__ qw, const float* __restrict__ kw, float eps,
                                                          const int32_t* __restrict__ prompt_len, const int32_t* __restrict__ step_ptr,
                                                          int group, T* __restrict__ qbuf, T* __restrict__ kg, T* __restrict__ vtg, int BB,
                                                          int nh, int nkv, int d, int dp, int G, float q_scale, int packed128,
                                                          int round_first) {
    const int lane = threadIdx.x & 63, heads = nh + 2 * nkv, half = d / 2;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= BB * heads) return;
    const int bb = row / heads, hh = row - bb * heads;
    const int step = min(max(step_ptr[0], 0), G - 1);
    const int pos = prompt_len[bb / group] + step;
    const float* src = qkv + (int64_t)bb * ldq + (int64_t)hh * d;
    auto at = [&](int j) {
        int p = j;
        if (packed128) { const int blk = j >> 5; p = blk == 1 ? j + 32 : (blk == 2 ? j - 32 : j); }
        const float v = src[p];
        return round_first ? to_f32(from_f32<T>(v)) : v;
    };
    float x1 = 0.f, x2 = 0.f;
    if (lane < half) { x1 = at(lane); x2 = at(lane + half); }
    if (hh < nh + nkv) {
        const bool is_q = hh < nh;
        const float qs = is_q ? q_scale : 1.0f;
        float c = 1.f, s = 0.f;
        if (lane < half) {
            const float a = __fmul_rn((float)pos, inv_freq[lane]);
            c = cosf(a);
            s = sinf(a);
        }
        float o1, o2;
        if (qw) {                                   // Qwen3: RMSNorm over the head, rotation, then the folded scale (qk_norm_rope_kernel)
            const float* w = is_q ? qw : kw;
            const float ss = wave_sum(x1 * x1 + x2 * x2);
            const float rstd = rsqrtf(ss / (float)d + eps);
            const float a1 = lane < half ? w[lane] * (x1 * rstd) : 0.f, a2 = lane < half ? w[lane + half] * (x2 * rstd) : 0.f;
            o1 = (a1 * c - a2 * s) * qs;
            o2 = (a2 * c + a1 * s) * qs;
        } else {                                    // scale first, then the rotation (the QKV + RoPE epilogue / qkv_post_kernel)
            const float a1 = x1 * qs, a2 = x2 * qs;
            o1 = a1 * c - a2 * s;
            o2 = a2 * c + a1 * s;
        }
        T* dst = is_q ? qbuf + ((int64_t)bb * nh + hh) * dp : kg + (((int64_t)bb * nkv + (hh - nh)) * G + step) * dp;
        if (lane < half) { dst[lane] = from_f32<T>(o1); dst[lane + half] = from_f32<T>(o2); }
        for (int cc = d + lane; cc < dp; cc += 64) dst[cc] = from_f32<T>(0.f);
    } else {
        T* dst = vtg + ((int64_t)bb * nkv + (hh - nh - nkv)) * dp * G + step;
        if (lane < half) { dst[(int64_t)lane * G] = from_f32<T>(x1); dst[(int64_t)(lane + half) * G] = from_f32<T>(x2); }
        for (int cc = d + lane; cc < dp; cc += 64) dst[(int64_t)cc * G] = from_f32<T>(0.f);
    }
}

// ---------------------------------------------------------------------------------------------
// attention of ONE query token per row over the two cache segments
// ---------------------------------------------------------------------------------------------
// grid (BB * nkv, head chunks): ONE block owns all keys of a (row, kv head) -- cut into 64-key tiles (prompt segment, then generated
// segment), tile t goes to wave t mod NW.  A wave keeps a running (max, sum, output) per query head of the GQA group -- they all read
// the same keys -- in base-2 form; the waves are merged in LDS in wave order (decode_merge): no atomics, nothing through memory.
//   scores: lane = key, 16-byte pieces of its K row against q (f32 in LDS, broadcast reads);
//   values: lane = feature (rows of the transposed segment), 16-byte pieces = 8 (bf16) / 4 (f32) consecutive keys against P in LDS.
template <typename T> struct Piece;
template <> struct Piece<bf16_t> { static constexpr int N = 8; };
template <> struct Piece<float> { static constexpr int N = 4; };
template <typename T, int N> __device__ __forceinline__ void load_piece(const T* p, float (&v)[N]) {
    if constexpr (N == 8) load8(p, v); else load4(p, v);
}

// Merge of a block's per-wave results (red[w][head][0..DP-1] = un-normalised output, [DP] = running max, [DP + 1] = sum; base-2 form),
// in wave order, and the store of the normalised rows.  One block owns ALL keys of its (row, kv head): a split over several blocks
// with a last-arriver merge through global memory was measured at 42 us per launch against 24 for this form at 8 x 1121 keys -- three
// more dependent round trips to memory and a release fence per block cost more than the idle CUs.
template <typename T, int DP, int GH, int NW>
__device__ __forceinline__ void decode_merge(float (&red)[NW][GH][DP + 2], int bb, int h0, int nhead, int d, T* __restrict__ out, int64_t ld_out) {
    for (int i = threadIdx.x; i < GH * DP; i += NW * 64) {
        const int g = i / DP, c = i - g * DP;
        if (g >= nhead || c >= d) continue;
        float M = -INFINITY;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) M = fmaxf(M, red[ww][g][DP]);
        float o = 0.f, L = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            const float mw = red[ww][g][DP];
            if (mw > -INFINITY) {
                const float f = exp2f(mw - M);
                o += f * red[ww][g][c];
                L += f * red[ww][g][DP + 1];
            }
        }
        out[(int64_t)bb * ld_out + (int64_t)(h0 + g) * d + c] = from_f32<T>(o / L);
    }
}

template <typename T, int DP, int GH>
__global__ void __launch_bounds__(512) attn_decode_kernel(const T* __restrict__ q, const T* __restrict__ kp, const T* __restrict__ vtp,
                                                          const T* __restrict__ kg, const T* __restrict__ vtg,
                                                          const int32_t* __restrict__ prompt_len, const int32_t* __restrict__ step_ptr,
                                                          int group, int nh, int nkv, int G, int Tp, int Gcap, float c_exp, int round_p,
                                                          T* __restrict__ out, int64_t ld_out, int d) {
    constexpr int PN = Piece<T>::N, R = DP > 64 ? DP / 64 : 1, PW = DP + 2, NW = 8;
    __shared__ float sq[GH][DP];
    __shared__ float sp[NW][GH][64];
    __shared__ float red[NW][GH][PW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pair = blockIdx.x, bb = pair / nkv, kvh = pair - bb * nkv, b0 = bb / group;
    const int hc = blockIdx.y;
    const int h0 = kvh * G + hc * GH, nhead = min(GH, G - hc * GH);          // query heads h0 .. h0 + nhead - 1
    for (int i = threadIdx.x; i < GH * DP; i += NW * 64) {
        const int g = i / DP, c = i - g * DP;
        sq[g][c] = g < nhead ? to_f32(q[((int64_t)bb * nh + h0 + g) * DP + c]) : 0.f;
    }
    __syncthreads();
    const int n0 = min(max(prompt_len[b0], 0), Tp), n1 = min(max(step_ptr[0], 0) + 1, Gcap);
    const int tiles0 = (n0 + 63) >> 6, tiles = tiles0 + ((n1 + 63) >> 6);
    const T* k_seg0 = kp + ((int64_t)b0 * nkv + kvh) * Tp * DP;
    const T* k_seg1 = kg + ((int64_t)bb * nkv + kvh) * Gcap * DP;
    const T* v_seg0 = vtp + ((int64_t)b0 * nkv + kvh) * DP * Tp;
    const T* v_seg1 = vtg + ((int64_t)bb * nkv + kvh) * DP * Gcap;

    float m_run[GH], l_run[GH], acc[GH][R];
#pragma unroll
    for (int g = 0; g < GH; ++g) {
        m_run[g] = -INFINITY;
        l_run[g] = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) acc[g][r] = 0.f;
    }
    for (int t = w; t < tiles; t += NW) {
        const int sg = t >= tiles0, tt = sg ? t - tiles0 : t, n = sg ? n1 : n0, cap = sg ? Gcap : Tp;
        const int key = tt * 64 + lane;
        const bool valid = key < n, full = tt * 64 + 64 <= n;
        const T* krow = (sg ? k_seg1 : k_seg0) + (int64_t)key * DP;
        float sc[GH];
#pragma unroll
        for (int g = 0; g < GH; ++g) sc[g] = 0.f;
        if (valid) {
#pragma unroll 4
            for (int c = 0; c < DP; c += PN) {
                float kv[PN];
                load_piece<T, PN>(krow + c, kv);
#pragma unroll
                for (int g = 0; g < GH; ++g) {
#pragma unroll
                    for (int e = 0; e < PN; ++e) sc[g] = fmaf(kv[e], sq[g][c + e], sc[g]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < GH; ++g) {
            const float e2 = valid ? sc[g] * c_exp : -INFINITY;
            const float m_new = fmaxf(m_run[g], wave_max(e2));            // every tile holds at least one valid key
            const float alpha = exp2f(m_run[g] - m_new);
            float p = valid ? exp2f(e2 - m_new) : 0.f;
            l_run[g] = l_run[g] * alpha + p;
            if (round_p) p = to_f32(from_f32<T>(p));                      // the MFMA forward feeds P to the matrix pipe in bf16
            sp[w][g][lane] = p;
            m_run[g] = m_new;
#pragma unroll
            for (int r = 0; r < R; ++r) acc[g][r] *= alpha;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int dim = lane + 64 * r;
            if (dim < DP) {
                const T* vrow = (sg ? v_seg1 : v_seg0) + (int64_t)dim * cap + tt * 64;
#pragma unroll 2
                for (int j = 0; j < 64; j += PN) {
                    float vv[PN];
                    load_piece<T, PN>(vrow + j, vv);
                    if (!full) {
#pragma unroll
                        for (int e = 0; e < PN; ++e) vv[e] = tt * 64 + j + e < n ? vv[e] : 0.f;
                    }
#pragma unroll
                    for (int g = 0; g < GH; ++g) {
#pragma unroll
                        for (int e = 0; e < PN; ++e) acc[g][r] = fmaf(sp[w][g][j + e], vv[e], acc[g][r]);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int g = 0; g < GH; ++g) {
        const float l = wave_sum(l_run[g]);
        if (lane == 0) { red[w][g][DP] = m_run[g]; red[w][g][DP + 1] = l; }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (lane + 64 * r < DP) red[w][g][lane + 64 * r] = acc[g][r];
    }
    __syncthreads();
    decode_merge<T, DP, GH, NW>(red, bb, h0, nhead, d, out, ld_out);
}

// bf16 form on the matrix pipe (`v_mfma_f32_16x16x32_bf16`, fragments straight from global memory -- no LDS staging, the step is a
// stream).  Per wave and 64-key tile:
//   S[key, head] = K . q^T: A = 16 keys x 32 features of K (a lane: 16 bytes of one key row), B = q (16 head slots, the group's G
//     real heads first, zeros behind), so a lane holds 4 keys of ONE head slot -- the softmax statistics of a head live in the four
//     lanes l, l + 16, l + 32, l + 48;
//   the A rows of the sub-tiles are permuted (row 4 g + r of sub-tile t <-> key 32 (t / 2) + 8 g + 4 (t % 2) + r), so that the eight
//     probabilities a lane holds after two sub-tiles are 8 CONSECUTIVE keys: exactly the B operand of
//   O[feature, head] += V^T . P: A = 16 features x 32 keys of the transposed value segment (a lane: 16 bytes = 8 consecutive keys of
//     one feature), no exchange between lanes anywhere.
template <int DP, int GH, int NW>
__global__ void __launch_bounds__(NW * 64) attn_decode_mfma_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kp,
                                                               const bf16_t* __restrict__ vtp, const bf16_t* __restrict__ kg,
                                                               const bf16_t* __restrict__ vtg, const int32_t* __restrict__ prompt_len,
                                                               const int32_t* __restrict__ step_ptr, int group, int nh, int nkv, int G, int Tp,
                                                               int Gcap, float c_exp, bf16_t* __restrict__ out, int64_t ld_out, int d) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    constexpr int PW = DP + 2, KK = DP / 32, DT = DP / 16;
    __shared__ float red[NW][GH][PW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
    const int pair = blockIdx.x, bb = pair / nkv, kvh = pair - bb * nkv, b0 = bb / group;
    const int hc = blockIdx.y;
    const int h0 = kvh * G + hc * GH, nhead = min(GH, G - hc * GH);
    const bf16x8 zero8 = {};
    bf16x8 qf[KK];                                                  // B operand of S: head slot j, features 32 kk + 8 g ..
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
        qf[kk] = j < nhead ? *reinterpret_cast<const bf16x8*>(q + ((int64_t)bb * nh + h0 + j) * DP + 32 * kk + 8 * g) : zero8;
    const int n0 = min(max(prompt_len[b0], 0), Tp), n1 = min(max(step_ptr[0], 0) + 1, Gcap);
    const int tiles0 = (n0 + 63) >> 6, tiles = tiles0 + ((n1 + 63) >> 6);
    const bf16_t* k_seg0 = kp + ((int64_t)b0 * nkv + kvh) * Tp * DP;
    const bf16_t* k_seg1 = kg + ((int64_t)bb * nkv + kvh) * Gcap * DP;
    const bf16_t* v_seg0 = vtp + ((int64_t)b0 * nkv + kvh) * DP * Tp;
    const bf16_t* v_seg1 = vtg + ((int64_t)bb * nkv + kvh) * DP * Gcap;
    // A rows of S: lane row i = j <-> key offset 8 (i / 4) + (i % 4) inside a 32-key half, + 4 for the odd sub-tile
    const int krow = 8 * (j >> 2) + (j & 3);
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = w; t < tiles; t += NW) {
        const int sg = t >= tiles0, tt = sg ? t - tiles0 : t, n = sg ? n1 : n0, cap = sg ? Gcap : Tp;
        const bool full = tt * 64 + 64 <= n;
        const bf16_t* kb = (sg ? k_seg1 : k_seg0) + (int64_t)tt * 64 * DP + 8 * g;
        const bf16_t* vb = (sg ? v_seg1 : v_seg0) + (int64_t)tt * 64 + 8 * g;
        // all fragments of the tile first (32 loads of 16 bytes in flight per lane), then the arithmetic
        bf16x8 kf[4][KK], vf[2][DT];
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
                kf[st][kk] = *reinterpret_cast<const bf16x8*>(kb + (int64_t)(32 * (st >> 1) + 4 * (st & 1) + krow) * DP + 32 * kk);
#pragma unroll
        for (int hv = 0; hv < 2; ++hv)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vf[hv][dt] = *reinterpret_cast<const bf16x8*>(vb + (int64_t)(16 * dt + j) * cap + 32 * hv);
        f32x4 sc[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            sc[st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) sc[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[st][kk], qf[kk], sc[st], 0, 0, 0);
        }
        // lane (head j, g): sc[st][r] is the key 32 (st / 2) + 8 g + 4 (st % 2) + r of the tile
        float mx = -INFINITY;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = full || tt * 64 + 32 * (st >> 1) + 8 * g + 4 * (st & 1) + r < n;
                sc[st][r] = ok ? sc[st][r] * c_exp : -INFINITY;
                mx = fmaxf(mx, sc[st][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);                          // every tile holds at least one valid key
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int hv = 0; hv < 2; ++hv) {
            float p[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                p[e] = exp2f(sc[2 * hv + (e >> 2)][e & 3] - m_new);
                psum += p[e];
            }
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            pf[hv] = __builtin_bit_cast(bf16x8, u32x4{pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), pack_bf16x2(p[6], p[7])});
        }
        l_run = l_run * alpha + psum;
        if (!full) {                                                  // keys behind the valid prefix: 0 x garbage must stay 0
#pragma unroll
            for (int hv = 0; hv < 2; ++hv)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    s16x8 v = __builtin_bit_cast(s16x8, vf[hv][dt]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = tt * 64 + 32 * hv + 8 * g + e < n ? v[e] : (short)0;
                    vf[hv][dt] = __builtin_bit_cast(bf16x8, v);
                }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            acc[dt] *= alpha;
#pragma unroll
            for (int hv = 0; hv < 2; ++hv) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[hv][dt], pf[hv], acc[dt], 0, 0, 0);
        }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (j < GH) {                                                     // lane (head j, g) holds the features 16 dt + 4 g + r of its head
        if (g == 0) { red[w][j][DP] = m_run; red[w][j][DP + 1] = l_run; }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w][j][16 * dt + 4 * g + r] = acc[dt][r];
    }
    __syncthreads();
    decode_merge<bf16_t, DP, GH, NW>(red, bb, h0, nhead, d, out, ld_out);
}

// ---------------------------------------------------------------------------------------------
// token selection (greedy) and step bookkeeping
// ---------------------------------------------------------------------------------------------
// torch.argmax semantics over the first V columns (lowest index among equal maxima), then HF's finished-row rule
// (generation/utils.py, _sample: next = next * unfinished + pad * (1 - unfinished); a row finishes once it emits an eos id).
template <typename T>
__global__ void __launch_bounds__(1024) greedy_select_kernel(const T* __restrict__ logits, int64_t ld, int V, const int64_t* __restrict__ eos,
                                                             int n_eos, int64_t pad, int32_t* __restrict__ finished,
                                                             int64_t* __restrict__ next, int64_t* __restrict__ out_tokens, int64_t ld_tok,
                                                             const int32_t* __restrict__ step_ptr, int Gcap) {
    constexpr int PN = Piece<T>::N, kNone = 0x7fffffff;
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int bb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const T* row = logits + (int64_t)bb * ld;
    float best = -INFINITY;
    int idx = kNone;
    auto take = [&](float v, int c) {                   // ascending c per thread: a later equal value never replaces an earlier one
        if (idx == kNone || v > best) { best = v; idx = c; }
    };
    const int Vv = (ld % PN == 0) ? V / PN * PN : 0;     // 16-byte pieces while the row pitch keeps them aligned
    for (int c = tid * PN; c < Vv; c += 1024 * PN) {
        float v[PN];
        load_piece<T, PN>(row + c, v);
#pragma unroll
        for (int e = 0; e < PN; ++e) take(v[e], c + e);
    }
    for (int c = Vv + tid; c < V; c += 1024) take(to_f32(row[c]), c);
    auto better = [](float v, int i, float bvv, int bii) { return i != kNone && (bii == kNone || v > bvv || (v == bvv && i < bii)); };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v = __shfl_xor(best, o, 64);
        const int i2 = __shfl_xor(idx, o, 64);
        if (better(v, i2, best, idx)) { best = v; idx = i2; }
    }
    if (lane == 0) { bv[w] = best; bi[w] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int ww = 1; ww < 16; ++ww)
            if (better(bv[ww], bi[ww], best, idx)) { best = bv[ww]; idx = bi[ww]; }
        int64_t tok = idx;
        const int fin = finished[bb];
        if (fin) tok = pad;
        next[bb] = tok;
        const int st = min(max(step_ptr[0], 0), Gcap - 1);
        out_tokens[(int64_t)bb * ld_tok + st] = tok;
        if (!fin) {
            for (int e = 0; e < n_eos; ++e)
                if (tok == eos[e]) finished[bb] = 1;
        }
    }
}

// beam re-ordering of the generated segment: grid (layers * BB * kv_heads)
template <typename T>
__global__ void __launch_bounds__(256) kv_reorder_kernel(const T* __restrict__ ks, const T* __restrict__ vs, T* __restrict__ kd, T* __restrict__ vd,
                                                         const int64_t* __restrict__ src_row, const int32_t* __restrict__ step_ptr, int BB, int nkv,
                                                         int dp, int G) {
    const int n = min(max(step_ptr[0], 0), G);
    const int idx = blockIdx.x, kvh = idx % nkv, r = (idx / nkv) % BB, l = idx / (nkv * BB);
    int64_t sr = src_row[r];
    if (sr < 0 || sr >= BB) sr = r;
    const int64_t to = (((int64_t)l * BB + r) * nkv + kvh) * (int64_t)G * dp, from = (((int64_t)l * BB + sr) * nkv + kvh) * (int64_t)G * dp;
    for (int i = threadIdx.x; i < n * dp; i += 256) kd[to + i] = ks[from + i];
    for (int i = threadIdx.x; i < n * dp; i += 256) {
        const int c = i / n, j = i - c * n;
        vd[to + (int64_t)c * G + j] = vs[from + (int64_t)c * G + j];
    }
}

__global__ void advance_kernel(int32_t* step) { step[0] += 1; }

struct DecodeBuffers {
    float* x; void* h; float* qkv; void* qb; void* ao; void* act; float* inv_freq;
    uint8_t* hq; uint8_t* hs; uint8_t* aoq; uint8_t* aos; uint8_t* actq; uint8_t* acts;      // gemm_fp8: e4m3 rows + their E8M0 scales
    int ZC, GH;
};

int pick_gh(int G) { return G <= 1 ? 1 : (G <= 2 ? 2 : (G <= 4 ? 4 : 8)); }

// GH query heads of a GQA group per block, ZC chunks of them (more than 8 heads per kv head: the keys are read once per chunk)
void attn_decode_plan(int nh, int nkv, int* ZC, int* GH) {
    const int G = nh / nkv;
    *GH = pick_gh(G);
    *ZC = (G + *GH - 1) / *GH;
}

size_t decode_plan(const p2t_llama_config* c, int BB, int Tp, int Gcap, Arena* ar, DecodeBuffers* b) {
    const size_t e = dtype_size(c->dtype);
    const int64_t H = c->hidden, F = c->ffn, Hp = round_up(H, 64), Fp = round_up(F, 64);
    const int d = c->head_dim, dp = head_dim_padded(d), nh = c->heads, nkv = c->kv_heads;
    const int64_t QO = round_up((int64_t)nh * d, 64), NQKV = (int64_t)(nh + 2 * nkv) * d;
    Arena local(nullptr, ~(size_t)0 >> 1);
    Arena& a = ar ? *ar : local;
    DecodeBuffers t;
    attn_decode_plan(nh, nkv, &t.ZC, &t.GH);
    t.x = (float*)a.take(sizeof(float) * (size_t)BB * H);
    t.h = a.take(e * (size_t)BB * Hp);
    t.qkv = (float*)a.take(sizeof(float) * (size_t)BB * NQKV);
    t.qb = a.take(e * (size_t)BB * nh * dp);
    t.ao = a.take(e * (size_t)BB * QO);
    t.act = a.take(e * (size_t)BB * Fp);
    t.inv_freq = (float*)a.take(sizeof(float) * (d / 2 + 1));
    t.hq = t.hs = t.aoq = t.aos = t.actq = t.acts = nullptr;
    if (c->gemm_fp8) {
        const int64_t Hq = round_up(H, 128), Fq = round_up(F, 128), QOq = round_up((int64_t)nh * d, 128);
        t.hq = (uint8_t*)a.take((size_t)BB * Hq);    t.hs = (uint8_t*)a.take((size_t)BB);
        t.aoq = (uint8_t*)a.take((size_t)BB * QOq);  t.aos = (uint8_t*)a.take((size_t)BB);
        t.actq = (uint8_t*)a.take((size_t)BB * Fq);  t.acts = (uint8_t*)a.take((size_t)BB);
    }
    if (b) *b = t;
    return a.off + 256;
}

template <typename T>
int launch_attn_decode_t(const DecodeBuffers& b, const p2t_kv_cache* kc, int layer, int BB, int nh, int nkv, int d, int dp, float c_exp, int round_p,
                         int64_t QO, hipStream_t s, int use_mfma = 1) {
    const int G = nh / nkv;
    const size_t per_p = (size_t)kc->B0 * nkv * kc->Tp * dp, per_g = (size_t)BB * nkv * kc->G * dp;
    const T* kp = (const T*)kc->k_prompt + per_p * layer;
    const T* vtp = (const T*)kc->vt_prompt + per_p * layer;
    const T* kg = (const T*)kc->k_gen + per_g * layer;
    const T* vtg = (const T*)kc->vt_gen + per_g * layer;
    const dim3 grid((unsigned)(BB * nkv), (unsigned)b.ZC);
    if constexpr (sizeof(T) == 2) {
        if (use_mfma) {                      // (0: the lane-per-key kernel below, the form every dtype can run)
#define P2T_ADM(DPV, GHV)                                                                                                             \
    attn_decode_mfma_kernel<DPV, GHV, (GHV <= 4 && DPV <= 64 ? 16 : 8)><<<grid, (GHV <= 4 && DPV <= 64 ? 16 : 8) * 64, 0, s>>>(                                  \
        (const bf16_t*)b.qb, (const bf16_t*)kp, (const bf16_t*)vtp, (const bf16_t*)kg, (const bf16_t*)vtg, kc->prompt_len, kc->step, kc->group, nh, \
        nkv, G, kc->Tp, kc->G, c_exp, (bf16_t*)b.ao, QO, d)
#define P2T_ADM_G(DPV)                                                                                                                \
    do {                                                                                                                              \
        if (b.GH == 1) P2T_ADM(DPV, 1); else if (b.GH == 2) P2T_ADM(DPV, 2); else if (b.GH == 4) P2T_ADM(DPV, 4); else P2T_ADM(DPV, 8);  \
    } while (0)
            if (dp == 32) P2T_ADM_G(32); else if (dp == 64) P2T_ADM_G(64); else P2T_ADM_G(128);
#undef P2T_ADM_G
#undef P2T_ADM
            P2T_LAUNCH_CHECK();
            return P2T_OK;
        }
    }
#define P2T_AD(DPV, GHV)                                                                                                              \
    attn_decode_kernel<T, DPV, GHV><<<grid, 512, 0, s>>>((const T*)b.qb, kp, vtp, kg, vtg, kc->prompt_len, kc->step, kc->group, nh, nkv, G, \
                                                         kc->Tp, kc->G, c_exp, round_p, (T*)b.ao, QO, d)
#define P2T_AD_G(DPV)                                                                                                                 \
    do {                                                                                                                              \
        if (b.GH == 1) P2T_AD(DPV, 1); else if (b.GH == 2) P2T_AD(DPV, 2); else if (b.GH == 4) P2T_AD(DPV, 4); else P2T_AD(DPV, 8);      \
    } while (0)
    if (dp == 32) P2T_AD_G(32); else if (dp == 64) P2T_AD_G(64); else P2T_AD_G(128);
#undef P2T_AD_G
#undef P2T_AD
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

int check_cache(const p2t_llama_config* c, const p2t_kv_cache* kc, const char* who) {
    P2T_REQUIRE(kc && kc->k_prompt && kc->vt_prompt && kc->k_gen && kc->vt_gen && kc->prompt_len && kc->step, "%s: null cache field", who);
    P2T_REQUIRE(kc->B0 > 0 && kc->group > 0 && kc->Tp > 0 && kc->G > 0 && kc->Tp % 64 == 0 && kc->G % 64 == 0,
                "%s: cache capacities must be positive multiples of 64 (Tp %d, G %d)", who, kc->Tp, kc->G);
    P2T_REQUIRE(c->heads % c->kv_heads == 0 && c->head_dim % 2 == 0 && c->head_dim <= 128, "%s: unsupported head shape", who);
    return P2T_OK;
}

}  // namespace

// prefill hook of llama_forward_impl (towers.hip): layer l's rotated keys / values -> prompt segment
int llama_kv_store(const p2t_llama_config* c, const p2t_kv_cache* kc, int layer, const void* k, const void* v, int B, int T, hipStream_t s) {
    const int dp = head_dim_padded(c->head_dim), nkv = c->kv_heads;
    const size_t e = dtype_size(c->dtype), per = (size_t)kc->B0 * nkv * kc->Tp * dp;
    char* kd = (char*)kc->k_prompt + per * layer * e;
    P2T_CHECK_HIP(hipMemcpy2DAsync(kd, (size_t)kc->Tp * dp * e, k, (size_t)T * dp * e, (size_t)T * dp * e, (size_t)B * nkv, hipMemcpyDeviceToDevice, s));
    const dim3 grid((unsigned)ceil_div(T, 64), (unsigned)(B * nkv));
    if (c->dtype == P2T_BF16)
        v_transpose_store_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)v, (bf16_t*)kc->vt_prompt + per * layer, T, dp, kc->Tp);
    else
        v_transpose_store_kernel<float><<<grid, 256, 0, s>>>((const float*)v, (float*)kc->vt_prompt + per * layer, T, dp, kc->Tp);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

}  // namespace p2t

using namespace p2t;

extern "C" int p2t_compact_rows(const float* x, const int64_t* mask, int B, int T, int H, float* out, int64_t* out_mask, int32_t* lens,
                                int32_t* scratch, p2t_stream stream) {
    P2T_REQUIRE(x && mask && out && out_mask && lens && scratch && B > 0 && T > 0 && H > 0 && H % 4 == 0 && x != out,
                "p2t_compact_rows: bad arguments (H must be a multiple of 4, out of place)");
    hipStream_t s = (hipStream_t)stream;
    P2T_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * T * H, s));
    compact_index_kernel<<<B, 256, 0, s>>>(mask, T, scratch, out_mask, lens);
    P2T_LAUNCH_CHECK();
    compact_copy_kernel<<<dim3((unsigned)T, (unsigned)B), 256, 0, s>>>(x, scratch, T, H, out);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" size_t p2t_llama_prefill_workspace_bytes(const p2t_llama_config* cfg, int B, int T) {
    if (!cfg || B <= 0 || T <= 0) return 0;
    return p2t_llama_workspace_bytes(cfg, B, T) + sizeof(float) * (size_t)B * T * cfg->hidden + 512;
}

extern "C" int p2t_llama_prefill(const p2t_llama_config* c, const p2t_llama_weights* w, const float* inputs_embeds, const int64_t* mask, int B,
                                 int T, const p2t_kv_cache* cache, float* last_hidden, void* workspace, size_t workspace_bytes,
                                 p2t_stream stream) {
    P2T_REQUIRE(c && w && inputs_embeds && mask && last_hidden && workspace && B > 0 && T > 0, "p2t_llama_prefill: null/empty argument");
    P2T_TRY(check_cache(c, cache, "p2t_llama_prefill"));
    P2T_REQUIRE(cache->B0 == B && T <= cache->Tp, "p2t_llama_prefill: cache holds %d rows x %d tokens, the prompt batch is %d x %d", cache->B0,
                cache->Tp, B, T);
    P2T_REQUIRE(workspace_bytes >= p2t_llama_prefill_workspace_bytes(c, B, T), "p2t_llama_prefill: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Arena ar(workspace, workspace_bytes);
    float* out = (float*)ar.take(sizeof(float) * (size_t)B * T * c->hidden);
    void* rest = ar.take(p2t_llama_workspace_bytes(c, B, T));
    P2T_REQUIRE(!ar.overflow, "p2t_llama_prefill: workspace overflow");
    P2T_TRY(llama_forward_impl(c, w, nullptr, inputs_embeds, mask, B, T, c->n_layers, out, rest, p2t_llama_workspace_bytes(c, B, T), stream, nullptr,
                               cache));
    gather_last_kernel<<<B, 256, 0, s>>>(out, cache->prompt_len, T, c->hidden, last_hidden);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" size_t p2t_llama_decode_workspace_bytes(const p2t_llama_config* cfg, int BB, int Tp, int G) {
    if (!cfg || BB <= 0 || Tp <= 0 || G <= 0) return 0;
    return decode_plan(cfg, BB, Tp, G, nullptr, nullptr);
}

extern "C" int p2t_llama_decode_step(const p2t_llama_config* c, const p2t_llama_weights* w, const p2t_llama_layer_stream* ws_layers,
                                     const void* lm_head, int64_t ld_head, int lm_head_preshuffled, const p2t_kv_cache* cache, const float* x_in, void* logits, int64_t ld_logits, int flags, void* workspace,
                                     size_t workspace_bytes, p2t_stream stream) {
    P2T_REQUIRE(c && w && w->layers && w->final_norm_w && lm_head && x_in && logits && workspace, "p2t_llama_decode_step: null argument");
    const bool fuse_rope = !(flags & P2T_DECODE_NO_ROPE_FUSION);
    P2T_TRY(check_cache(c, cache, "p2t_llama_decode_step"));
    P2T_REQUIRE(!c->gemm_fp8 || c->dtype == P2T_BF16, "p2t_llama_decode_step: gemm_fp8 needs bf16 activations (dtype = P2T_BF16)");
    const int BB = cache->B0 * cache->group;
    const int dt = c->dtype;
    const int64_t H = c->hidden, F = c->ffn, Hp = round_up(H, 64), Fp = round_up(F, 64);
    const int d = c->head_dim, dp = head_dim_padded(d), nh = c->heads, nkv = c->kv_heads;
    const int64_t NQKV = (int64_t)(nh + 2 * nkv) * d, QO = round_up((int64_t)nh * d, 64);
    P2T_REQUIRE((lm_head_preshuffled || ld_head >= Hp) && ld_logits >= c->vocab, "p2t_llama_decode_step: ld_head %lld < %lld or ld_logits %lld < vocab", (long long)ld_head,
                (long long)Hp, (long long)ld_logits);
    P2T_REQUIRE(workspace_bytes >= decode_plan(c, BB, cache->Tp, cache->G, nullptr, nullptr), "p2t_llama_decode_step: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Arena ar(workspace, workspace_bytes);
    DecodeBuffers b;
    decode_plan(c, BB, cache->Tp, cache->G, &ar, &b);
    P2T_REQUIRE(!ar.overflow, "p2t_llama_decode_step: workspace overflow");
    const float* inv_freq = w->inv_freq;
    if (!inv_freq) {
        P2T_TRY(launch_inv_freq(b.inv_freq, d / 2, c->rope_theta, c->rope_llama3, c->rope_factor, c->rope_low_freq_factor, c->rope_high_freq_factor,
                                (float)c->rope_original_max_pos, s));
        inv_freq = b.inv_freq;
    }
    const float scale = 1.0f / sqrtf((float)d);
    const int l2s = dt == P2T_BF16;                      // as the prefill: scale * log2(e) folded into q for bf16 models
    const float q_fold = l2s ? scale * kLog2e : 1.0f, c_exp = l2s ? 1.0f : scale * kLog2e;
    P2T_CHECK_HIP(hipMemcpyAsync(b.x, x_in, sizeof(float) * (size_t)BB * H, hipMemcpyDeviceToDevice, s));
    P2T_CHECK_HIP(hipMemsetAsync(b.ao, 0, dtype_size(dt) * (size_t)BB * QO, s));
    const int64_t M = BB;
    // the step's projections: the weight-streaming kernel (gemm_skinny.hip) for bf16 models, the general GEMM otherwise
    // `pre`: the pre-shuffled stream copy of the same weight (p2t_preshuffle_w), when the caller built one
    auto gemm_nt = [&](const GemmArgs& a, hipStream_t st, const void* pre = nullptr) {
        const int r = launch_gemm_skinny(a.A, a.lda, pre ? pre : a.W, a.ldw, a.out, a.ldc, a.M, a.N, a.K, a.dtype, a.out_dtype, a.epilogue, st, pre != nullptr);
        return r == P2T_ERR_UNSUPPORTED ? p2t::gemm_nt(a, st) : r;
    };
    const bool stream_w = ws_layers && dt == P2T_BF16;
    // gemm_fp8 models (e4m3 weights AND GEMM operands, bf16 activations elsewhere): the same step with every
    // projection on the e4m3 stream (half the weight bytes of a step); rows are quantised by the RMSNorm / a quantise pass as in the prefill
    const int64_t Hq = round_up(H, 128), Fq = round_up(F, 128), QOq = round_up((int64_t)nh * d, 128);
    auto gemm8 = [&](const uint8_t* A, int64_t lda, const uint8_t* as, const void* W, const uint8_t* wsc, const void* pre, void* out, int64_t ldc,
                     int64_t N, int64_t K, int out_dtype, int epi, const SkinnyRope* ra, hipStream_t st) {
        const int r = launch_gemm_skinny_fp8(A, lda, as, pre ? pre : W, K, wsc, out, ldc, M, N, K, out_dtype, epi, ra, st, pre != nullptr);
        if (r != P2T_ERR_UNSUPPORTED || ra) return r;
        GemmArgs g{A, lda, W, K, nullptr, out, ldc, nullptr, M, N, K, P2T_FP8, out_dtype, epi, 0, 1, -1, 0.f, 0, 0};
        g.a_scale = as; g.w_scale = wsc;
        if (epi == P2T_EPI_STORE_F32) g.n_zero = (int)N;
        return p2t::gemm_nt(g, st);
    };
    for (int l = 0; c->gemm_fp8 && l < c->n_layers; ++l) {
        const p2t_llama_layer& L = w->layers[l];
        P2T_REQUIRE(L.qkv_ws && L.o_ws && L.gu_ws && L.down_ws, "p2t_llama_decode_step: gemm_fp8 needs the row scales of layer %d", l);
        P2T_REQUIRE(!L.q_norm_w == !L.k_norm_w, "p2t_llama_decode_step: q_norm_w and k_norm_w go together (layer %d)", l);
        const int fused_prefill = !L.q_norm_w && (d == 64 || d == 128);
        const p2t_llama_layer_stream* S = ws_layers ? ws_layers + l : nullptr;
        const size_t per_g = (size_t)BB * nkv * cache->G * dp;
        P2T_TRY(launch_rmsnorm_fp8_few(b.x, H, L.ln1_w, c->rms_norm_eps, b.hq, Hq, b.hs, M, H, s));
        bool roped = false;
        if (fused_prefill && fuse_rope) {
            SkinnyRope ra;
            ra.inv_freq = inv_freq; ra.prompt_len = cache->prompt_len; ra.step = cache->step; ra.group = cache->group;
            ra.nh = nh; ra.nkv = nkv; ra.d = d; ra.G = cache->G; ra.q_scale = q_fold;
            ra.q = b.qb; ra.k = (bf16_t*)cache->k_gen + per_g * l; ra.vt = (bf16_t*)cache->vt_gen + per_g * l;
            const int r = gemm8(b.hq, Hq, b.hs, L.qkv_w, L.qkv_ws, S ? S->qkv_w : nullptr, nullptr, 0, NQKV, Hq, dt, 0, &ra, s);
            if (r != P2T_ERR_UNSUPPORTED) { P2T_TRY(r); roped = true; }
        }
        if (!roped) {
            P2T_TRY(gemm8(b.hq, Hq, b.hs, L.qkv_w, L.qkv_ws, S ? S->qkv_w : nullptr, b.qkv, NQKV, NQKV, Hq, P2T_F32, P2T_EPI_STORE_F32, nullptr, s));
            const unsigned grid = (unsigned)ceil_div((int64_t)BB * (nh + 2 * nkv), 4);
            rope_append_kernel<bf16_t><<<grid, 256, 0, s>>>(b.qkv, NQKV, inv_freq, L.q_norm_w, L.k_norm_w, c->rms_norm_eps, cache->prompt_len, cache->step,
                                                            cache->group, (bf16_t*)b.qb, (bf16_t*)cache->k_gen + per_g * l, (bf16_t*)cache->vt_gen + per_g * l,
                                                            BB, nh, nkv, d, dp, cache->G, q_fold, d == 128 && !L.q_norm_w, !fused_prefill);
            P2T_LAUNCH_CHECK();
        }
        P2T_TRY(launch_attn_decode_t<bf16_t>(b, cache, l, BB, nh, nkv, d, dp, c_exp, 1, QO, s));
        P2T_TRY(launch_quant_rows_few(b.ao, dt, QO, M, (int64_t)nh * d, b.aoq, QOq, b.aos, s));
        P2T_TRY(gemm8(b.aoq, QOq, b.aos, L.o_w, L.o_ws, S ? S->o_w : nullptr, b.x, H, H, QOq, P2T_F32, P2T_EPI_RESID, nullptr, s));
        P2T_TRY(launch_rmsnorm_fp8_few(b.x, H, L.ln2_w, c->rms_norm_eps, b.hq, Hq, b.hs, M, H, s));
        P2T_TRY(gemm8(b.hq, Hq, b.hs, L.gu_w, L.gu_ws, S ? S->gu_w : nullptr, b.act, Fp, 2 * F, Hq, dt, P2T_EPI_SWIGLU, nullptr, s));
        P2T_TRY(launch_quant_rows_few(b.act, dt, Fp, M, F, b.actq, Fq, b.acts, s));
        P2T_TRY(gemm8(b.actq, Fq, b.acts, L.down_w, L.down_ws, S ? S->down_w : nullptr, b.x, H, H, Fq, P2T_F32, P2T_EPI_RESID, nullptr, s));
    }
    for (int l = 0; !c->gemm_fp8 && l < c->n_layers; ++l) {
        const p2t_llama_layer& L = w->layers[l];
        P2T_REQUIRE(!L.q_norm_w == !L.k_norm_w, "p2t_llama_decode_step: q_norm_w and k_norm_w go together (layer %d)", l);
        const int fused_prefill = !L.q_norm_w && (d == 64 || d == 128);
        P2T_TRY(launch_rmsnorm_few_rows(b.x, H, L.ln1_w, c->rms_norm_eps, b.h, Hp, M, H, dt, s));
        bool roped = false;
        if (fused_prefill && dt == P2T_BF16 && fuse_rope) {   // projection + rotation + cache append in one launch (SK_QKV_ROPE)
            const size_t per_g = (size_t)BB * nkv * cache->G * dp;
            SkinnyRope ra;
            ra.inv_freq = inv_freq; ra.prompt_len = cache->prompt_len; ra.step = cache->step; ra.group = cache->group;
            ra.nh = nh; ra.nkv = nkv; ra.d = d; ra.G = cache->G; ra.q_scale = q_fold;
            ra.q = b.qb; ra.k = (bf16_t*)cache->k_gen + per_g * l; ra.vt = (bf16_t*)cache->vt_gen + per_g * l;
            const int r = launch_gemm_skinny_qkv_rope(b.h, Hp, stream_w ? ws_layers[l].qkv_w : L.qkv_w, Hp, M, NQKV, Hp, ra, s, stream_w);
            if (r != P2T_ERR_UNSUPPORTED) { P2T_TRY(r); roped = true; }
        }
        if (!roped) {
            GemmArgs g1{b.h, Hp, L.qkv_w, Hp, nullptr, b.qkv, NQKV, nullptr, M, NQKV, Hp, dt, P2T_F32, P2T_EPI_STORE_F32, 0, -1, (int)NQKV, 0.f, 0, 0};
            P2T_TRY(gemm_nt(g1, s, stream_w ? ws_layers[l].qkv_w : nullptr));
            {
                const unsigned grid = (unsigned)ceil_div((int64_t)BB * (nh + 2 * nkv), 4);
                const size_t per_g = (size_t)BB * nkv * cache->G * dp;
                if (dt == P2T_BF16)
                    rope_append_kernel<bf16_t><<<grid, 256, 0, s>>>(b.qkv, NQKV, inv_freq, L.q_norm_w, L.k_norm_w, c->rms_norm_eps, cache->prompt_len,
                                                                    cache->step, cache->group, (bf16_t*)b.qb, (bf16_t*)cache->k_gen + per_g * l,
                                                                    (bf16_t*)cache->vt_gen + per_g * l, BB, nh, nkv, d, dp, cache->G, q_fold,
                                                                    d == 128 && !L.q_norm_w, !fused_prefill);
                else
                    rope_append_kernel<float><<<grid, 256, 0, s>>>(b.qkv, NQKV, inv_freq, L.q_norm_w, L.k_norm_w, c->rms_norm_eps, cache->prompt_len,
                                                                   cache->step, cache->group, (float*)b.qb, (float*)cache->k_gen + per_g * l,
                                                                   (float*)cache->vt_gen + per_g * l, BB, nh, nkv, d, dp, cache->G, q_fold,
                                                                   d == 128 && !L.q_norm_w, 0);
                P2T_LAUNCH_CHECK();
            }
        }
        if (dt == P2T_BF16) P2T_TRY(launch_attn_decode_t<bf16_t>(b, cache, l, BB, nh, nkv, d, dp, c_exp, 1, QO, s));
        else P2T_TRY(launch_attn_decode_t<float>(b, cache, l, BB, nh, nkv, d, dp, c_exp, 0, QO, s));
        GemmArgs g2{b.ao, QO, L.o_w, QO, nullptr, b.x, H, nullptr, M, H, QO, dt, P2T_F32, P2T_EPI_RESID, 0, -1, -1, 0.f, 0, 0};
        P2T_TRY(gemm_nt(g2, s, stream_w ? ws_layers[l].o_w : nullptr));
        P2T_TRY(launch_rmsnorm_few_rows(b.x, H, L.ln2_w, c->rms_norm_eps, b.h, Hp, M, H, dt, s));
        GemmArgs g3{b.h, Hp, L.gu_w, Hp, nullptr, b.act, Fp, nullptr, M, 2 * F, Hp, dt, dt, P2T_EPI_SWIGLU, 0, -1, -1, 0.f, 0, 0};
        P2T_TRY(gemm_nt(g3, s, stream_w ? ws_layers[l].gu_w : nullptr));
        GemmArgs g4{b.act, Fp, L.down_w, Fp, nullptr, b.x, H, nullptr, M, H, Fp, dt, P2T_F32, P2T_EPI_RESID, 0, -1, -1, 0.f, 0, 0};
        P2T_TRY(gemm_nt(g4, s, stream_w ? ws_layers[l].down_w : nullptr));
    }
    P2T_REQUIRE(!lm_head_preshuffled || dt == P2T_BF16, "p2t_llama_decode_step: pre-shuffled weights are a bf16 layout");
    // the stream-order copy is only readable by the skinny kernels (<= 64 rows): beyond that the general GEMM would read the
    // shuffled tiles as a row-major [vocab, Hp] matrix and return wrong logits without an error
    P2T_REQUIRE(!lm_head_preshuffled || M <= 64, "p2t_llama_decode_step: a pre-shuffled LM head serves at most 64 rows (got %lld): pass the row-major one",
                (long long)M);
    P2T_TRY(launch_rmsnorm_few_rows(b.x, H, w->final_norm_w, c->rms_norm_eps, b.h, Hp, M, H, dt, s));
    GemmArgs gh{b.h, Hp, lm_head, ld_head, nullptr, logits, ld_logits, nullptr, M, c->vocab, Hp, dt, dt, P2T_EPI_STORE, 0, -1, -1, 0.f, 0, 0};
    P2T_TRY(gemm_nt(gh, s, lm_head_preshuffled ? lm_head : nullptr));
    advance_kernel<<<1, 1, 0, s>>>(cache->step);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_greedy_select(const void* logits, int dtype, int64_t ld, int V, int BB, const int64_t* eos_ids, int n_eos, int64_t pad_id,
                                 int32_t* finished, int64_t* next_tokens, int64_t* out_tokens, int64_t ld_tokens, const int32_t* step, int G,
                                 p2t_stream stream) {
    P2T_REQUIRE(logits && finished && next_tokens && out_tokens && step && BB > 0 && V > 0 && ld >= V && G > 0 && ld_tokens >= G && n_eos >= 0 &&
                    (n_eos == 0 || eos_ids),
                "p2t_greedy_select: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == P2T_BF16)
        greedy_select_kernel<bf16_t><<<BB, 1024, 0, s>>>((const bf16_t*)logits, ld, V, eos_ids, n_eos, pad_id, finished, next_tokens, out_tokens,
                                                        ld_tokens, step, G);
    else
        greedy_select_kernel<float><<<BB, 1024, 0, s>>>((const float*)logits, ld, V, eos_ids, n_eos, pad_id, finished, next_tokens, out_tokens,
                                                       ld_tokens, step, G);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_kv_reorder(const p2t_llama_config* c, const p2t_kv_cache* cache, const int64_t* src_row, void* k_dst, void* vt_dst,
                              p2t_stream stream) {
    P2T_REQUIRE(c && src_row && k_dst && vt_dst, "p2t_kv_reorder: null argument");
    P2T_TRY(check_cache(c, cache, "p2t_kv_reorder"));
    P2T_REQUIRE(k_dst != cache->k_gen && vt_dst != cache->vt_gen, "p2t_kv_reorder: out of place only");
    const int BB = cache->B0 * cache->group, nkv = c->kv_heads, dp = head_dim_padded(c->head_dim);
    const unsigned grid = (unsigned)((int64_t)c->n_layers * BB * nkv);
    hipStream_t s = (hipStream_t)stream;
    if (c->dtype == P2T_BF16)
        kv_reorder_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)cache->k_gen, (const bf16_t*)cache->vt_gen, (bf16_t*)k_dst, (bf16_t*)vt_dst, src_row,
                                                       cache->step, BB, nkv, dp, cache->G);
    else
        kv_reorder_kernel<float><<<grid, 256, 0, s>>>((const float*)cache->k_gen, (const float*)cache->vt_gen, (float*)k_dst, (float*)vt_dst, src_row,
                                                     cache->step, BB, nkv, dp, cache->G);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_attention_decode(const void* q, const void* k_prompt, const void* vt_prompt, const void* k_gen, const void* vt_gen,
                                    const int32_t* prompt_len, const int32_t* step, int B0, int group, int nh, int nkv, int head_dim, int Tp, int G,
                                    float scale, int log2_scores, int dtype, int use_mfma, void* out, int64_t ld_out, p2t_stream stream) {
    P2T_REQUIRE(q && out && (dtype == P2T_F32 || dtype == P2T_BF16) && nkv > 0 && nh % nkv == 0 && ld_out >= (int64_t)nh * head_dim,
                "p2t_attention_decode: bad arguments");
    p2t_llama_config c{};
    c.heads = nh; c.kv_heads = nkv; c.head_dim = head_dim; c.dtype = dtype;
    p2t_kv_cache kc{const_cast<void*>(k_prompt), const_cast<void*>(vt_prompt), const_cast<void*>(k_gen), const_cast<void*>(vt_gen), prompt_len,
                    const_cast<int32_t*>(step), B0, group, Tp, G};
    P2T_TRY(check_cache(&c, &kc, "p2t_attention_decode"));
    const int BB = B0 * group, dp = head_dim_padded(head_dim);
    DecodeBuffers b{};
    attn_decode_plan(nh, nkv, &b.ZC, &b.GH);
    b.qb = const_cast<void*>(q);
    b.ao = out;
    hipStream_t s = (hipStream_t)stream;
    const float c_exp = log2_scores ? 1.0f : scale * kLog2e;
    if (dtype == P2T_BF16) return launch_attn_decode_t<bf16_t>(b, &kc, 0, BB, nh, nkv, head_dim, dp, c_exp, 1, ld_out, s, use_mfma != 0);
    P2T_REQUIRE(use_mfma <= 0, "p2t_attention_decode: the matrix-pipe kernel is bf16 only");
    return launch_attn_decode_t<float>(b, &kc, 0, BB, nh, nkv, head_dim, dp, c_exp, 0, ld_out, s);
}
