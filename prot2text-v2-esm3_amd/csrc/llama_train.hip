// Stage-2 (SFT) training step through the FROZEN decoder: the forward that keeps what the backward needs (the "tape"), and
// the backward that carries d loss / d hidden back to d loss / d inputs_embeds -- the gradient the adapter is trained with
// (reference scripts/train_instruct.py:192-213 `loss = model(**batch).loss; loss.backward()` with the decoder's parameters
// frozen; models/modeling_esm2llama_instruct.py:195-215).  No weight gradients: every GEMM here is a dX GEMM,
// dX[M, K] = dY[M, N] . W[N, K], run on the same NT kernels as the forward with W^T stored once ([K, N], N-contiguous; the
// decoder is frozen, so the transposes are built when the engine is packed).
//
// What torch autograd does through HF LlamaDecoderLayer (transformers/models/llama/modeling_llama.py:296-324), op by op:
//   x2 = x1 + down(silu(gate(h2)) * up(h2)),  h2 = rmsnorm(x1)         -> swiglu_bwd, rmsnorm_bwd
//   x1 = x  + o(attn(rope(q(h1)), rope(k(h1)), v(h1))), h1 = rmsnorm(x) -> attn_bwd_{dq,dkv}, rope_bwd_pack, rmsnorm_bwd
// The residual-stream gradient stays fp32 throughout (one buffer, accumulated in place); GEMM operands are `dtype`.
#include "common.h"
#include "epilogue.h"
#include "kernels.h"

namespace p2t {

// ---------------------------------------------------------------------------------------------
// y = x * rsqrt(mean(x^2) + eps) * w   ->   g (+)= r * (w dy) - x r^3 mean(w dy x).  One wave per row, two passes over the
// row (the second hits L2).  dy: f32 or `dtype`.
template <typename Tdy>
__global__ void __launch_bounds__(256) rmsnorm_bwd_kernel(const float* __restrict__ x, int64_t ld_x, const float* __restrict__ w, float eps,
                                                          const Tdy* __restrict__ dy, int64_t ld_dy, float* __restrict__ g, int64_t ld_g,
                                                          int64_t rows, int cols, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ld_x;
    const Tdy* dr = dy + row * ld_dy;
    float ss = 0.f, dot = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
        float xv[4], dv[4], wv[4];
        load4(xr + c, xv); load4(dr + c, dv); load4(w + c, wv);
#pragma unroll
        for (int j = 0; j < 4; ++j) { ss = fmaf(xv[j], xv[j], ss); dot = fmaf(dv[j] * wv[j], xv[j], dot); }
    }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float r = rsqrtf(ss / (float)cols + eps);
    const float k = r * r * r * dot / (float)cols;
    float* gr = g + row * ld_g;
    for (int c = lane * 4; c < cols; c += 256) {
        float xv[4], dv[4], wv[4], o[4];
        load4(xr + c, xv); load4(dr + c, dv); load4(w + c, wv);
        if (accumulate) load4(gr + c, o); else o[0] = o[1] = o[2] = o[3] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += r * (dv[j] * wv[j]) - xv[j] * k;
        store4(gr + c, o);
    }
}

int launch_rmsnorm_bwd(const float* x, int64_t ld_x, const float* w, float eps, const void* dy, int64_t ld_dy, int dy_dtype, float* g,
                       int64_t ld_g, int64_t rows, int64_t cols, int accumulate, hipStream_t s) {
    P2T_REQUIRE(cols % 4 == 0 && ld_x % 4 == 0 && ld_dy % 4 == 0 && ld_g % 4 == 0, "rmsnorm backward: cols / strides must be multiples of 4");
    const dim3 grid((unsigned)ceil_div(rows, 4));
    if (dy_dtype == P2T_BF16)
        rmsnorm_bwd_kernel<bf16_t><<<grid, 256, 0, s>>>(x, ld_x, w, eps, (const bf16_t*)dy, ld_dy, g, ld_g, rows, (int)cols, accumulate);
    else
        rmsnorm_bwd_kernel<float><<<grid, 256, 0, s>>>(x, ld_x, w, eps, (const float*)dy, ld_dy, g, ld_g, rows, (int)cols, accumulate);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

// dst[m, c] = (Tdst)src[m, c], c < cols; 0 up to ld_dst: the fp32 residual gradient as a GEMM operand (with its K padding)
template <typename Ts, typename Td>
__global__ void __launch_bounds__(256) cast_rows_kernel(const Ts* __restrict__ src, int64_t ld_src, Td* __restrict__ dst, int64_t ld_dst, int64_t rows,
                                                        int cols) {
    const int per = (int)(ld_dst / 4);
    const int64_t n4 = rows * per, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int64_t m = i / per;
        const int c = (int)(i - m * per) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c + 3 < cols) {
            load4(src + m * ld_src + c, v);
        } else {
            for (int j = 0; j < 4; ++j)
                if (c + j < cols) v[j] = to_f32(src[m * ld_src + c + j]);
        }
        store4(dst + m * ld_dst + c, v);
    }
}

int launch_cast_rows(const void* src, int sd, int64_t ld_src, void* dst, int dd, int64_t ld_dst, int64_t rows, int64_t cols, hipStream_t s) {
    P2T_REQUIRE(ld_dst % 4 == 0 && ld_src % 4 == 0 && cols <= ld_dst && cols <= ld_src, "cast_rows: strides must be multiples of 4");
    if (rows == 0) return P2T_OK;
    const int64_t n4 = rows * (ld_dst / 4);
    const unsigned grid = (unsigned)(ceil_div(n4, 256) < 4096 ? ceil_div(n4, 256) : 4096);
    if (sd == P2T_F32 && dd == P2T_BF16) cast_rows_kernel<float, bf16_t><<<grid, 256, 0, s>>>((const float*)src, ld_src, (bf16_t*)dst, ld_dst, rows, (int)cols);
    else if (sd == P2T_F32 && dd == P2T_F32) cast_rows_kernel<float, float><<<grid, 256, 0, s>>>((const float*)src, ld_src, (float*)dst, ld_dst, rows, (int)cols);
    else if (sd == P2T_BF16 && dd == P2T_F32) cast_rows_kernel<bf16_t, float><<<grid, 256, 0, s>>>((const bf16_t*)src, ld_src, (float*)dst, ld_dst, rows, (int)cols);
    else P2T_REQUIRE(false, "cast_rows: dtypes %d -> %d", sd, dd);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

// ---------------------------------------------------------------------------------------------
// SwiGLU on the interleaved pre-activations the gate/up GEMM writes with a plain store: 64-column block jb of gu holds
// gate[32 jb .. +31] then up[32 jb .. +31] (the row order of gu_w, include/p2t_hip.h p2t_llama_layer).
//   forward : act[m, f] = silu(g) * u
//   backward: d_gu = (d_act * u * sigma(g) (1 + g (1 - sigma(g))),  d_act * silu(g))   in the same interleaved layout
template <typename T, bool BWD>
__global__ void __launch_bounds__(256) swiglu_gu_kernel(const T* __restrict__ gu, int64_t ld_gu, const T* __restrict__ d_act, int64_t ld_da,
                                                        T* __restrict__ out, int64_t ld_out, int64_t M, int F, int Fo) {
    // Fo: columns written per row of `out` in the forward (F rounded up to the next GEMM's K padding: zeros beyond F)
    const int per = (BWD ? F : Fo) / 4;
    const int64_t n4 = M * (int64_t)per, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int64_t m = i / per;
        const int f = (int)(i - m * per) * 4;
        if (!BWD && f >= F) {
            const float z[4] = {0.f, 0.f, 0.f, 0.f};
            store4(out + m * ld_out + f, z);
            continue;
        }
        const int col = (f >> 5) * 64 + (f & 31);
        float g[4], u[4];
        load4(gu + m * ld_gu + col, g);
        load4(gu + m * ld_gu + col + 32, u);
        if (!BWD) {
            float a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = silu_for<T>(g[j]) * u[j];
            store4(out + m * ld_out + f, a);
        } else {
            float da[4], dg[4], du[4];
            load4(d_act + m * ld_da + f, da);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sg = 1.0f / (1.0f + expf(-g[j]));
                dg[j] = da[j] * u[j] * (sg * (1.0f + g[j] * (1.0f - sg)));
                du[j] = da[j] * (g[j] * sg);
            }
            store4(out + m * ld_out + col, dg);
            store4(out + m * ld_out + col + 32, du);
        }
    }
}

template <bool BWD>
static int launch_swiglu_gu(const void* gu, int64_t ld_gu, const void* d_act, int64_t ld_da, void* out, int64_t ld_out, int64_t M, int64_t F,
                            int dtype, hipStream_t s) {
    P2T_REQUIRE(F % 32 == 0 && ld_gu % 4 == 0 && ld_out % 4 == 0, "swiglu: F must be a multiple of 32");
    const int64_t Fo = BWD ? F : (round_up(F, 64) < ld_out ? round_up(F, 64) : ld_out);
    const int64_t n4 = M * (Fo / 4);
    const unsigned grid = (unsigned)(ceil_div(n4, 256) < 4096 ? ceil_div(n4, 256) : 4096);
    if (dtype == P2T_BF16)
        swiglu_gu_kernel<bf16_t, BWD><<<grid, 256, 0, s>>>((const bf16_t*)gu, ld_gu, (const bf16_t*)d_act, ld_da, (bf16_t*)out, ld_out, M, (int)F, (int)Fo);
    else
        swiglu_gu_kernel<float, BWD><<<grid, 256, 0, s>>>((const float*)gu, ld_gu, (const float*)d_act, ld_da, (float*)out, ld_out, M, (int)F, (int)Fo);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}
int launch_swiglu_from_gu(const void* gu, int64_t ld_gu, void* act, int64_t ld_act, int64_t M, int64_t F, int dtype, hipStream_t s) {
    return launch_swiglu_gu<false>(gu, ld_gu, nullptr, 0, act, ld_act, M, F, dtype, s);
}

// ---------------------------------------------------------------------------------------------
// Attention backward, exact-fp32 arithmetic on `T` operands (the counterpart of attn_simple.hip; an MFMA form is the next
// step for the bf16 path).  lse [B, nh, T]: natural-log sum-exp of the EFFECTIVE logits c_s * q.k (c_s = the softmax scale, or
// ln 2 when q carries scale * log2 e: kernels.h attention()), +inf for a row without a visible key.
//   P = exp(c_s q.k - lse),  D_i = dO_i . O_i,  dS = P o (dO v^T - D),  dq = c_s dS k,  dk = c_s dS^T q,  dv = P^T dO
// dq kernel: one wave per (b, h, query); also writes D.  dk/dv kernel: one wave per (b, kv head, key), summing over the
// query heads of the group (GQA: repeat_kv shares one key / value head, modeling_llama.py:181-188).
template <typename T>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                          const T* __restrict__ o, int64_t ld_o, const T* __restrict__ d_o, int64_t ld_do,
                                                          const float* __restrict__ lse, const uint8_t* __restrict__ key_mask,
                                                          const int32_t* __restrict__ kv_end, float* __restrict__ dq, float* __restrict__ D,
                                                          int seq, int nh, int nkv, int d, int dp, float c_s, int causal) {
    __shared__ float s_q[4][128];
    __shared__ float s_do[4][128];
    __shared__ float s_ds[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
    if (i >= seq) return;
    const int hk = h / (nh / nkv);
    const T* qrow = q + ((int64_t)(b * nh + h) * seq + i) * dp;
    const T* kbase = k + ((int64_t)(b * nkv + hk) * seq) * dp;
    const T* vbase = v + ((int64_t)(b * nkv + hk) * seq) * dp;
    const T* orow = o + ((int64_t)b * seq + i) * ld_o + (int64_t)h * d;
    const T* dorow = d_o + ((int64_t)b * seq + i) * ld_do + (int64_t)h * d;
    float dd = 0.f;
    for (int c = lane; c < dp; c += 64) {
        const float qv = to_f32(qrow[c]), dv = c < d ? to_f32(dorow[c]) : 0.f;
        s_q[w][c] = qv;
        s_do[w][c] = dv;
        if (c < d) dd = fmaf(dv, to_f32(orow[c]), dd);
    }
    dd = wave_sum(dd);
    const float l = lse[(int64_t)(b * nh + h) * seq + i];
    if (lane == 0) D[(int64_t)(b * nh + h) * seq + i] = dd;
    int end = kv_end[b];
    if (causal) end = min(end, i + 1);
    float acc0 = 0.f, acc1 = 0.f;
    for (int j0 = 0; j0 < end; j0 += 64) {
        const int j = j0 + lane;
        float ds = 0.f;
        if (j < end && key_mask[(int64_t)b * seq + j]) {
            const T* kr = kbase + (int64_t)j * dp;
            const T* vr = vbase + (int64_t)j * dp;
            float dot = 0.f, dpj = 0.f;
            for (int c = 0; c < d; c += 4) {
                float kv[4], vv[4];
                load4(kr + c, kv);
                load4(vr + c, vv);
#pragma unroll
                for (int e = 0; e < 4; ++e) { dot = fmaf(s_q[w][c + e], kv[e], dot); dpj = fmaf(s_do[w][c + e], vv[e], dpj); }
            }
            ds = expf(c_s * dot - l) * (dpj - dd);               // l = +inf (no visible key): p = 0
        }
        s_ds[w][lane] = ds;                                     // same-wave LDS exchange: program order suffices
        const int nj = min(64, end - j0);
        if (lane < d) {
            for (int jj = 0; jj < nj; ++jj) acc0 = fmaf(s_ds[w][jj], to_f32(kbase[(int64_t)(j0 + jj) * dp + lane]), acc0);
        }
        if (lane + 64 < d) {
            for (int jj = 0; jj < nj; ++jj) acc1 = fmaf(s_ds[w][jj], to_f32(kbase[(int64_t)(j0 + jj) * dp + lane + 64]), acc1);
        }
    }
    float* out = dq + ((int64_t)(b * nh + h) * seq + i) * dp;
    if (lane < dp) out[lane] = lane < d ? c_s * acc0 : 0.f;            // (dp = 32: a row is half a wave wide)
    if (dp > 64) out[lane + 64] = lane + 64 < d ? c_s * acc1 : 0.f;
}

template <typename T>
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                           const T* __restrict__ d_o, int64_t ld_do, const float* __restrict__ lse,
                                                           const float* __restrict__ D, const uint8_t* __restrict__ key_mask,
                                                           float* __restrict__ dk, float* __restrict__ dv, int seq, int nh, int nkv, int d,
                                                           int dp, float c_s, int causal) {
    __shared__ float s_k[4][128];
    __shared__ float s_v[4][128];
    __shared__ float s_p[4][64];
    __shared__ float s_ds[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + w, hk = blockIdx.y, b = blockIdx.z;
    if (j >= seq) return;
    float* dkr = dk + ((int64_t)(b * nkv + hk) * seq + j) * dp;
    float* dvr = dv + ((int64_t)(b * nkv + hk) * seq + j) * dp;
    if (!key_mask[(int64_t)b * seq + j]) {                       // a masked key is read by no query
        if (lane < dp) { dkr[lane] = 0.f; dvr[lane] = 0.f; }
        if (dp > 64) { dkr[lane + 64] = 0.f; dvr[lane + 64] = 0.f; }
        return;
    }
    const T* krow = k + ((int64_t)(b * nkv + hk) * seq + j) * dp;
    const T* vrow = v + ((int64_t)(b * nkv + hk) * seq + j) * dp;
    for (int c = lane; c < dp; c += 64) { s_k[w][c] = to_f32(krow[c]); s_v[w][c] = to_f32(vrow[c]); }
    const int rep = nh / nkv;
    float ak0 = 0.f, ak1 = 0.f, av0 = 0.f, av1 = 0.f;
    for (int r = 0; r < rep; ++r) {
        const int h = hk * rep + r;
        const T* qbase = q + ((int64_t)(b * nh + h) * seq) * dp;
        const float* lrow = lse + (int64_t)(b * nh + h) * seq;
        const float* Drow = D + (int64_t)(b * nh + h) * seq;
        for (int i0 = causal ? (j & ~63) : 0; i0 < seq; i0 += 64) {
            const int i = i0 + lane;
            float p = 0.f, ds = 0.f;
            if (i < seq && (!causal || i >= j)) {
                const T* qr = qbase + (int64_t)i * dp;
                const T* dor = d_o + ((int64_t)b * seq + i) * ld_do + (int64_t)h * d;
                float dot = 0.f, dpj = 0.f;
                for (int c = 0; c < d; c += 4) {
                    float qv[4], ov[4];
                    load4(qr + c, qv);
                    load4(dor + c, ov);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { dot = fmaf(qv[e], s_k[w][c + e], dot); dpj = fmaf(ov[e], s_v[w][c + e], dpj); }
                }
                p = expf(c_s * dot - lrow[i]);
                ds = p * (dpj - Drow[i]);
            }
            s_p[w][lane] = p;
            s_ds[w][lane] = ds;
            const int ni = min(64, seq - i0);
            if (lane < d) {
                for (int ii = 0; ii < ni; ++ii) {
                    av0 = fmaf(s_p[w][ii], to_f32(d_o[((int64_t)b * seq + i0 + ii) * ld_do + (int64_t)h * d + lane]), av0);
                    ak0 = fmaf(s_ds[w][ii], to_f32(qbase[(int64_t)(i0 + ii) * dp + lane]), ak0);
                }
            }
            if (lane + 64 < d) {
                for (int ii = 0; ii < ni; ++ii) {
                    av1 = fmaf(s_p[w][ii], to_f32(d_o[((int64_t)b * seq + i0 + ii) * ld_do + (int64_t)h * d + lane + 64]), av1);
                    ak1 = fmaf(s_ds[w][ii], to_f32(qbase[(int64_t)(i0 + ii) * dp + lane + 64]), ak1);
                }
            }
        }
    }
    if (lane < dp) {
        dkr[lane] = lane < d ? c_s * ak0 : 0.f;
        dvr[lane] = lane < d ? av0 : 0.f;
    }
    if (dp > 64) {
        dkr[lane + 64] = lane + 64 < d ? c_s * ak1 : 0.f;
        dvr[lane + 64] = lane + 64 < d ? av1 : 0.f;
    }
}

int launch_attn_bwd_mfma(const void* q, const void* k, const void* v, const void* o, int64_t ld_o, const void* d_o, int64_t ld_do, const float* lse,
                         const uint8_t* key_mask, const int32_t* kv_info, float* dq, float* dk, float* dv, float* D, int B, int T, int nh, int nkv,
                         int d, int dp, int causal, hipStream_t s);       // attn_bwd_mfma.hip

// c_s: the factor between q.k and the logits (the softmax scale, or ln 2 in the log2_scores form).  use_mfma: -1 auto (the MFMA
// kernels for bf16 + log2_scores + head_dim 64 / 128), 0 the exact kernels, 1 require MFMA.
int launch_attn_bwd(const void* q, const void* k, const void* v, const void* o, int64_t ld_o, const void* d_o, int64_t ld_do, const float* lse,
                    const uint8_t* key_mask, const int32_t* kv_info, float* dq, float* dk, float* dv, float* D, int B, int T, int nh, int nkv,
                    int d, int dp, float c_s, int causal, int dtype, hipStream_t s, int log2_scores = 0, int use_mfma = 0) {
    P2T_REQUIRE(q && k && v && o && d_o && lse && key_mask && kv_info && dq && dk && dv && D, "attention backward: null argument");
    P2T_REQUIRE(d % 4 == 0 && d <= 128 && (dp == 32 || dp == 64 || dp == 128) && d <= dp && nh % nkv == 0 && ld_o % 4 == 0 && ld_do % 4 == 0,
                "attention backward: unsupported shape d=%d dp=%d heads %d/%d", d, dp, nh, nkv);
    if (dtype == P2T_BF16 && log2_scores && use_mfma != 0) {
        const int rc = launch_attn_bwd_mfma(q, k, v, o, ld_o, d_o, ld_do, lse, key_mask, kv_info, dq, dk, dv, D, B, T, nh, nkv, d, dp, causal, s);
        if (rc != P2T_ERR_UNSUPPORTED) return rc;
    }
    P2T_REQUIRE(use_mfma != 1, "attention backward: the MFMA kernels need bf16, log2_scores and head_dim 64 / 128");
    const dim3 gq((unsigned)ceil_div(T, 4), (unsigned)nh, (unsigned)B), gk((unsigned)ceil_div(T, 4), (unsigned)nkv, (unsigned)B);
    if (dtype == P2T_BF16) {
        attn_bwd_dq_kernel<bf16_t><<<gq, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, ld_o, (const bf16_t*)d_o,
                                                      ld_do, lse, key_mask, kv_info, dq, D, T, nh, nkv, d, dp, c_s, causal);
        attn_bwd_dkv_kernel<bf16_t><<<gk, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, ld_do, lse, D, key_mask,
                                                       dk, dv, T, nh, nkv, d, dp, c_s, causal);
    } else {
        attn_bwd_dq_kernel<float><<<gq, 256, 0, s>>>((const float*)q, (const float*)k, (const float*)v, (const float*)o, ld_o, (const float*)d_o, ld_do,
                                                     lse, key_mask, kv_info, dq, D, T, nh, nkv, d, dp, c_s, causal);
        attn_bwd_dkv_kernel<float><<<gk, 256, 0, s>>>((const float*)q, (const float*)k, (const float*)v, (const float*)d_o, ld_do, lse, D, key_mask, dk,
                                                      dv, T, nh, nkv, d, dp, c_s, causal);
    }
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward of the head split + rotation (+ the scale folded into q): dq [B, nh, T, dp], dk, dv [B, nkv, T, dp] (f32) ->
// d_qkv `T` [B*T, ld] rows = [q heads | k heads | v heads] in the NATURAL order of q_proj / k_proj / v_proj.
// Forward o1 = a1 c - a2 s, o2 = a2 c + a1 s  =>  da1 = do1 c + do2 s, da2 = do2 c - do1 s.  One wave per (token, head).
template <typename T>
__global__ void __launch_bounds__(256) rope_bwd_pack_kernel(const float* __restrict__ dq, const float* __restrict__ dk, const float* __restrict__ dv,
                                                            const float* __restrict__ cs, T* __restrict__ out, int64_t ld, int64_t rows, int seq,
                                                            int nh, int nkv, int d, int dp, float q_scale) {
    const int lane = threadIdx.x & 63, heads = nh + 2 * nkv, half = d / 2;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t bt = row / heads;
    const int hh = (int)(row - bt * heads);
    const int b = (int)(bt / seq), t = (int)(bt - (int64_t)b * seq);
    T* dst = out + bt * ld + (int64_t)hh * d;
    if (lane >= half) return;
    if (hh < nh + nkv) {
        const bool is_q = hh < nh;
        const float* src = is_q ? dq + (((int64_t)b * nh + hh) * seq + t) * dp : dk + (((int64_t)b * nkv + (hh - nh)) * seq + t) * dp;
        const float sc = is_q ? q_scale : 1.0f;
        const float o1 = src[lane] * sc, o2 = src[lane + half] * sc;
        const float c = cs[(int64_t)t * d + lane], s = cs[(int64_t)t * d + half + lane];
        dst[lane] = from_f32<T>(o1 * c + o2 * s);
        dst[lane + half] = from_f32<T>(o2 * c - o1 * s);
    } else {
        const float* src = dv + (((int64_t)b * nkv + (hh - nh - nkv)) * seq + t) * dp;
        dst[lane] = from_f32<T>(src[lane]);
        dst[lane + half] = from_f32<T>(src[lane + half]);
    }
}

int launch_rope_bwd_pack(const float* dq, const float* dk, const float* dv, const float* cs, void* out, int64_t ld, int B, int T, int nh, int nkv,
                         int d, int dp, float q_scale, int dtype, hipStream_t s) {
    P2T_REQUIRE(d % 2 == 0 && d <= 128 && dp >= d, "rope backward: head_dim %d unsupported", d);
    const int64_t rows = (int64_t)B * T * (nh + 2 * nkv);
    const dim3 grid((unsigned)ceil_div(rows, 4));
    if (dtype == P2T_BF16)
        rope_bwd_pack_kernel<bf16_t><<<grid, 256, 0, s>>>(dq, dk, dv, cs, (bf16_t*)out, ld, rows, T, nh, nkv, d, dp, q_scale);
    else
        rope_bwd_pack_kernel<float><<<grid, 256, 0, s>>>(dq, dk, dv, cs, (float*)out, ld, rows, T, nh, nkv, d, dp, q_scale);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

// ---------------------------------------------------------------------------------------------
// d loss / d logits of p2t_cross_entropy_shifted: row (b, t) with a counted target y = labels[b, t+1]:
// (softmax(logits) - onehot(y)) / count; every other row and the padding columns: 0.  One block per row.
template <typename T>
__global__ void __launch_bounds__(256) ce_bwd_rows_kernel(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels, int seq, int V,
                                                          int64_t ignore_index, const int32_t* __restrict__ count, T* __restrict__ dl, int64_t ld_d,
                                                          int cols_d) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const int t = (int)(row % seq);
    T* dr = dl + row * ld_d;
    int64_t label = ignore_index;
    if (t + 1 < seq) label = labels[row + 1];
    if (label == ignore_index || label < 0 || label >= V) {
        for (int c = threadIdx.x; c < cols_d; c += 256) dr[c] = from_f32<T>(0.f);
        return;
    }
    const T* x = logits + row * ld;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) m = fmaxf(m, to_f32(x[c]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) sum += expf(to_f32(x[c]) - m);
    sum = block_sum<4>(sum, red);
    const float inv = 1.0f / (sum * (float)count[0]), invc = 1.0f / (float)count[0];
    for (int c = threadIdx.x; c < cols_d; c += 256) {
        float g = 0.f;
        if (c < V) g = expf(to_f32(x[c]) - m) * inv - (c == (int)label ? invc : 0.f);
        dr[c] = from_f32<T>(g);
    }
}

// dst[dst_pos[r], :H] = src[src_pos[r], :H] for r < min(*n_dst, *n_src), f32 -> f32: the backward of p2t_scatter_rows'
// boolean-mask assignment with the roles of the two position lists swapped (rows of dst not listed stay as they are).
__global__ void __launch_bounds__(256) gather_rows_f32_kernel(float* __restrict__ dst, int64_t ld_dst, const int32_t* __restrict__ dst_pos,
                                                              const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ src_pos,
                                                              const int32_t* __restrict__ n_dst, const int32_t* __restrict__ n_src, int H) {
    const int n = min(n_dst[0], n_src[0]);
    for (int r = blockIdx.x; r < n; r += gridDim.x) {
        float* d = dst + (int64_t)dst_pos[r] * ld_dst;
        const float* s = src + (int64_t)src_pos[r] * ld_src;
        for (int c = threadIdx.x; c < H; c += 256) d[c] = s[c];
    }
}

// ---------------------------------------------------------------------------------------------
// tape = what the backward reads again, per layer; carved from one caller-owned buffer.
size_t llama_tape_plan(const p2t_llama_config* c, int B, int T, void* base, size_t bytes, LlamaTape* tape) {
    const size_t e = dtype_size(c->dtype);
    const int64_t M = (int64_t)B * T, H = c->hidden, F = c->ffn;
    const int d = c->head_dim, dp = head_dim_padded(d), nh = c->heads, nkv = c->kv_heads;
    const int64_t QO = round_up((int64_t)nh * d, 64);
    Arena a(base, base ? bytes : (~(size_t)0 >> 1));
    LlamaTape t{};
    t.n_layers = c->n_layers;
    for (int l = 0; l < c->n_layers && l < LlamaTape::kMaxLayers; ++l) {
        LlamaTapeLayer& L = t.layer[l];
        L.x_in = (float*)a.take(sizeof(float) * (size_t)M * H);
        L.x_mid = (float*)a.take(sizeof(float) * (size_t)M * H);
        L.q = a.take(e * (size_t)B * nh * T * dp);
        L.k = a.take(e * (size_t)B * nkv * T * dp);
        L.v = a.take(e * (size_t)B * nkv * T * dp);
        L.lse = (float*)a.take(sizeof(float) * (size_t)B * nh * T);
        L.ao = a.take(e * (size_t)M * QO);
        L.gu = a.take(e * (size_t)M * 2 * F);
    }
    t.x_last = (float*)a.take(sizeof(float) * (size_t)M * H);
    t.overflow = a.overflow;
    if (tape) *tape = t;
    return a.off + 256;
}

}  // namespace p2t

using namespace p2t;

namespace {
struct TrainBuffers {
    uint8_t* key_mask; int32_t* kv_info; float* inv_freq; float* cs;
    void* g16; void* d_act; void* d_gu; float* d_h; void* d_ao; float* dq; float* dk; float* dv; float* D; void* d_qkv; void* fix;
};
size_t train_plan(const p2t_llama_config* c, int B, int T, Arena* ar, TrainBuffers* b) {
    const size_t e = dtype_size(c->dtype);
    const int64_t M = (int64_t)B * T, H = c->hidden, F = c->ffn, Hp = round_up(H, 64), Fp = round_up(F, 64);
    const int d = c->head_dim, dp = head_dim_padded(d), nh = c->heads, nkv = c->kv_heads;
    const int64_t QO = round_up((int64_t)nh * d, 64), NQp = round_up((int64_t)(nh + 2 * nkv) * d, 64);
    Arena local(nullptr, ~(size_t)0 >> 1);
    Arena& a = ar ? *ar : local;
    TrainBuffers t;
    t.key_mask = (uint8_t*)a.take((size_t)M);
    t.kv_info = (int32_t*)a.take(sizeof(int32_t) * 2 * B);
    t.inv_freq = (float*)a.take(sizeof(float) * (d / 2 + 1));
    t.cs = (float*)a.take(sizeof(float) * (size_t)T * d);
    t.g16 = a.take(e * (size_t)M * Hp);
    t.d_act = a.take(e * (size_t)M * Fp);
    t.d_gu = a.take(e * (size_t)M * 2 * F);
    t.d_h = (float*)a.take(sizeof(float) * (size_t)M * H);
    t.d_ao = a.take(e * (size_t)M * QO);
    t.dq = (float*)a.take(sizeof(float) * (size_t)B * nh * T * dp);
    t.dk = (float*)a.take(sizeof(float) * (size_t)B * nkv * T * dp);
    t.dv = (float*)a.take(sizeof(float) * (size_t)B * nkv * T * dp);
    t.D = (float*)a.take(sizeof(float) * (size_t)B * nh * T);
    t.d_qkv = a.take(e * (size_t)M * NQp);
    t.fix = a.take(gemm_fix_workspace_bytes());
    if (b) *b = t;
    return a.off + 256;
}
}  // namespace

extern "C" size_t p2t_llama_tape_bytes(const p2t_llama_config* cfg, int B, int T) {
    if (!cfg || B <= 0 || T <= 0 || cfg->n_layers > LlamaTape::kMaxLayers) return 0;
    return llama_tape_plan(cfg, B, T, nullptr, 0, nullptr);
}

extern "C" size_t p2t_llama_train_workspace_bytes(const p2t_llama_config* cfg, int B, int T) {
    if (!cfg || B <= 0 || T <= 0) return 0;
    const size_t fwd = p2t_llama_workspace_bytes(cfg, B, T), bwd = train_plan(cfg, B, T, nullptr, nullptr);
    return fwd > bwd ? fwd : bwd;
}

extern "C" int p2t_llama_train_forward(const p2t_llama_config* c, const p2t_llama_weights* w, const float* inputs_embeds, const int64_t* mask,
                                       int B, int T, float* out, void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes,
                                       p2t_stream stream) {
    P2T_REQUIRE(c && w && inputs_embeds && mask && out && tape && workspace && B > 0 && T > 0, "p2t_llama_train_forward: null/empty argument");
    P2T_REQUIRE(!c->gemm_fp8, "p2t_llama_train_forward: the training path runs the GEMMs in the model dtype (gemm_fp8 = 0)");
    P2T_REQUIRE(c->n_layers <= LlamaTape::kMaxLayers, "p2t_llama_train_forward: more than %d layers", LlamaTape::kMaxLayers);
    P2T_REQUIRE(tape_bytes >= p2t_llama_tape_bytes(c, B, T), "p2t_llama_train_forward: tape too small (%zu < %zu)", tape_bytes, p2t_llama_tape_bytes(c, B, T));
    LlamaTape t;
    llama_tape_plan(c, B, T, tape, tape_bytes, &t);
    P2T_REQUIRE(!t.overflow, "p2t_llama_train_forward: tape overflow");
    return llama_forward_impl(c, w, nullptr, inputs_embeds, mask, B, T, c->n_layers, out, workspace, workspace_bytes, stream, &t);
}

extern "C" int p2t_llama_train_backward(const p2t_llama_config* c, const p2t_llama_weights* w, const p2t_llama_layer_t* wT, const int64_t* mask,
                                        int B, int T, const float* d_out, const void* tape, size_t tape_bytes, float* d_inputs_embeds,
                                        void* workspace, size_t workspace_bytes, p2t_stream stream) {
    P2T_REQUIRE(c && w && wT && mask && d_out && tape && d_inputs_embeds && workspace && B > 0 && T > 0, "p2t_llama_train_backward: null/empty argument");
    P2T_REQUIRE(!c->gemm_fp8 && c->n_layers <= LlamaTape::kMaxLayers, "p2t_llama_train_backward: unsupported configuration");
    P2T_REQUIRE(w->layers && w->final_norm_w, "p2t_llama_train_backward: missing weights");
    P2T_REQUIRE(tape_bytes >= p2t_llama_tape_bytes(c, B, T) && workspace_bytes >= p2t_llama_train_workspace_bytes(c, B, T),
                "p2t_llama_train_backward: tape / workspace too small");
    hipStream_t s = (hipStream_t)stream;
    LlamaTape t;
    llama_tape_plan(c, B, T, const_cast<void*>(tape), tape_bytes, &t);
    Arena ar(workspace, workspace_bytes);
    TrainBuffers b;
    train_plan(c, B, T, &ar, &b);
    P2T_REQUIRE(!ar.overflow && !t.overflow, "p2t_llama_train_backward: workspace overflow");
    const int dt = c->dtype;
    const int64_t M = (int64_t)B * T, H = c->hidden, F = c->ffn, Hp = round_up(H, 64), Fp = round_up(F, 64);
    const int d = c->head_dim, dp = head_dim_padded(d), nh = c->heads, nkv = c->kv_heads;
    const int64_t NQKV = (int64_t)(nh + 2 * nkv) * d, NQp = round_up(NQKV, 64), QO = round_up((int64_t)nh * d, 64);
    // the statistics of the forward: mask bytes / ends, rotary table (the same launches as llama_forward_impl)
    P2T_TRY(launch_mask_prepare(nullptr, mask, B, T, -1, 0, b.key_mask, b.kv_info, nullptr, s));
    const float* inv_freq = w->inv_freq;
    if (!inv_freq) {
        P2T_TRY(launch_inv_freq(b.inv_freq, d / 2, c->rope_theta, c->rope_llama3, c->rope_factor, c->rope_low_freq_factor, c->rope_high_freq_factor,
                                (float)c->rope_original_max_pos, s));
        inv_freq = b.inv_freq;
    }
    P2T_TRY(launch_rope_table(inv_freq, T, d / 2, b.cs, s));
    const float scale = 1.0f / sqrtf((float)d);
    const int l2s = dt == P2T_BF16;                             // as the forward: q carries scale * log2 e, the logits are ln 2 * q.k
    const float c_s = l2s ? kLn2 : scale, q_fold = l2s ? scale * kLog2e : 1.0f;
    float* g = d_inputs_embeds;                                 // the residual-stream gradient, fp32, accumulated in place
    // final RMSNorm (modeling_llama.py:405)
    P2T_TRY(launch_rmsnorm_bwd(t.x_last, H, w->final_norm_w, c->rms_norm_eps, d_out, H, P2T_F32, g, H, M, H, 0, s));
    P2T_CHECK_HIP(hipMemsetAsync(b.fix, 0, gemm_fix_header_bytes(), s));
    if (NQp != NQKV) P2T_CHECK_HIP(hipMemsetAsync(b.d_qkv, 0, dtype_size(dt) * (size_t)M * NQp, s));      // the K padding of the QKV dX GEMM
    unsigned epoch = 0;
    auto dx_gemm = [&](const void* A, int64_t lda, const void* WT, int64_t K, int64_t N, void* out, int64_t ldc, int out_dtype, int epi) {
        // dX[M, N] = A[M, K] . WT[N, K]^T with WT = the forward weight transposed ([N = forward K][ld >= forward N])
        GemmArgs a{A, lda, WT, round_up(K, 64), nullptr, out, ldc, nullptr, M, N, round_up(K, 64), dt, out_dtype, epi, 0, -1, -1, 0.f, 0, 0};
        a.fix_ws = b.fix; a.fix_bytes = gemm_fix_workspace_bytes(); a.fix_epoch = ++epoch;
        return gemm_nt(a, s);
    };
    for (int l = c->n_layers - 1; l >= 0; --l) {
        const p2t_llama_layer& L = w->layers[l];
        const p2t_llama_layer_t& LT = wT[l];
        const LlamaTapeLayer& S = t.layer[l];
        P2T_REQUIRE(!L.q_norm_w, "p2t_llama_train_backward: per-head q/k RMSNorm (Qwen3) has no backward on this path yet");
        P2T_REQUIRE(LT.qkv_wT && LT.o_wT && LT.gu_wT && LT.down_wT, "p2t_llama_train_backward: missing transposed weights of layer %d", l);
        // ---- MLP branch: x2 = x1 + down(silu(g) u)
        P2T_TRY(launch_cast_rows(g, P2T_F32, H, b.g16, dt, Hp, M, H, s));
        P2T_TRY(dx_gemm(b.g16, Hp, LT.down_wT, H, F, b.d_act, Fp, dt, P2T_EPI_STORE));
        P2T_TRY(launch_swiglu_gu<true>(S.gu, 2 * F, b.d_act, Fp, b.d_gu, 2 * F, M, F, dt, s));
        P2T_TRY(dx_gemm(b.d_gu, 2 * F, LT.gu_wT, 2 * F, H, b.d_h, H, P2T_F32, P2T_EPI_STORE_F32));
        P2T_TRY(launch_rmsnorm_bwd(S.x_mid, H, L.ln2_w, c->rms_norm_eps, b.d_h, H, P2T_F32, g, H, M, H, 1, s));
        // ---- attention branch: x1 = x + o(attn(...))
        P2T_TRY(launch_cast_rows(g, P2T_F32, H, b.g16, dt, Hp, M, H, s));
        P2T_TRY(dx_gemm(b.g16, Hp, LT.o_wT, H, (int64_t)nh * d, b.d_ao, QO, dt, P2T_EPI_STORE));
        P2T_TRY(launch_attn_bwd(S.q, S.k, S.v, S.ao, QO, b.d_ao, QO, S.lse, b.key_mask, b.kv_info, b.dq, b.dk, b.dv, b.D, B, T, nh, nkv, d, dp, c_s, 1,
                                dt, s, l2s, -1));
        P2T_TRY(launch_rope_bwd_pack(b.dq, b.dk, b.dv, b.cs, b.d_qkv, NQp, B, T, nh, nkv, d, dp, q_fold, dt, s));
        P2T_TRY(dx_gemm(b.d_qkv, NQp, LT.qkv_wT, NQKV, H, b.d_h, H, P2T_F32, P2T_EPI_STORE_F32));
        P2T_TRY(launch_rmsnorm_bwd(S.x_in, H, L.ln1_w, c->rms_norm_eps, b.d_h, H, P2T_F32, g, H, M, H, 1, s));
    }
    return P2T_OK;
}

extern "C" int p2t_attention_backward(const void* q, const void* k, const void* v, const void* o, int64_t ld_o, const void* d_o, int64_t ld_do,
                                      const float* lse, const uint8_t* key_mask, const int32_t* kv_info, float* dq, float* dk, float* dv,
                                      float* D_scratch, int B, int T, int nh, int nkv, int d, int dp, float scale, int causal, int dtype,
                                      int log2_scores, int use_mfma, p2t_stream stream) {
    return launch_attn_bwd(q, k, v, o, ld_o, d_o, ld_do, lse, key_mask, kv_info, dq, dk, dv, D_scratch, B, T, nh, nkv, d, dp,
                           log2_scores ? kLn2 : scale, causal, dtype, (hipStream_t)stream, log2_scores, use_mfma);
}

extern "C" int p2t_rmsnorm_backward(const float* x, int64_t ld_x, const float* w, float eps, const void* dy, int64_t ld_dy, int dy_dtype, float* dx,
                                    int64_t ld_dx, int64_t rows, int64_t cols, int accumulate, p2t_stream stream) {
    P2T_REQUIRE(x && w && dy && dx && rows >= 0 && cols > 0, "p2t_rmsnorm_backward: bad arguments");
    if (rows == 0) return P2T_OK;
    return launch_rmsnorm_bwd(x, ld_x, w, eps, dy, ld_dy, dy_dtype, dx, ld_dx, rows, cols, accumulate, (hipStream_t)stream);
}

extern "C" int p2t_cross_entropy_shifted_backward(const void* logits, int64_t ld, int dtype, const int64_t* labels, int B, int T, int V,
                                                  int64_t ignore_index, const int32_t* count, void* d_logits, int64_t ld_d, p2t_stream stream) {
    P2T_REQUIRE(logits && labels && count && d_logits && B > 0 && T > 0 && V > 0 && ld >= V && ld_d >= V, "p2t_cross_entropy_shifted_backward: bad arguments");
    P2T_REQUIRE(dtype == P2T_F32 || dtype == P2T_BF16, "p2t_cross_entropy_shifted_backward: unsupported dtype %d", dtype);
    const int64_t M = (int64_t)B * T;
    const int cols_d = (int)(round_up(V, 64) < ld_d ? round_up(V, 64) : ld_d);          // the K padding of the LM-head dX GEMM is zeroed
    if (dtype == P2T_BF16)
        ce_bwd_rows_kernel<bf16_t><<<(unsigned)M, 256, 0, (hipStream_t)stream>>>((const bf16_t*)logits, ld, labels, T, V, ignore_index, count,
                                                                                 (bf16_t*)d_logits, ld_d, cols_d);
    else
        ce_bwd_rows_kernel<float><<<(unsigned)M, 256, 0, (hipStream_t)stream>>>((const float*)logits, ld, labels, T, V, ignore_index, count,
                                                                                (float*)d_logits, ld_d, cols_d);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

// ---------------------------------------------------------------------------------------------
// dst (+)= keep(seed, m * K + c) ? src / (1 - p) : 0: the LoRA branch's input dropout (peft lora_dropout, train_instruct.py:158) and,
// with the same seed, its backward (the mask is regenerated, never stored).
template <typename Ts, typename Td>
__global__ void __launch_bounds__(256) dropout_rows_kernel(const Ts* __restrict__ src, int64_t ld_src, Td* __restrict__ dst, int64_t ld_dst, int64_t M,
                                                           int K, float p, float scale, uint64_t seed, int accumulate) {
    const int64_t n = M * (int64_t)K, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int64_t m = i / K;
        const int c = (int)(i - m * K);
        float x = to_f32(src[m * ld_src + c]);
        x = (p > 0.f && !dropout_keep(seed, i, p)) ? 0.f : x * scale;
        Td* d = dst + m * ld_dst + c;
        *d = from_f32<Td>(accumulate ? to_f32(*d) + x : x);
    }
}

extern "C" int p2t_dropout_rows(const void* src, int src_dtype, int64_t ld_src, void* dst, int dst_dtype, int64_t ld_dst, int64_t M, int64_t K, float p,
                                uint64_t seed, int accumulate, p2t_stream stream) {
    P2T_REQUIRE(src && dst && M >= 0 && K > 0 && ld_src >= K && ld_dst >= K && p >= 0.f && p < 1.f, "p2t_dropout_rows: bad arguments");
    if (M == 0) return P2T_OK;
    const float scale = 1.0f / (1.0f - p);
    const int64_t n = M * K;
    const unsigned grid = (unsigned)(ceil_div(n, 256) < 8192 ? ceil_div(n, 256) : 8192);
    hipStream_t s = (hipStream_t)stream;
#define P2T_DROP(TS, TD) dropout_rows_kernel<TS, TD><<<grid, 256, 0, s>>>((const TS*)src, ld_src, (TD*)dst, ld_dst, M, (int)K, p, scale, seed, accumulate)
    if (src_dtype == P2T_BF16 && dst_dtype == P2T_BF16) P2T_DROP(bf16_t, bf16_t);
    else if (src_dtype == P2T_F32 && dst_dtype == P2T_BF16) P2T_DROP(float, bf16_t);
    else if (src_dtype == P2T_BF16 && dst_dtype == P2T_F32) P2T_DROP(bf16_t, float);
    else if (src_dtype == P2T_F32 && dst_dtype == P2T_F32) P2T_DROP(float, float);
    else { set_error("p2t_dropout_rows: unsupported dtypes %d -> %d", src_dtype, dst_dtype); return P2T_ERR_ARG; }
#undef P2T_DROP
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

extern "C" int p2t_swiglu_gu(const void* gu, int64_t ld_gu, const void* d_act, int64_t ld_da, void* out, int64_t ld_out, int64_t M, int64_t F, int dtype,
                             p2t_stream stream) {
    P2T_REQUIRE(gu && out && M >= 0 && F > 0 && (dtype == P2T_F32 || dtype == P2T_BF16), "p2t_swiglu_gu: bad arguments");
    if (M == 0) return P2T_OK;
    if (d_act) return launch_swiglu_gu<true>(gu, ld_gu, d_act, ld_da, out, ld_out, M, F, dtype, (hipStream_t)stream);
    return launch_swiglu_gu<false>(gu, ld_gu, nullptr, 0, out, ld_out, M, F, dtype, (hipStream_t)stream);
}

extern "C" int p2t_rope_backward_pack(const float* dq, const float* dk, const float* dv, const float* inv_freq, float* cos_sin_scratch, void* d_qkv,
                                      int64_t ld, int B, int T, int nh, int nkv, int d, int dp, float q_scale, int dtype, p2t_stream stream) {
    P2T_REQUIRE(dq && dk && dv && inv_freq && cos_sin_scratch && d_qkv && B > 0 && T > 0 && ld >= (int64_t)(nh + 2 * nkv) * d,
                "p2t_rope_backward_pack: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    P2T_TRY(launch_rope_table(inv_freq, T, d / 2, cos_sin_scratch, s));
    return launch_rope_bwd_pack(dq, dk, dv, cos_sin_scratch, d_qkv, ld, B, T, nh, nkv, d, dp, q_scale, dtype, s);
}

extern "C" int p2t_gather_rows_f32(float* dst, int64_t ld_dst, const int32_t* dst_pos, const float* src, int64_t ld_src, const int32_t* src_pos,
                                   const int32_t* n_dst, const int32_t* n_src, int64_t max_rows, int H, p2t_stream stream) {
    P2T_REQUIRE(dst && dst_pos && src && src_pos && n_dst && n_src && max_rows > 0 && H > 0 && ld_dst >= H && ld_src >= H, "p2t_gather_rows_f32: bad arguments");
    const unsigned grid = (unsigned)(max_rows < 4096 ? max_rows : 4096);
    gather_rows_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(dst, ld_dst, dst_pos, src, ld_src, src_pos, n_dst, n_src, H);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}
