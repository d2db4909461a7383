// LayerNorm / RMSNorm / row L2-normalise.  HBM-bound: one wavefront per row, the row held in
// registers (16 B per lane per load), wave-shuffle reductions, no LDS, no block barrier.
//   LayerNorm: HF EsmLayer pre-LN + emb_layer_norm_after (transformers/models/esm/modeling_esm.py:418,429,480,518,529,553)
//   RMSNorm:   HF LlamaRMSNorm (transformers/models/llama/modeling_llama.py:62-67)
//   L2 rows:   torch.nn.functional.normalize (reference models/modeling_esm2llama_instruct.py:67,
//              scripts/train_contrast.py:354,365)
#include "common.h"
#include "kernels.h"

namespace p2t {

// NV = float4 loads per lane; covers cols <= NV*256.
template <int NV, typename Tout, bool RMS>
__global__ void __launch_bounds__(256) norm_kernel(const float* __restrict__ x, int64_t ld_x, const float* __restrict__ w,
                                                   const float* __restrict__ b, float eps, Tout* __restrict__ y,
                                                   int64_t ld_y, int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ld_x;
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            // the fp32 stream is not read again before hundreds of MB of GEMM traffic: a non-temporal load keeps its 168 MB
            // from evicting the next GEMM's operand panels out of the L2s (+0.3 % step, same-box A/B)
            typedef float f4nt __attribute__((ext_vector_type(4)));
            const f4nt t = __builtin_nontemporal_load(reinterpret_cast<const f4nt*>(xr + c));
            v[i][0] = t[0]; v[i][1] = t[1]; v[i][2] = t[2]; v[i][3] = t[3];
        } else {
            v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
        }
        s += RMS ? (v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3])
                 : (v[i][0] + v[i][1] + v[i][2] + v[i][3]);
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / (float)cols + eps);
    } else {
        mean = s / (float)cols;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < cols) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[i][j] - mean;
                    q += d * d;
                }
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / (float)cols + eps);
    }
    Tout* yr = y + row * ld_y;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            float wv[4], o[4];
            load4(w + c, wv);
            if (RMS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = wv[j] * (v[i][j] * rstd);
            } else {
                float bv[4];
                load4(b + c, bv);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * wv[j] + bv[j];
            }
            store4(yr + c, o);
        } else if (c < ld_y) {
            const float z[4] = {0.f, 0.f, 0.f, 0.f};
            store4(yr + c, z);
        }
    }
}

template <typename Tout, bool RMS>
static int launch_norm_t(const float* x, int64_t ld_x, const float* w, const float* b, float eps, Tout* y, int64_t ld_y,
                         int64_t rows, int64_t cols, hipStream_t s) {
    const dim3 grid((unsigned)ceil_div(rows, 4));
    const int64_t span = ld_y > cols ? ld_y : cols;
#define P2T_NORM_CASE(NV)                                                                                   \
    if (span <= (NV) * 256) {                                                                               \
        norm_kernel<NV, Tout, RMS><<<grid, 256, 0, s>>>(x, ld_x, w, b, eps, y, ld_y, rows, (int)cols);      \
        P2T_LAUNCH_CHECK();                                                                                 \
        return P2T_OK;                                                                                      \
    }
    P2T_NORM_CASE(1) P2T_NORM_CASE(2) P2T_NORM_CASE(4) P2T_NORM_CASE(8) P2T_NORM_CASE(10) P2T_NORM_CASE(16)
    P2T_NORM_CASE(32)
#undef P2T_NORM_CASE
    set_error("norm: %lld columns exceed the 8192 supported", (long long)cols);
    return P2T_ERR_UNSUPPORTED;
}

// A handful of rows (one decode step: rows = sequences that generate): one BLOCK per row, the row's and the weight's 16-byte pieces
// all requested before anything is waited for -- one round trip to memory instead of the wave-per-row kernel's two (9.2 -> 4 us
// for 8 x 4096, 65 of them per generated token).  Same arithmetic as norm_kernel<.., RMS> except for the order of the sum.
template <int NV, typename Tout>
__global__ void __launch_bounds__(256) rmsnorm_rows_kernel(const float* __restrict__ x, int64_t ld_x, const float* __restrict__ w, float eps,
                                                           Tout* __restrict__ y, int64_t ld_y, int cols) {
    __shared__ float red[4];
    const float* xr = x + (int64_t)blockIdx.x * ld_x;
    float v[NV][4], wv[NV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 256 + threadIdx.x) * 4;
        if (c < cols) { load4(xr + c, v[i]); load4(w + c, wv[i]); }
        else { v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f; wv[i][0] = wv[i][1] = wv[i][2] = wv[i][3] = 0.f; }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    const float rstd = rsqrtf(block_sum<4>(s, red) / (float)cols + eps);
    Tout* yr = y + (int64_t)blockIdx.x * ld_y;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 256 + threadIdx.x) * 4;
        if (c < ld_y) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = c < cols ? wv[i][j] * (v[i][j] * rstd) : 0.f;
            store4(yr + c, o);
        }
    }
}

template <typename Tout>
static int launch_rmsnorm_rows(const float* x, int64_t ld_x, const float* w, float eps, Tout* y, int64_t ld_y, int64_t rows, int64_t cols, hipStream_t s) {
    const int64_t span = ld_y > cols ? ld_y : cols;
#define P2T_RROWS_CASE(NV)                                                                                            \
    if (span <= (NV) * 1024) {                                                                                        \
        rmsnorm_rows_kernel<NV, Tout><<<(unsigned)rows, 256, 0, s>>>(x, ld_x, w, eps, y, ld_y, (int)cols);            \
        P2T_LAUNCH_CHECK();                                                                                           \
        return P2T_OK;                                                                                                \
    }
    P2T_RROWS_CASE(1) P2T_RROWS_CASE(2) P2T_RROWS_CASE(4) P2T_RROWS_CASE(8)
#undef P2T_RROWS_CASE
    return P2T_ERR_UNSUPPORTED;
}

int launch_layernorm(const float* x, int64_t ld_x, const float* w, const float* b, float eps, void* y, int64_t ld_y,
                     int64_t rows, int64_t cols, int out_dtype, hipStream_t s) {
    if (rows == 0) return P2T_OK;
    if (out_dtype == P2T_BF16) return launch_norm_t<bf16_t, false>(x, ld_x, w, b, eps, (bf16_t*)y, ld_y, rows, cols, s);
    return launch_norm_t<float, false>(x, ld_x, w, b, eps, (float*)y, ld_y, rows, cols, s);
}
int launch_rmsnorm(const float* x, int64_t ld_x, const float* w, float eps, void* y, int64_t ld_y, int64_t rows,
                   int64_t cols, int out_dtype, hipStream_t s) {
    if (rows == 0) return P2T_OK;
    if (out_dtype == P2T_BF16) return launch_norm_t<bf16_t, true>(x, ld_x, w, nullptr, eps, (bf16_t*)y, ld_y, rows, cols, s);
    return launch_norm_t<float, true>(x, ld_x, w, nullptr, eps, (float*)y, ld_y, rows, cols, s);
}

// the decode step's RMSNorm (a few rows): block per row, one round trip; falls back to the wave-per-row kernel beyond 8192 columns
int launch_rmsnorm_few_rows(const float* x, int64_t ld_x, const float* w, float eps, void* y, int64_t ld_y, int64_t rows, int64_t cols, int out_dtype,
                            hipStream_t s) {
    if (rows == 0) return P2T_OK;
    int r = out_dtype == P2T_BF16 ? launch_rmsnorm_rows<bf16_t>(x, ld_x, w, eps, (bf16_t*)y, ld_y, rows, cols, s)
                                  : launch_rmsnorm_rows<float>(x, ld_x, w, eps, (float*)y, ld_y, rows, cols, s);
    return r == P2T_ERR_UNSUPPORTED ? launch_rmsnorm(x, ld_x, w, eps, y, ld_y, rows, cols, out_dtype, s) : r;
}

// ---------------------------------------------------------------------------------------------
// y = x / max(||x||, eps), one wave per row, second pass re-reads the row (L1/L2 resident).
template <typename Tin, typename Tout>
__global__ void __launch_bounds__(256) l2norm_kernel(const Tin* __restrict__ x, int64_t ld_x, Tout* __restrict__ y,
                                                     int64_t ld_y, float* __restrict__ inv_norm, int64_t rows, int cols,
                                                     float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const Tin* xr = x + row * ld_x;
    float s = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
        float v[4];
        load4(xr + c, v);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    s = wave_sum(s);
    const float inv = 1.0f / fmaxf(sqrtf(s), eps);
    if (inv_norm && lane == 0) inv_norm[row] = inv;
    Tout* yr = y + row * ld_y;
    for (int c = lane * 4; c < ld_y; c += 256) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < cols) {
            load4(xr + c, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= inv;
        }
        store4(yr + c, v);
    }
}

// dx = (dy - y (dy . y)) / max(||x||, eps)
__global__ void __launch_bounds__(256) l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dx, int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * cols;
    const float* gr = dy + row * cols;
    float s = 0.f, d = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
        float v[4], g[4];
        load4(xr + c, v);
        load4(gr + c, g);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        d += v[0] * g[0] + v[1] * g[1] + v[2] * g[2] + v[3] * g[3];
    }
    s = wave_sum(s);
    d = wave_sum(d);
    const float inv = 1.0f / fmaxf(sqrtf(s), eps);
    const float dot = d * inv;          // dy . y
    for (int c = lane * 4; c < cols; c += 256) {
        float v[4], g[4], o[4];
        load4(xr + c, v);
        load4(gr + c, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (g[j] - v[j] * inv * dot) * inv;
        store4(dx + row * cols + c, o);
    }
}

int launch_l2norm(const void* x, int in_dtype, int64_t ld_x, void* y, int out_dtype, int64_t ld_y, float* inv_norm,
                  int64_t rows, int64_t cols, float eps, hipStream_t s) {
    if (rows == 0) return P2T_OK;
    P2T_REQUIRE(cols % 4 == 0 && ld_x % 4 == 0 && ld_y % 4 == 0, "l2norm: cols/ld must be multiples of 4");
    const dim3 grid((unsigned)ceil_div(rows, 4));
    if (in_dtype == P2T_BF16 && out_dtype == P2T_BF16)
        l2norm_kernel<bf16_t, bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, ld_x, (bf16_t*)y, ld_y, inv_norm, rows, (int)cols, eps);
    else if (in_dtype == P2T_F32 && out_dtype == P2T_F32)
        l2norm_kernel<float, float><<<grid, 256, 0, s>>>((const float*)x, ld_x, (float*)y, ld_y, inv_norm, rows, (int)cols, eps);
    else if (in_dtype == P2T_BF16 && out_dtype == P2T_F32)
        l2norm_kernel<bf16_t, float><<<grid, 256, 0, s>>>((const bf16_t*)x, ld_x, (float*)y, ld_y, inv_norm, rows, (int)cols, eps);
    else
        l2norm_kernel<float, bf16_t><<<grid, 256, 0, s>>>((const float*)x, ld_x, (bf16_t*)y, ld_y, inv_norm, rows, (int)cols, eps);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

}  // namespace p2t

using namespace p2t;

extern "C" int p2t_layernorm(const float* x, int64_t ld_x, const float* w, const float* b, float eps, void* y,
                             int64_t ld_y, int64_t rows, int64_t cols, int out_dtype, p2t_stream stream) {
    P2T_REQUIRE(x && w && b && y && cols > 0 && cols % 4 == 0 && ld_x % 4 == 0 && ld_y % 4 == 0 && ld_y >= cols,
                "p2t_layernorm: bad arguments (cols and strides must be multiples of 4)");
    return launch_layernorm(x, ld_x, w, b, eps, y, ld_y, rows, cols, out_dtype, (hipStream_t)stream);
}
extern "C" int p2t_rmsnorm(const float* x, int64_t ld_x, const float* w, float eps, void* y, int64_t ld_y, int64_t rows,
                           int64_t cols, int out_dtype, p2t_stream stream) {
    P2T_REQUIRE(x && w && y && cols > 0 && cols % 4 == 0 && ld_x % 4 == 0 && ld_y % 4 == 0 && ld_y >= cols,
                "p2t_rmsnorm: bad arguments (cols and strides must be multiples of 4)");
    return launch_rmsnorm(x, ld_x, w, eps, y, ld_y, rows, cols, out_dtype, (hipStream_t)stream);
}
extern "C" int p2t_l2norm_rows(const float* x, float* y, float* inv_norm, int64_t rows, int64_t cols, float eps,
                               p2t_stream stream) {
    P2T_REQUIRE(x && y && cols > 0, "p2t_l2norm_rows: bad arguments");
    return launch_l2norm(x, P2T_F32, cols, y, P2T_F32, cols, inv_norm, rows, cols, eps, (hipStream_t)stream);
}
extern "C" int p2t_l2norm_rows_backward(const float* x, const float* dy, float* dx, int64_t rows, int64_t cols,
                                        float eps, p2t_stream stream) {
    P2T_REQUIRE(x && dy && dx && cols > 0 && cols % 4 == 0, "p2t_l2norm_rows_backward: bad arguments");
    if (rows == 0) return P2T_OK;
    l2norm_bwd_kernel<<<dim3((unsigned)ceil_div(rows, 4)), 256, 0, (hipStream_t)stream>>>(x, dy, dx, rows, (int)cols, eps);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}
