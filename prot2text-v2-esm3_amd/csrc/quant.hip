// fp8 (OCP e4m3fn) quantisation of GEMM operands for BASELINE.json configs[4] ("fp8 weights, CDNA4 fp8 MFMA").
//
// Scheme: every ROW of a GEMM operand (a token's activations, an output channel's weights) gets
// one power-of-two scale stored as an E8M0 byte  E = 127 + e,  2^e = the smallest power of two with amax / 2^e <= 448
// (the e4m3 maximum), and its elements are stored as e4m3fn( x * 2^-e ), round-to-nearest-even (v_cvt_pk_fp8_f32).
// The scaling itself is exact (power of two), so the only rounding is the e4m3 one, and the numpy oracle reproduces
// bytes and scale bytes bit for bit (oracle/p2t_oracle.py quant_rows_e4m3).  The E8M0 bytes are what the block-scaled
// MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) takes as its scale operands: the hardware applies 2^(Ea-127) * 2^(Ew-127).
//
//   quant_rows_kernel     bf16 / f32 [rows, ld_x] -> fp8 [rows, ld_q] (+ E8M0 [rows]); columns [cols, ld_q) zeroed
//   norm_fp8_kernel       LayerNorm / RMSNorm of the f32 residual stream written straight as fp8 (+ E8M0), the GEMM
//                         operand the next projection reads: no bf16 intermediate, no extra pass
#include "common.h"
#include "kernels.h"

namespace p2t {

// biased E8M0 exponent of the row scale from the row's absolute maximum (bit-exact integer rule, mirrored in numpy):
// amax = (1 + f) 2^ea;  amax / 448 = (1 + f) / 1.75 * 2^(ea - 8)  ->  e = ea - 8 + (f > 0.75)
__device__ __forceinline__ int e8m0_of_amax(float amax) {
    const unsigned u = __float_as_uint(amax);
    const int ea = (int)((u >> 23) & 0xFF) - 127;
    const int e = ea - 8 + ((u & 0x7FFFFF) > 0x600000 ? 1 : 0);
    const int E = e + 127;
    return amax > 0.f ? (E < 1 ? 1 : (E > 254 ? 254 : E)) : 127;
}
__device__ __forceinline__ float pow2_neg(int E) {            // 2^-(E - 127), exact
    return __uint_as_float((unsigned)(254 - E) << 23);
}
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    int r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (unsigned)r;
}

// one wave per row, two passes over the row (the second one hits L1 / L2)
template <typename Tin>
__global__ void __launch_bounds__(256) quant_rows_kernel(const Tin* __restrict__ x, int64_t ld_x, int64_t rows, int cols,
                                                         uint8_t* __restrict__ q, int64_t ld_q, uint8_t* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const Tin* xr = x + row * ld_x;
    float amax = 0.f;
    for (int c = lane * 8; c < cols; c += 512) {
        float v[8];
        if (c + 8 <= cols) {
            if constexpr (sizeof(Tin) == 2) { load8(xr + c, v); } else { float a[4], b[4]; load4(xr + c, a); load4(xr + c + 4, b);
                for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; } }
        } else {
            for (int j = 0; j < 8; ++j) v[j] = c + j < cols ? to_f32(xr[c + j]) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
    }
    amax = wave_max(amax);
    const int E = e8m0_of_amax(amax);
    const float inv = pow2_neg(E);
    if (lane == 0) scale[row] = (uint8_t)E;
    uint8_t* qr = q + row * ld_q;
    for (int c = lane * 8; c < ld_q; c += 512) {
        float v[8];
        if (c + 8 <= cols) {
            if constexpr (sizeof(Tin) == 2) { load8(xr + c, v); } else { float a[4], b[4]; load4(xr + c, a); load4(xr + c + 4, b);
                for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; } }
        } else {
            for (int j = 0; j < 8; ++j) v[j] = c + j < cols ? to_f32(xr[c + j]) : 0.f;
        }
        *reinterpret_cast<uint2*>(qr + c) = make_uint2(pack_fp8x4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv),
                                                       pack_fp8x4(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv));
    }
}

int launch_quant_rows(const void* x, int dtype, int64_t ld_x, int64_t rows, int64_t cols, void* q, int64_t ld_q, uint8_t* scale,
                      hipStream_t s) {
    if (rows == 0) return P2T_OK;
    P2T_REQUIRE(ld_q % 8 == 0 && ld_q >= cols && ld_x % 8 == 0, "quant_rows: ld_q and ld_x must be multiples of 8 (ld_q=%lld ld_x=%lld)",
                (long long)ld_q, (long long)ld_x);
    const dim3 grid((unsigned)ceil_div(rows, 4));
    if (dtype == P2T_BF16)
        quant_rows_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, ld_x, rows, (int)cols, (uint8_t*)q, ld_q, scale);
    else
        quant_rows_kernel<float><<<grid, 256, 0, s>>>((const float*)x, ld_x, rows, (int)cols, (uint8_t*)q, ld_q, scale);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

// ---- a handful of rows (one decode step of a gemm_fp8 model): one BLOCK per row, the row in registers, every load requested up
// front (the wave-per-row kernels above walk a 14 336-wide row in 28 dependent trips: 12 us; this form: one) ----
__device__ __forceinline__ float block_max4(float v, float* red) {            // 4 waves; every thread gets the maximum
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// NV chunks of 8 elements per thread: cols <= NV * 2048.  Bit-identical to quant_rows_kernel (a maximum has no order).
template <typename Tin, int NV>
__global__ void __launch_bounds__(256) quant_rows_few_kernel(const Tin* __restrict__ x, int64_t ld_x, int cols, uint8_t* __restrict__ q, int64_t ld_q,
                                                             uint8_t* __restrict__ scale) {
    __shared__ float red[4];
    const Tin* xr = x + (int64_t)blockIdx.x * ld_x;
    float v[NV][8];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
        if (c + 8 <= cols) {
            if constexpr (sizeof(Tin) == 2) { load8(xr + c, v[i]); } else { float a[4], b[4]; load4(xr + c, a); load4(xr + c + 4, b);
                for (int j = 0; j < 4; ++j) { v[i][j] = a[j]; v[i][4 + j] = b[j]; } }
        } else {
            for (int j = 0; j < 8; ++j) v[i][j] = c + j < cols ? to_f32(xr[c + j]) : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[i][j]));
    amax = block_max4(amax, red);
    const int E = e8m0_of_amax(amax);
    const float inv = pow2_neg(E);
    if (threadIdx.x == 0) scale[blockIdx.x] = (uint8_t)E;
    uint8_t* qr = q + (int64_t)blockIdx.x * ld_q;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 256 + threadIdx.x) * 8;
        if (c < ld_q)
            *reinterpret_cast<uint2*>(qr + c) = make_uint2(pack_fp8x4(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv),
                                                           pack_fp8x4(v[i][4] * inv, v[i][5] * inv, v[i][6] * inv, v[i][7] * inv));
    }
}

int launch_quant_rows_few(const void* x, int dtype, int64_t ld_x, int64_t rows, int64_t cols, void* q, int64_t ld_q, uint8_t* scale, hipStream_t s) {
    if (rows == 0) return P2T_OK;
    if (ld_q % 8 || ld_q < cols || ld_x % 8 || ld_q > 8 * 2048) return launch_quant_rows(x, dtype, ld_x, rows, cols, q, ld_q, scale, s);
#define P2T_QF(NV)                                                                                                                 \
    do {                                                                                                                           \
        if (dtype == P2T_BF16) quant_rows_few_kernel<bf16_t, NV><<<(unsigned)rows, 256, 0, s>>>((const bf16_t*)x, ld_x, (int)cols, (uint8_t*)q, ld_q, scale); \
        else quant_rows_few_kernel<float, NV><<<(unsigned)rows, 256, 0, s>>>((const float*)x, ld_x, (int)cols, (uint8_t*)q, ld_q, scale);                   \
    } while (0)
    if (ld_q <= 2048) P2T_QF(1); else if (ld_q <= 4096) P2T_QF(2); else if (ld_q <= 8192) P2T_QF(4); else P2T_QF(8);
#undef P2T_QF
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

// RMSNorm of a few rows of the f32 stream written as e4m3 + E8M0 (norm_fp8_kernel<.., RMS>'s arithmetic, another order of the sum)
template <int NV>
__global__ void __launch_bounds__(256) rmsnorm_fp8_rows_kernel(const float* __restrict__ x, int64_t ld_x, const float* __restrict__ w, float eps,
                                                               uint8_t* __restrict__ q, int64_t ld_q, uint8_t* __restrict__ scale, int cols) {
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
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[i][j] = wv[i][j] * (v[i][j] * rstd);
            amax = fmaxf(amax, fabsf(v[i][j]));
        }
    amax = block_max4(amax, red);
    const int E = e8m0_of_amax(amax);
    const float inv = pow2_neg(E);
    if (threadIdx.x == 0) scale[blockIdx.x] = (uint8_t)E;
    uint8_t* qr = q + (int64_t)blockIdx.x * ld_q;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 256 + threadIdx.x) * 4;
        if (c < cols) *reinterpret_cast<unsigned*>(qr + c) = pack_fp8x4(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv);
        else if (c < ld_q) *reinterpret_cast<unsigned*>(qr + c) = 0u;
    }
}

// LayerNorm / RMSNorm with the row kept in registers (norm.hip's structure), output quantised in place.
template <int NV, bool RMS>
__global__ void __launch_bounds__(256) norm_fp8_kernel(const float* __restrict__ x, int64_t ld_x, const float* __restrict__ w,
                                                       const float* __restrict__ b, float eps, uint8_t* __restrict__ q, int64_t ld_q,
                                                       uint8_t* __restrict__ scale, int64_t rows, int cols, float bound_w, float bound_b,
                                                       uint8_t* __restrict__ bound_scale) {
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
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < cols) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[i][j] - mean;
                    qq += d * d;
                }
            }
        }
        qq = wave_sum(qq);
        rstd = rsqrtf(qq / (float)cols + eps);
    }
    float amax = 0.f, ssq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            float wv[4];
            load4(w + c, wv);
            if (RMS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[i][j] = wv[j] * (v[i][j] * rstd);
            } else {
                float bv[4];
                load4(b + c, bv);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[i][j] = (v[i][j] - mean) * rstd * wv[j] + bv[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { amax = fmaxf(amax, fabsf(v[i][j])); ssq = fmaf(v[i][j], v[i][j], ssq); }
        }
    }
    amax = wave_max(amax);
    if (bound_scale) {           // scale of the NEXT projection's gelu output, from the Cauchy-Schwarz bound of its pre-activation
        ssq = wave_sum(ssq);
        if (lane == 0) bound_scale[row] = (uint8_t)e8m0_of_amax(fmaf(sqrtf(ssq), bound_w, bound_b));
    }
    const int E = e8m0_of_amax(amax);
    const float inv = pow2_neg(E);
    if (lane == 0) scale[row] = (uint8_t)E;
    uint8_t* qr = q + row * ld_q;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) *reinterpret_cast<unsigned*>(qr + c) = pack_fp8x4(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv);
        else if (c < ld_q) *reinterpret_cast<unsigned*>(qr + c) = 0u;
    }
}

template <bool RMS>
static int launch_norm_fp8_t(const float* x, int64_t ld_x, const float* w, const float* b, float eps, uint8_t* q, int64_t ld_q,
                             uint8_t* scale, int64_t rows, int64_t cols, float bound_w, float bound_b, uint8_t* bound_scale, hipStream_t s) {
    const dim3 grid((unsigned)ceil_div(rows, 4));
    const int64_t span = ld_q > cols ? ld_q : cols;
#define P2T_NORM8_CASE(NV)                                                                                       \
    if (span <= (NV) * 256) {                                                                                    \
        norm_fp8_kernel<NV, RMS><<<grid, 256, 0, s>>>(x, ld_x, w, b, eps, q, ld_q, scale, rows, (int)cols, bound_w, bound_b, bound_scale); \
        P2T_LAUNCH_CHECK();                                                                                      \
        return P2T_OK;                                                                                           \
    }
    P2T_NORM8_CASE(1) P2T_NORM8_CASE(2) P2T_NORM8_CASE(4) P2T_NORM8_CASE(8) P2T_NORM8_CASE(10) P2T_NORM8_CASE(16) P2T_NORM8_CASE(32)
#undef P2T_NORM8_CASE
    set_error("norm (fp8 output): %lld columns exceed the 8192 supported", (long long)cols);
    return P2T_ERR_UNSUPPORTED;
}

int launch_layernorm_fp8(const float* x, int64_t ld_x, const float* w, const float* b, float eps, void* q, int64_t ld_q,
                         uint8_t* scale, int64_t rows, int64_t cols, float bound_w, float bound_b, uint8_t* bound_scale, hipStream_t s) {
    if (rows == 0) return P2T_OK;
    P2T_REQUIRE(cols % 4 == 0 && ld_q % 4 == 0 && ld_q >= cols, "layernorm (fp8 output): cols and ld_q must be multiples of 4");
    return launch_norm_fp8_t<false>(x, ld_x, w, b, eps, (uint8_t*)q, ld_q, scale, rows, cols, bound_w, bound_b, bound_scale, s);
}
int launch_rmsnorm_fp8(const float* x, int64_t ld_x, const float* w, float eps, void* q, int64_t ld_q, uint8_t* scale, int64_t rows,
                       int64_t cols, hipStream_t s) {
    if (rows == 0) return P2T_OK;
    P2T_REQUIRE(cols % 4 == 0 && ld_q % 4 == 0 && ld_q >= cols, "rmsnorm (fp8 output): cols and ld_q must be multiples of 4");
    return launch_norm_fp8_t<true>(x, ld_x, w, nullptr, eps, (uint8_t*)q, ld_q, scale, rows, cols, 0.f, 0.f, nullptr, s);
}

// the decode step's form (a few rows): block per row; wider rows than 8192 take the wave-per-row kernel
int launch_rmsnorm_fp8_few(const float* x, int64_t ld_x, const float* w, float eps, void* q, int64_t ld_q, uint8_t* scale, int64_t rows, int64_t cols,
                           hipStream_t s) {
    if (rows == 0) return P2T_OK;
    const int64_t span = ld_q > cols ? ld_q : cols;
    if (cols % 4 || ld_q % 4 || ld_q < cols || span > 8 * 1024) return launch_rmsnorm_fp8(x, ld_x, w, eps, q, ld_q, scale, rows, cols, s);
#define P2T_RF(NV) rmsnorm_fp8_rows_kernel<NV><<<(unsigned)rows, 256, 0, s>>>(x, ld_x, w, eps, (uint8_t*)q, ld_q, scale, (int)cols)
    if (span <= 1024) P2T_RF(1); else if (span <= 2048) P2T_RF(2); else if (span <= 4096) P2T_RF(4); else P2T_RF(8);
#undef P2T_RF
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

}  // namespace p2t

using namespace p2t;

extern "C" int p2t_quant_rows_fp8(const void* x, int dtype, int64_t ld_x, int64_t rows, int64_t cols, void* q, int64_t ld_q,
                                  uint8_t* scale, p2t_stream stream) {
    P2T_REQUIRE(x && q && scale && rows >= 0 && cols > 0 && (dtype == P2T_F32 || dtype == P2T_BF16), "p2t_quant_rows_fp8: bad arguments");
    return launch_quant_rows(x, dtype, ld_x, rows, cols, q, ld_q, scale, (hipStream_t)stream);
}

extern "C" int p2t_layernorm_fp8(const float* x, int64_t ld_x, const float* w, const float* b, float eps, void* q, int64_t ld_q,
                                 uint8_t* scale, int64_t rows, int64_t cols, float bound_w, float bound_b, uint8_t* bound_scale,
                                 p2t_stream stream) {
    P2T_REQUIRE(x && w && b && q && scale && rows >= 0 && cols > 0 && bound_w >= 0.f && bound_b >= 0.f, "p2t_layernorm_fp8: bad arguments");
    return launch_layernorm_fp8(x, ld_x, w, b, eps, q, ld_q, scale, rows, cols, bound_w, bound_b, bound_scale, (hipStream_t)stream);
}

extern "C" int p2t_rmsnorm_fp8(const float* x, int64_t ld_x, const float* w, float eps, void* q, int64_t ld_q, uint8_t* scale,
                               int64_t rows, int64_t cols, p2t_stream stream) {
    P2T_REQUIRE(x && w && q && scale && rows >= 0 && cols > 0, "p2t_rmsnorm_fp8: bad arguments");
    return launch_rmsnorm_fp8(x, ld_x, w, eps, q, ld_q, scale, rows, cols, (hipStream_t)stream);
}
