// Shared device/host helpers for libp2t_hip (gfx950 / CDNA4 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/p2t_hip.h"

typedef __bf16 bf16_t;
typedef short short8 __attribute__((ext_vector_type(8)));
typedef short short4_ __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define P2T_WAVE 64

namespace p2t {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define P2T_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            p2t::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return P2T_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)
#define P2T_REQUIRE(cond, ...)                                                                \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            p2t::set_error(__VA_ARGS__);                                                      \
            return P2T_ERR_ARG;                                                               \
        }                                                                                     \
    } while (0)
#define P2T_LAUNCH_CHECK() P2T_CHECK_HIP(hipGetLastError())
#define P2T_TRY(expr)                                                                         \
    do {                                                                                      \
        int _r = (expr);                                                                      \
        if (_r != P2T_OK) return _r;                                                          \
    } while (0)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }
static inline size_t dtype_size(int dt) { return dt == P2T_BF16 ? 2 : 4; }

// ---- bf16 <-> f32 ---------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned)b) << 16);
}
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }   // RNE, NaN-safe

// 4 consecutive elements, vectorised
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {       // one v_cvt_pk_bf16_f32 (round to nearest even)
    typedef float p2t_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 p2t_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(p2t_f32x2{lo, hi}, p2t_bf16x2));
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}
// 8 consecutive bf16 -> f32
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xFFFF0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xFFFF0000u);
}

// ---- wave / block reductions ------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// sum over a block of NW waves; `red` is NW floats of LDS; every thread gets the result
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float gelu_erf(float x) {            // x * 0.5 * (1 + erf(x / sqrt(2)))
    return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, a third of erff's instruction count): used where
// the result is rounded to bf16 (2^-9 relative) anyway; fp32 outputs keep erff.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
template <typename Tout> __device__ __forceinline__ float gelu_erf_for(float x) {
    if constexpr (sizeof(Tout) == 2) return x * 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f));
    else return gelu_erf(x);
}
// the same arithmetic on a pair of elements: the four Horner steps, the products and the final scaling become packed-fp32
// instructions (v_pk_fma_f32 / v_pk_mul_f32: two elements per issue); rcp / exp stay one per element.  `scale` multiplies the result
// (the e4m3 epilogue's 2^-(E - 127)).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf_as_x2(f32x2 x, float scale) {
    const f32x2 z = x * 0.70710678118654752440f;
    const f32x2 az = __builtin_elementwise_abs(z);
    f32x2 t;
    t.x = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az.x, 1.0f));
    t.y = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az.y, 1.0f));
    f32x2 p = t * 1.061405429f + (-1.453152027f);
    p = p * t + 1.421413741f;
    p = p * t + (-0.284496736f);
    p = p * t + 0.254829592f;
    const f32x2 w = z * 1.2011224087864498f;              // sqrt(log2 e): exp(-z^2) = exp2(-(w^2)), one packed product less
    const f32x2 a = -(w * w);
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(a.x);
    e.y = __builtin_amdgcn_exp2f(a.y);
    f32x2 r = 1.0f - p * t * e;
    r.x = copysignf(r.x, z.x);
    r.y = copysignf(r.y, z.y);
    return (x * (0.5f * scale)) * (1.0f + r);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;
    return cdf + x * pdf;
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
// where the result is rounded to bf16: hardware exp2 / rcp (1 ulp each) instead of expf's range handling and the IEEE division
// (5 instructions instead of ~20 per element)
template <typename Tout> __device__ __forceinline__ float silu_for(float x) {
    if constexpr (sizeof(Tout) == 2) return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
    else return silu(x);
}

// counter hash shared with p2t_hip/synth.py (splitmix64 finaliser)
__device__ __host__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

}  // namespace p2t
