// GEMM epilogues shared by the fp32-FMA kernel (gemm_simple.hip) and the bf16 MFMA kernel
// (gemm_mfma.hip).  A kernel hands over, for one output row m, two groups of 4 consecutive
// columns: v0 at n0..n0+3 and v1 at n0+16..n0+19 with (n0 % 32) < 16 -- the shape both the
// 16x16 MFMA accumulator fragment (swapped operands: a lane owns 4 consecutive n of one m) and
// the FMA micro-tile produce.  The +16 partner is what lets SwiGLU (gate/up interleaved in
// 16-row blocks of the packed weight) finish in registers.
#pragma once
#include "common.h"

namespace p2t {

struct EpiParams {
    const float* bias;      // [N] or nullptr
    void* out;              // [M, ldc]
    void* z;                // optional pre-activation (GELU), same layout as out
    int64_t ldc;
    int64_t M;
    int N;                  // logical columns of the GEMM (rows of W)
    int n_zero;             // output columns [N_out, n_zero) are written as zeros (K padding of the consumer)
    int accumulate;
    float drop_p;           // GELU only: dropout on the activation output
    float drop_scale;       // 1 / (1 - p)
    uint64_t drop_seed;
};

__device__ __forceinline__ bool dropout_keep(uint64_t seed, int64_t idx, float p) {
    const uint32_t u = (uint32_t)(mix64((uint64_t)idx + seed) >> 40);
    return (float)u >= p * 16777216.0f;
}

template <typename Tout>
struct EpiStore {
    static constexpr bool kPair = false;
    __device__ __forceinline__ static void group(const EpiParams& p, int64_t m, int n, const float (&v)[4]) {
        Tout* o = (Tout*)p.out + m * p.ldc + n;
        if (n < p.N) {
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = v[j] + (p.bias ? p.bias[n + j] : 0.f);
            store4(o, r);
        } else if (n < p.n_zero) {
            const float zz[4] = {0.f, 0.f, 0.f, 0.f};
            store4(o, zz);
        }
    }
    __device__ __forceinline__ static void apply2(const EpiParams& p, int64_t m, int n0, const float (&v0)[4],
                                                  const float (&v1)[4]) {
        group(p, m, n0, v0);
        group(p, m, n0 + 16, v1);
    }
};

template <typename Tout>
struct EpiGelu {
    static constexpr bool kPair = false;
    __device__ __forceinline__ static void group(const EpiParams& p, int64_t m, int n, const float (&v)[4]) {
        Tout* o = (Tout*)p.out + m * p.ldc + n;
        if (n < p.N) {
            float zv[4], r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                zv[j] = v[j] + (p.bias ? p.bias[n + j] : 0.f);
                r[j] = gelu_erf(zv[j]);
                if (p.drop_p > 0.f) r[j] = dropout_keep(p.drop_seed, m * (int64_t)p.N + n + j, p.drop_p) ? r[j] * p.drop_scale : 0.f;
            }
            if (p.z) store4((Tout*)p.z + m * p.ldc + n, zv);
            store4(o, r);
        } else if (n < p.n_zero) {
            const float zz[4] = {0.f, 0.f, 0.f, 0.f};
            store4(o, zz);
            if (p.z) store4((Tout*)p.z + m * p.ldc + n, zz);
        }
    }
    __device__ __forceinline__ static void apply2(const EpiParams& p, int64_t m, int n0, const float (&v0)[4],
                                                  const float (&v1)[4]) {
        group(p, m, n0, v0);
        group(p, m, n0 + 16, v1);
    }
};

// backward through dropout(gelu(z)): out = acc * gelu'(z) * dropout_mask; z is READ (layout of out)
template <typename Tout>
struct EpiGeluBwd {
    static constexpr bool kPair = false;
    __device__ __forceinline__ static void group(const EpiParams& p, int64_t m, int n, const float (&v)[4]) {
        Tout* o = (Tout*)p.out + m * p.ldc + n;
        if (n < p.N) {
            float zv[4], r[4];
            load4((const Tout*)p.z + m * p.ldc + n, zv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[j] = v[j] * gelu_erf_grad(zv[j]);
                if (p.drop_p > 0.f) r[j] = dropout_keep(p.drop_seed, m * (int64_t)p.N + n + j, p.drop_p) ? r[j] * p.drop_scale : 0.f;
            }
            store4(o, r);
        } else if (n < p.n_zero) {
            const float zz[4] = {0.f, 0.f, 0.f, 0.f};
            store4(o, zz);
        }
    }
    __device__ __forceinline__ static void apply2(const EpiParams& p, int64_t m, int n0, const float (&v0)[4],
                                                  const float (&v1)[4]) {
        group(p, m, n0, v0);
        group(p, m, n0 + 16, v1);
    }
};

// residual stream (f32) += acc + bias, in place
struct EpiResid {
    static constexpr bool kPair = false;
    __device__ __forceinline__ static void group(const EpiParams& p, int64_t m, int n, const float (&v)[4]) {
        if (n >= p.N) return;
        float* o = (float*)p.out + m * p.ldc + n;
        float r[4];
        load4(o, r);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] += v[j] + (p.bias ? p.bias[n + j] : 0.f);
        store4(o, r);
    }
    __device__ __forceinline__ static void apply2(const EpiParams& p, int64_t m, int n0, const float (&v0)[4],
                                                  const float (&v1)[4]) {
        group(p, m, n0, v0);
        group(p, m, n0 + 16, v1);
    }
};

// f32 store / accumulate (gradients)
struct EpiF32 {
    static constexpr bool kPair = false;
    __device__ __forceinline__ static void group(const EpiParams& p, int64_t m, int n, const float (&v)[4]) {
        if (n >= p.N) return;
        float* o = (float*)p.out + m * p.ldc + n;
        float r[4] = {v[0], v[1], v[2], v[3]};
        if (p.accumulate) {
            float c[4];
            load4(o, c);
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] += c[j];
        }
        store4(o, r);
    }
    __device__ __forceinline__ static void apply2(const EpiParams& p, int64_t m, int n0, const float (&v0)[4],
                                                  const float (&v1)[4]) {
        group(p, m, n0, v0);
        group(p, m, n0 + 16, v1);
    }
};

// out[m, f] = silu(gate_f) * up_f; packed W rows 32j..32j+15 = gate features 16j.., rows 32j+16.. = up.
template <typename Tout>
struct EpiSwiglu {
    static constexpr bool kPair = true;
    __device__ __forceinline__ static void apply2(const EpiParams& p, int64_t m, int n0, const float (&g)[4],
                                                  const float (&u)[4]) {
        const int f = (n0 >> 5) * 16 + (n0 & 15);
        Tout* o = (Tout*)p.out + m * p.ldc + f;
        if (n0 < p.N) {
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = silu(g[j]) * u[j];
            store4(o, r);
        } else if (f < p.n_zero) {
            const float zz[4] = {0.f, 0.f, 0.f, 0.f};
            store4(o, zz);
        }
    }
};

}  // namespace p2t
