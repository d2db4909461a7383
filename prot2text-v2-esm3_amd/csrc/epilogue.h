// GEMM epilogues shared by the fp32-FMA kernel (gemm_simple.hip) and the bf16 MFMA kernel
// (gemm_mfma.hip).  A kernel hands over, for one output row m, two groups of W consecutive columns:
// v0 at n .. n+W-1 and v1 at n+32 .. n+32+W-1 with (n % 64) < 32, plus the matching bias values.
// The +32 partner is what lets the pairwise epilogues finish in registers:
//   * SwiGLU      -- gate/up rows interleaved in 32-row blocks of the packed weight (rows 64g..64g+31 = gate
//                    features 32g.., rows 64g+32..64g+63 = up features 32g..);
//   * QKV + RoPE  -- head_dim 64: column j and its rotate-half partner j+32 of the same head.
// W = 8 for the MFMA kernel (its W-row permutation gives a lane 8 consecutive columns -> 16-byte bf16 stores),
// W = 4 for the FMA kernel.
#pragma once
#include "common.h"

namespace p2t {

struct EpiParams {
    const float* bias;      // [N] or nullptr
    void* out;              // [M, ldc]
    void* z;                // optional pre-activation (GELU: written; GELU_BWD: read), same layout as out
    int64_t ldc;
    int64_t M;
    int N;                  // logical columns of the GEMM (rows of W)
    int n_zero;             // output columns [N_out, n_zero) are written as zeros (K padding of the consumer)
    int accumulate;
    float drop_p;           // GELU / GELU_BWD: dropout on the activation output
    float drop_scale;       // 1 / (1 - p)
    uint64_t drop_seed;
    // QKV + RoPE epilogue (head_dim 64): outputs [B, heads, T, 64]
    const float* cs;        // [T, 64]: cos (32) | sin (32)
    void* q; void* k; void* v;
    int seq, nh, nkv;
    int head_dim;           // 64, or 128 with the packed row order of include/p2t_hip.h (p2t_llama_layer)
    float q_scale;
    const uint8_t* row_scale = nullptr;   // GELU_FP8: E8M0 scale byte of every OUTPUT row (known before the GEMM runs)
};

__device__ __forceinline__ bool dropout_keep(uint64_t seed, int64_t idx, float p) {
    const uint32_t u = (uint32_t)(mix64((uint64_t)idx + seed) >> 40);
    return (float)u >= p * 16777216.0f;
}

// GEMM outputs are hundreds of MB written once and read by a LATER kernel: non-temporal stores keep a round's
// 32 MB of results from evicting the operand panels out of the 4 MB L2s (measured +1 % on the GEMM average).
typedef float p2t_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned p2t_u32x4 __attribute__((ext_vector_type(4)));
#ifdef P2T_STORE_PLAIN            // A/B switch: epilogue stores through the L2s (write-back) instead of streaming past them
#define P2T_EPI_STORE(val, ptr) (*(ptr) = (val))
#else
#define P2T_EPI_STORE(val, ptr) __builtin_nontemporal_store(val, ptr)
#endif
template <int W> __device__ __forceinline__ void storeW(float* p, const float (&v)[W]) {
#pragma unroll
    for (int c = 0; c < W; c += 4) {
        P2T_EPI_STORE((p2t_f32x4{v[c], v[c + 1], v[c + 2], v[c + 3]}), reinterpret_cast<p2t_f32x4*>(p + c));
    }
}
template <int W> __device__ __forceinline__ void storeW(bf16_t* p, const float (&v)[W]) {
    if constexpr (W == 8) {
        P2T_EPI_STORE((p2t_u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])}),
                      reinterpret_cast<p2t_u32x4*>(p));
    } else {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
}
template <int W> __device__ __forceinline__ void loadW(const float* p, float (&v)[W]) {
#pragma unroll
    for (int c = 0; c < W; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + c);
        v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w;
    }
}
template <int W> __device__ __forceinline__ void loadW(const bf16_t* p, float (&v)[W]) {
    if constexpr (W == 8) {
        load8(p, v);
    } else {
        load4(p, v);
    }
}
template <int W> __device__ __forceinline__ void zeroW(float (&v)[W]) {
#pragma unroll
    for (int c = 0; c < W; ++c) v[c] = 0.f;
}

// Every functor: group(p, m, n, v, b) handles W consecutive columns starting at n; apply2 = both groups.  IN: the caller
// guarantees the tile lies inside the output (n + 32 + W <= N): no per-lane column checks, the epilogue of a tile is one
// basic block the scheduler can interleave across rows.
#define P2T_EPI_APPLY2                                                                                          \
    template <int W, bool IN = false>                                                                           \
    __device__ __forceinline__ static void apply2(const EpiParams& p, int64_t m, int n, const float (&v0)[W],   \
                                                  const float (&v1)[W], const float (&b0)[W], const float (&b1)[W]) { \
        group<W, IN>(p, m, n, v0, b0);                                                                          \
        group<W, IN>(p, m, n + 32, v1, b1);                                                                     \
    }

// kMinOps (every functor): a LOWER bound of the vector-memory instructions one apply2<8> call issues on a tile that
// lies fully inside the output -- the persistent GEMM keeps that many more operations in flight across a tile
// boundary (counted s_waitcnt vmcnt), so the next tile's main loop does not wait for the epilogue stores to drain.
template <typename Tout> constexpr int kStoreOps8 = (int)(8 * sizeof(Tout) / 16);     // 16-byte stores per 8 values

template <typename Tout>
struct EpiStore {
    static constexpr bool kRmw = false;
    static constexpr int kMinOps = 2 * kStoreOps8<Tout>;
    template <int W, bool IN = false>
    __device__ __forceinline__ static void group(const EpiParams& p, int64_t m, int n, const float (&v)[W], const float (&b)[W]) {
        Tout* o = (Tout*)p.out + m * p.ldc + n;
        float r[W];
        if (IN || n < p.N) {
#pragma unroll
            for (int j = 0; j < W; ++j) r[j] = v[j] + b[j];
            storeW<W>(o, r);
        } else if (n < p.n_zero) {
            zeroW<W>(r);
            storeW<W>(o, r);
        }
    }
    P2T_EPI_APPLY2
};

// EXTRA: also dropout (p.drop_p > 0) and / or the pre-activation copy (p.z) -- the trained adapter's forward.  The towers'
// FFN uses the plain form: no per-group branches, so the 16 erf evaluations of a row pair overlap.
template <typename Tout, bool EXTRA = false>
struct EpiGelu {
    static constexpr bool kRmw = false;
    static constexpr int kMinOps = 2 * kStoreOps8<Tout>;
    template <int W, bool IN = false>
    __device__ __forceinline__ static void group(const EpiParams& p, int64_t m, int n, const float (&v)[W], const float (&b)[W]) {
        Tout* o = (Tout*)p.out + m * p.ldc + n;
        float zv[W], r[W];
        if (IN || n < p.N) {
#pragma unroll
            for (int j = 0; j < W; ++j) {
                zv[j] = v[j] + b[j];
                r[j] = gelu_erf_for<Tout>(zv[j]);
                if constexpr (EXTRA) {
                    if (p.drop_p > 0.f) r[j] = dropout_keep(p.drop_seed, m * (int64_t)p.N + n + j, p.drop_p) ? r[j] * p.drop_scale : 0.f;
                }
            }
            if constexpr (EXTRA) {
                if (p.z) storeW<W>((Tout*)p.z + m * p.ldc + n, zv);
            }
            storeW<W>(o, r);
        } else if (n < p.n_zero) {
            zeroW<W>(r);
            storeW<W>(o, r);
            if constexpr (EXTRA) {
                if (p.z) storeW<W>((Tout*)p.z + m * p.ldc + n, r);
            }
        }
    }
    P2T_EPI_APPLY2
};

// fp8 towers: C = e4m3(gelu_erf(acc + bias) * 2^-(E_m - 127)) with the row's E8M0 scale byte E_m given (p.row_scale): the
// output feeds the next fp8 GEMM directly, no bf16 copy and no quantise pass.  E_m must bound the row (quant.hip derives it
// from the Cauchy-Schwarz bound of the pre-activation); out is bytes, ldc in bytes.
__device__ __forceinline__ unsigned epi_pack_fp8x4(float a, float b, float c, float d) {
    int r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (unsigned)r;
}
struct EpiGeluFp8 {
    static constexpr bool kRmw = false;
    static constexpr int kMinOps = 2;
    template <int W, bool IN = false>
    __device__ __forceinline__ static void group(const EpiParams& p, int64_t m, int n, const float (&v)[W], const float (&b)[W]) {
        static_assert(W == 8, "fp8 output: MFMA kernels only (8 consecutive columns per lane)");
        uint8_t* o = (uint8_t*)p.out + m * p.ldc + n;
        if (IN || n < p.N) {
            const float inv = __uint_as_float((unsigned)(254 - (int)p.row_scale[m]) << 23);      // 2^-(E - 127), exact
            float r[W];      // e4m3 keeps 3 mantissa bits: the 1.5e-7 erf of the bf16 epilogue is ample (erff: 2.5 x the instructions)
#pragma unroll
            for (int j = 0; j < W; j += 2) {
                const f32x2 g = gelu_erf_as_x2(f32x2{v[j] + b[j], v[j + 1] + b[j + 1]}, inv);
                r[j] = g.x; r[j + 1] = g.y;
            }
            *reinterpret_cast<uint2*>(o) = make_uint2(epi_pack_fp8x4(r[0], r[1], r[2], r[3]), epi_pack_fp8x4(r[4], r[5], r[6], r[7]));
        } else if (n < p.n_zero) {
            *reinterpret_cast<uint2*>(o) = make_uint2(0u, 0u);
        }
    }
    P2T_EPI_APPLY2
};

// backward through dropout(gelu(z)): out = acc * gelu'(z) * dropout_mask; z is READ (layout of out)
template <typename Tout>
struct EpiGeluBwd {
    static constexpr bool kRmw = false;
    static constexpr int kMinOps = 4 * kStoreOps8<Tout>;
    template <int W, bool IN = false>
    __device__ __forceinline__ static void group(const EpiParams& p, int64_t m, int n, const float (&v)[W], const float (&b)[W]) {
        Tout* o = (Tout*)p.out + m * p.ldc + n;
        float zv[W], r[W];
        if (IN || n < p.N) {
            loadW<W>((const Tout*)p.z + m * p.ldc + n, zv);
#pragma unroll
            for (int j = 0; j < W; ++j) {
                r[j] = v[j] * gelu_erf_grad(zv[j]);
                if (p.drop_p > 0.f) r[j] = dropout_keep(p.drop_seed, m * (int64_t)p.N + n + j, p.drop_p) ? r[j] * p.drop_scale : 0.f;
            }
            storeW<W>(o, r);
        } else if (n < p.n_zero) {
            zeroW<W>(r);
            storeW<W>(o, r);
        }
    }
    P2T_EPI_APPLY2
};

// residual stream (f32) += acc + bias, in place
struct EpiResid {
    static constexpr bool kRmw = true;
    static constexpr int kMinOps = 8;
    // interior tiles only: the stream values of both column groups, fetched ahead of the adds
    template <int W, bool UNI = false>
    __device__ __forceinline__ static void fetch2(const EpiParams& p, int64_t m, int n, float (&r0)[W], float (&r1)[W]) {
        const float* o = (const float*)p.out + m * p.ldc + n;
        loadW<W>(o, r0);
        loadW<W>(o + 32, r1);
    }
    template <int W, bool UNI = false>
    __device__ __forceinline__ static void apply2_fetched(const EpiParams& p, int64_t m, int n, const float (&v0)[W], const float (&v1)[W],
                                                          const float (&b0)[W], const float (&b1)[W], float (&r0)[W], float (&r1)[W]) {
        float* o = (float*)p.out + m * p.ldc + n;
#pragma unroll
        for (int j = 0; j < W; ++j) { r0[j] += v0[j] + b0[j]; r1[j] += v1[j] + b1[j]; }
        storeW<W>(o, r0);
        storeW<W>(o + 32, r1);
    }
    template <int W, bool IN = false>
    __device__ __forceinline__ static void group(const EpiParams& p, int64_t m, int n, const float (&v)[W], const float (&b)[W]) {
        if (!IN && n >= p.N) return;
        float* o = (float*)p.out + m * p.ldc + n;
        float r[W];
        loadW<W>(o, r);
#pragma unroll
        for (int j = 0; j < W; ++j) r[j] += v[j] + b[j];
        storeW<W>(o, r);
    }
    P2T_EPI_APPLY2
};

// f32 store / accumulate (gradients)
struct EpiF32 {
    static constexpr bool kRmw = false;
    static constexpr int kMinOps = 4;
    template <int W, bool IN = false>
    __device__ __forceinline__ static void group(const EpiParams& p, int64_t m, int n, const float (&v)[W], const float (&b)[W]) {
        if (!IN && n >= p.N) return;
        float* o = (float*)p.out + m * p.ldc + n;
        float r[W];
        if (p.accumulate) {
            loadW<W>(o, r);
#pragma unroll
            for (int j = 0; j < W; ++j) r[j] += v[j];
        } else {
#pragma unroll
            for (int j = 0; j < W; ++j) r[j] = v[j];
        }
        storeW<W>(o, r);
    }
    P2T_EPI_APPLY2
};

// out[m, f] = silu(gate_f) * up_f with f = (n / 64) * 32 + (n % 32): gate in the first group, up in the partner.
template <typename Tout>
struct EpiSwiglu {
    static constexpr bool kRmw = false;
    static constexpr int kMinOps = kStoreOps8<Tout>;
    template <int W, bool IN = false>
    __device__ __forceinline__ static void apply2(const EpiParams& p, int64_t m, int n, const float (&g)[W], const float (&u)[W],
                                                  const float (&b0)[W], const float (&b1)[W]) {
        const int f = (n >> 6) * 32 + (n & 31);
        Tout* o = (Tout*)p.out + m * p.ldc + f;
        float r[W];
        if (IN || n < p.N) {
#pragma unroll
            for (int j = 0; j < W; ++j) r[j] = silu_for<Tout>(g[j]) * u[j];
            storeW<W>(o, r);
        } else if (f < p.n_zero) {
            zeroW<W>(r);
            storeW<W>(o, r);
        }
    }
};

// QKV projection + bias + query scale + rotary + head split: q [B, nh, T, d], k [B, nkv, T, d], v [B, nkv, T, d], d = 64 or 128.
// The kernels hand a lane the columns n .. n+W-1 and n+32 .. n+32+W-1 of a 64-column block.  d = 64: the block is a head
// and the partner IS the rotate-half partner (j, j+32).  d = 128: the weight rows of every head are PACKED in the order
// 0..31, 64..95, 32..63, 96..127, so block 2h holds the channels (j, j+64) for j < 32 and block 2h+1 those for 32 <= j < 64
// -- again (first group, partner group).  Restates HF EsmSelfAttention.forward up to the attention call
// (modeling_esm.py:362-378) and LlamaAttention.forward (modeling_llama.py:254-259).
template <typename Tout>
struct EpiQkvRope {
    static constexpr bool kRmw = false;
    static constexpr int kMinOps = 2 * kStoreOps8<Tout>;
    // The rotation in two parts, so that a caller can request the cos / sin rows of the NEXT row group before it stores this one
    // (gemm_tile_common.h, tile_epilogue_pair): fetch2 = the two table loads (nothing for a V head), apply2_fetched = the rest.
    static constexpr bool kFetch = true;
    // UNI: the caller walks whole wave tiles (tile_epilogue_pair) -- every lane of the wave is in the same 64-column block, so the head index
    // is a scalar; and the two table loads are issued for V heads too (a valid row, never used): a branch around them lets the compiler sink
    // the loads to their use, which is exactly the latency the caller issues them early to hide.
    // fetch_shared: with head_dim 64 a lane's rotary channels are the same in every 64-column block, so one fetch serves both halves of a row.
    __device__ __forceinline__ static bool fetch_shared(const EpiParams& p) { return p.head_dim == 64; }
    template <int W, bool UNI = false>
    __device__ __forceinline__ static void fetch2(const EpiParams& p, int64_t m, int n, float (&c)[W], float (&s)[W]) {
        const int hd = p.head_dim, half = hd >> 1;
        const int blk = UNI ? __builtin_amdgcn_readfirstlane(n >> 6) : n >> 6;
        const int head = hd == 64 ? blk : blk >> 1;
        const int j = (hd == 64 ? 0 : 32 * (blk & 1)) + (n & 31);
        const uint32_t mu = (uint32_t)m, sq = (uint32_t)p.seq;
        const uint32_t b = mu / sq, t = mu - b * sq;
        if (UNI || head < p.nh + p.nkv) {
            loadW<W>(p.cs + (size_t)(t * (uint32_t)hd) + j, c);
            loadW<W>(p.cs + (size_t)(t * (uint32_t)hd) + half + j, s);
        }
    }
    template <int W, bool UNI = false>
    __device__ __forceinline__ static void apply2_fetched(const EpiParams& p, int64_t m, int n, const float (&v0)[W], const float (&v1)[W],
                                                          const float (&b0)[W], const float (&b1)[W], const float (&c)[W], const float (&s)[W]) {
        const int hd = p.head_dim, half = hd >> 1;
        const int blk = UNI ? __builtin_amdgcn_readfirstlane(n >> 6) : n >> 6;
        const int head = hd == 64 ? blk : blk >> 1;
        const int j = (hd == 64 ? 0 : 32 * (blk & 1)) + (n & 31);          // rotary channel: pairs (j, j + hd/2)
        // 32-bit unsigned arithmetic (gemm_nt checks M < 2^31 and B * heads * seq < 2^31): a 64-bit division per row was a
        // third of this epilogue's instructions
        const uint32_t mu = (uint32_t)m, sq = (uint32_t)p.seq;
        const uint32_t b = mu / sq, t = mu - b * sq;
        float x1[W], x2[W];
#pragma unroll
        for (int e = 0; e < W; ++e) { x1[e] = v0[e] + b0[e]; x2[e] = v1[e] + b1[e]; }
        Tout* dst;
        if (head < p.nh + p.nkv) {
            const bool is_q = head < p.nh;
            const float sc = is_q ? p.q_scale : 1.0f;
            float o1[W], o2[W];
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const float a1 = x1[e] * sc, a2 = x2[e] * sc;
                o1[e] = a1 * c[e] - a2 * s[e];
                o2[e] = a2 * c[e] + a1 * s[e];
            }
            dst = is_q ? (Tout*)p.q + (size_t)((b * (uint32_t)p.nh + (uint32_t)head) * sq + t) * (size_t)hd
                       : (Tout*)p.k + (size_t)((b * (uint32_t)p.nkv + (uint32_t)(head - p.nh)) * sq + t) * (size_t)hd;
            storeW<W>(dst + j, o1);
            storeW<W>(dst + half + j, o2);
        } else {
            dst = (Tout*)p.v + (size_t)((b * (uint32_t)p.nkv + (uint32_t)(head - p.nh - p.nkv)) * sq + t) * (size_t)hd;
            storeW<W>(dst + j, x1);
            storeW<W>(dst + half + j, x2);
        }
    }
    template <int W, bool IN = false>
    __device__ __forceinline__ static void apply2(const EpiParams& p, int64_t m, int n, const float (&v0)[W], const float (&v1)[W],
                                                  const float (&b0)[W], const float (&b1)[W]) {
        if (!IN && n >= p.N) return;
        float c[W], s[W];
#pragma unroll
        for (int e = 0; e < W; ++e) c[e] = s[e] = 0.f;
        fetch2<W>(p, m, n, c, s);
        apply2_fetched<W>(p, m, n, v0, v1, b0, b1, c, s);
    }
};

}  // namespace p2t
