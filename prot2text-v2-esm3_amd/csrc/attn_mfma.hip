// bf16 flash attention forward for gfx950 (ESM2 bidirectional + key padding, Llama causal GQA).
//
// One workgroup = 4 wavefronts = 128 queries of one (batch, head); each wave owns 32 queries.
// Per 64-key step:   S^T = K . Q^T   (v_mfma_f32_32x32x16_bf16, keys on the accumulator rows, the
// query on the lane, so the softmax row statistics are lane-local + one lane^32 exchange), then
// O^T += V^T . P^T with the S^T accumulators converted in place to the B operand (no LDS round trip:
// "accumulator tile as the next MFMA's operand", cdna_hip_programming.md section 3).  The K rows are
// loaded in the order that makes the matching V^T operand 8 CONSECUTIVE keys: row i of the K
// fragment holds key pi(i), pi = swap bits 2 and 3 -- a pure address permutation.
// K and V are both row-major [keys][dp] in HBM and in LDS; the V^T operand is produced by the
// hardware transpose read ds_read_b64_tr_b16 (technique T10), so no transposed copy of V exists.
// K / V tiles are staged with global_load_lds (double buffered), the bank swizzle applied on the
// source address and on the reads (rule 21): 16-B chunk c of row r is stored at c ^ g(r) with
//   dp = 32: g = (r >> 2) & 3      dp = 64: g = ((r >> 1) & 1) << 2 | (r >> 2) & 3      dp = 128: g = (r & 3) << 2 | (r >> 2) & 3
// which is conflict-free both for the 32-row ds_read_b128 fragment pattern (K) and for the 4-row x 64-byte
// transpose-read pattern (V).
// Softmax in fp32 with exp2 (log2(e) folded into the scale); P is rounded to bf16 for the PV MFMA.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace p2t {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;
using lds_s4_t = __attribute__((address_space(3))) short4v*;

__device__ __forceinline__ int perm23(int i) {       // swap bits 2 and 3
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}

template <int DP>
__device__ __forceinline__ int swz_g(int row) {
    if (DP == 32) return (row >> 2) & 3;
    if (DP == 64) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    return ((row & 3) << 2) | ((row >> 2) & 3);
}

template <int DP>
__global__ void __launch_bounds__(256) attn_mfma_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, const uint8_t* __restrict__ key_mask,
                                                        const int32_t* __restrict__ kv_info, bf16_t* __restrict__ out,
                                                        int64_t ld_out, int B, int seq, int nh, int nkv, int d,
                                                        float scale_log2e, int causal, int out_cols) {
    constexpr int RB = DP * 2;                 // tile row bytes
    constexpr int T_BYTES = 64 * RB;           // one tile: 64 keys
    constexpr int STAGE = 2 * T_BYTES;         // K tile + V tile
    constexpr int NI = DP / 32;                // glds instructions per wave per tile
    constexpr int DK = DP / 16;                // QK^T k-steps
    constexpr int DT = DP / 32;                // O^T row tiles
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = blockIdx.y, b = blockIdx.z;
    const int hk = h / (nh / nkv);
    const int q0 = blockIdx.x * 128 + w * 32;
    const int lq = lane & 31, hh = lane >> 5;
    const int query = q0 + lq;

    int end = kv_info[b];
    const int prefix = kv_info[B + b];
    if (causal) end = min(end, (int)blockIdx.x * 128 + 128);
    const int n_it = (end + 63) >> 6;

    const bf16_t* kbase = k + ((int64_t)(b * nkv + hk) * seq) * DP;
    const bf16_t* vbase = v + ((int64_t)(b * nkv + hk) * seq) * DP;

    // ---- Q fragments (B operand of K.Q^T): lane holds Q[query][kk*16 + 8*hh .. +8] ----
    bf16x8 qf[DK];
    {
        const int qr = query < seq ? query : seq - 1;
        const bf16_t* qrow = q + ((int64_t)(b * nh + h) * seq + qr) * DP + 8 * hh;
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qrow + kk * 16);
    }

    // ---- staging (global_load_lds, 1 KiB per wave instruction); K and V tiles share the layout ----
    auto stage = [&](int buf, int it) {
        const int kb = it * 64;
        char* sb = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int ii = w + 4 * i;
            const int byte = ii * 1024 + lane * 16;
            const int row = byte / RB, p = (byte % RB) >> 4;
            const int c = p ^ swz_g<DP>(row);
            int key = kb + row;
            key = key < seq ? key : seq - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(kbase + (int64_t)key * DP + c * 8), (lptr_t)(sb + ii * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(vbase + (int64_t)key * DP + c * 8), (lptr_t)(sb + T_BYTES + ii * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets ----
    int k_off[2][DK];                          // [tile][k-step]: ds_read_b128 of K row pi(lq), chunk 2 kk + hh
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = t * 32 + perm23(lq);
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) k_off[t][kk] = row * RB + (((kk * 2 + hh) ^ swz_g<DP>(row)) << 4);
    }
    // V^T operand of d-tile dt, key step ss (16 keys), half r (keys 8 hh + 4 r .. +3): the 16-lane group
    // (lane >> 4) reads the 4 x 16 block rows ss*16 + 8*hh + 4*r + qq, columns dt*32 + 16*((lane >> 4) & 1) + 4 pp .. +3
    // where lane & 15 = 4 qq + pp supplies row qq / column piece pp (ds_read_b64_tr_b16 semantics).
    int v_off[DT][4][2];
    {
        const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int ss = 0; ss < 4; ++ss)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int row = ss * 16 + 8 * hh + 4 * r + qq;
                    const int col = dt * 32 + 16 * g1 + 4 * pp;           // element column; chunk = col / 8
                    v_off[dt][ss][r] = T_BYTES + row * RB + ((((col >> 3)) ^ swz_g<DP>(row)) << 4) + (col & 7) * 2;
                }
    }

    f32x16 ot[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // One 64-key step on LDS buffer BUF (a compile-time constant: every ds_read address is then a per-lane VGPR
    // plus an immediate, no per-read address arithmetic on the VALU, which is the busier pipe in this kernel).
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto step = [&](auto bufc, int it) {
        constexpr int BUF = decltype(bufc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (it + 1 < n_it) stage(BUF ^ 1, it + 1);
        const char* sb = smem + BUF * STAGE;
        const int kb = it * 64;

        // S^T tiles: rows = keys (permuted), cols = queries
        f32x16 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sb + k_off[t][kk]);
                st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], kk == 0 ? zero16 : st[t], 0, 0, 0);
            }
        }
        // Masking only where a key of this 64-key step can be hidden from a query of this wave (tail of the
        // sequence, causal diagonal, or a non-prefix mask): a wave-uniform branch keeps the VALU work of the
        // interior steps at max + fma + exp2 + add per score (the softmax, not the MFMAs, is the longer pipe).
        // Register r of tile t is key kb + 32t + 16(r>>3) + 8hh + (r&7).
        const bool need_mask = (kb + 64 > end) || (causal && kb + 63 > q0) || !prefix;
        if (need_mask) {
            // visible keys of this lane's query: key < lim, lim = min(end, causal ? query + 1 : end); per register
            // the key is kb + 8 hh + (32 t + 16 (r >> 3) + (r & 7)), so one compare against a per-lane limit
            const int lim = (causal ? min(end, query + 1) : end) - kb - 8 * hh;
            if (prefix) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[t][r] = (32 * t + 16 * (r >> 3) + (r & 7)) < lim ? st[t][r] : -INFINITY;
            } else {
                const uint8_t* mrow = key_mask + (int64_t)b * seq + kb + 8 * hh;    // arbitrary mask: byte per key
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int off = 32 * t + 16 * (r >> 3) + (r & 7);
                        const bool ok = off < lim && mrow[min(off, seq - 1 - kb - 8 * hh)] != 0;
                        st[t][r] = ok ? st[t][r] : -INFINITY;
                    }
            }
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[t][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);                      // running max of the RAW scores
        const float m_use = m_new == -INFINITY ? 0.f : m_new;       // fully masked so far: p = exp2(-inf) = 0
        const float mc = m_use * scale_log2e;
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * scale_log2e);   // m_run = -inf -> 0
        float lsum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[t][r], scale_log2e, -mc));   // exp(scale * (s - m))
                st[t][r] = p;
                lsum += p;
            }
        lsum += __shfl_xor(lsum, 32, 64);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
        // O^T += V^T . P^T
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)st[ss >> 1][8 * (ss & 1) + j];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sb + v_off[dt][ss][0]));
                const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sb + v_off[dt][ss][1]));
                typedef short short8v __attribute__((ext_vector_type(8)));
                const short8v v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), pf, ot[dt], 0, 0, 0);
            }
        }
    };
    if (n_it > 0) stage(0, 0);
    for (int it = 0; it < n_it; it += 2) {
        step(std::integral_constant<int, 0>{}, it);
        if (it + 1 < n_it) step(std::integral_constant<int, 1>{}, it + 1);
    }

    // ---- epilogue: O^T rows = channels (r&3) + 8(r>>2) + 4hh, col = query ----
    if (query < seq) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        bf16_t* orow = out + ((int64_t)b * seq + query) * ld_out + h * d;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int c0 = dt * 32 + 8 * rg + 4 * hh;
                if (c0 < d) {
                    const float vv[4] = {ot[dt][4 * rg] * inv, ot[dt][4 * rg + 1] * inv, ot[dt][4 * rg + 2] * inv,
                                         ot[dt][4 * rg + 3] * inv};
                    store4(orow + c0, vv);
                }
            }
        if (h == nh - 1)
            for (int c = nh * d + 4 * hh; c < out_cols; c += 8) {
                const float z[4] = {0.f, 0.f, 0.f, 0.f};
                store4(out + ((int64_t)b * seq + query) * ld_out + c, z);
            }
    }
}

int launch_attn_mfma(const void* q, const void* k, const void* v, const uint8_t* key_mask, const int32_t* kv_info,
                     void* out, int64_t ld_out, int B, int T, int nh, int nkv, int d, int dp, float scale, int causal,
                     hipStream_t s) {
    P2T_REQUIRE(d % 4 == 0 && (dp == 32 || dp == 64 || dp == 128) && d <= dp && nh % nkv == 0 && (nh * d) % 4 == 0 && ld_out % 4 == 0,
                "attention(mfma): unsupported shape d=%d dp=%d heads %d/%d", d, dp, nh, nkv);
    const dim3 grid((unsigned)ceil_div(T, 128), (unsigned)nh, (unsigned)B);
    const int out_cols = (int)(round_up((int64_t)nh * d, 64) < ld_out ? round_up((int64_t)nh * d, 64) : ld_out);
    const float sl = scale * 1.4426950408889634f;
#define P2T_ATTN(DPV)                                                                                              \
    attn_mfma_kernel<DPV><<<grid, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_mask, kv_info, \
                                               (bf16_t*)out, ld_out, B, T, nh, nkv, d, sl, causal, out_cols)
    if (dp == 32) P2T_ATTN(32);
    else if (dp == 64) P2T_ATTN(64);
    else P2T_ATTN(128);
#undef P2T_ATTN
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

}  // namespace p2t
