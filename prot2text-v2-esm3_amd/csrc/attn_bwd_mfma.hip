// bf16 flash-attention BACKWARD for gfx950 (stage-2 training through the frozen decoder, llama_train.hip): dq, dk, dv from
// q, k, v, the forward's output o and log-sum-exps, and d_o.  Two kernels, no float atomics (sums stay deterministic):
//
//   dq kernel  -- the forward kernel's structure (attn_mfma.hip): a workgroup = 4 waves = 128 queries of one (batch, head),
//                 K / V tiles of 64 keys by LDS DMA.  Per step:  S'^T = K . Q^T - lse,  dP'^T = V . dO^T - D  (the row constants
//                 ride in as the INITIAL accumulators of the two MFMA chains: the query is on the lane, so they are lane-local),
//                 P = exp2(S'),  dS = P o dP',  dQ^T += K^T . dS^T  with the dS accumulators converted in place to the B operand
//                 (as P is in the forward) and K^T read with the hardware transpose read.  Also writes D = rowsum(dO o O).
//   dk/dv kernel -- roles swapped: a workgroup = 4 waves = 128 keys of one (batch, kv head); K and V fragments live in
//                 registers, Q / dO tiles of 64 queries stream through LDS (all query heads of the GQA group, one after the
//                 other), the key is on the lane:  S' = Q . K^T - lse,  dP' = dO . V^T - D  (row constants now differ per
//                 accumulator ROW: a 64-float strip per tile, DMA'd to LDS beside the tile),  dV^T += dO^T . P,
//                 dK^T += Q^T . dS, both with accumulator -> B-operand conversion and transpose reads of the dO / Q tiles.
//
// Seven MFMA products per (query, key) tile pair instead of five: the price of not summing dq across key blocks with atomics.
// Scores are in log2 units (q stored pre-multiplied by scale * log2 e: kernels.h attention(), log2_scores); lse arrives as the
// natural-log sum-exp the forward wrote and is converted once.  head_dim == padded head_dim in {64, 128}.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace p2t {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
using lds_s4_t = __attribute__((address_space(3))) short4v*;

namespace {

__device__ __forceinline__ int perm23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }      // swap bits 2 and 3

template <int DP>
__device__ __forceinline__ int swz_g(int row) {        // 16-B chunk c of tile row r is stored at c ^ swz_g(r) (attn_mfma.hip)
    if (DP == 64) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    return ((row & 3) << 2) | ((row >> 2) & 3);
}

// LDS DMA, saddr form: wave-uniform 64-bit base + 32-bit lane offset, M0 = LDS address of the wave's piece.  The base is
// uniform by construction (block / step indices) but reaches here through integer divisions and 64-bit multiplies, which the
// compiler evaluates on the vector pipe: readfirstlane hands the assembler the scalar registers the instruction needs.
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
    const uint64_t u = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void dma16(uint32_t voff, const char* base, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(uniform_ptr(base)), "s"(__builtin_amdgcn_readfirstlane(lds)) : "memory");
}
__device__ __forceinline__ void dma4(uint32_t voff, const char* base, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(uniform_ptr(base)), "s"(__builtin_amdgcn_readfirstlane(lds)) : "memory");
}

// the 16 accumulator registers of a 32 x 32 tile as the bf16 B operand of the next MFMA (k = 16 rows of step `ss` of the
// 64-row tile pair): register r of tile t is tile row 32 t + 16 (r >> 3) + 8 hh + (r & 7) (rows fetched through perm23)
__device__ __forceinline__ bf16x8 pack8(const f32x16 (&a)[2], int ss) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16_t)a[ss >> 1][8 * (ss & 1) + j];
    return f;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
template <int DP>
__global__ void __launch_bounds__(256, DP == 64 ? 2 : 1) attn_bwd_dq_mfma_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                  const bf16_t* __restrict__ o, int64_t ld_o, const bf16_t* __restrict__ d_o, int64_t ld_do,
                                                                  const float* __restrict__ lse, const uint8_t* __restrict__ key_mask,
                                                                  const int32_t* __restrict__ kv_info, float* __restrict__ dq, float* __restrict__ Dout,
                                                                  int B, int seq, int nh, int nkv, int causal, float c_out) {
    constexpr int RB = DP * 2, T_BYTES = 64 * RB, STAGE = 2 * T_BYTES, NI = DP / 32, DK = DP / 16, DT = DP / 32;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // heads dealt to the XCDs in groups of eight, an XCD walks the q-tiles of its head back to back (attn_mfma.hip)
    const int n_qt = (seq + 127) >> 7, n_hb = nh * B, hb_full = n_hb & ~7;
    const int id = blockIdx.x;
    int qt, hb;
    if (id < hb_full * n_qt) {
        const int slot = id >> 3;
        hb = (slot / n_qt) * 8 + (id & 7);
        qt = slot % n_qt;
    } else {
        const int rid = id - hb_full * n_qt;
        hb = hb_full + rid / n_qt;
        qt = rid % n_qt;
    }
    const int h = hb % nh, b = hb / nh, hk = h / (nh / nkv);
    const int q0 = qt * 128 + w * 32, lq = lane & 31, hh = lane >> 5;
    const int query = q0 + lq, qr = query < seq ? query : seq - 1;
    int end = kv_info[b];
    const int prefix = kv_info[B + b];
    if (causal) end = min(end, qt * 128 + 128);
    const int n_it = (end + 63) >> 6;
    const bf16_t* kbase = k + ((int64_t)(b * nkv + hk) * seq) * DP;
    const bf16_t* vbase = v + ((int64_t)(b * nkv + hk) * seq) * DP;

    // ---- per-query operands: Q and dO fragments (B operands: lane holds row `query`, k elements kk*16 + 8 hh .. +8), D, lse ----
    bf16x8 qf[DK], dof[DK];
    float Dq = 0.f;
    {
        const bf16_t* qrow = q + ((int64_t)(b * nh + h) * seq + qr) * DP + 8 * hh;
        const bf16_t* dorow = d_o + ((int64_t)b * seq + qr) * ld_do + (int64_t)h * DP + 8 * hh;
        const bf16_t* orow = o + ((int64_t)b * seq + qr) * ld_o + (int64_t)h * DP + 8 * hh;
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) {
            qf[kk] = *reinterpret_cast<const bf16x8*>(qrow + kk * 16);
            dof[kk] = *reinterpret_cast<const bf16x8*>(dorow + kk * 16);
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(orow + kk * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) Dq = fmaf((float)dof[kk][j], (float)of[j], Dq);
        }
    }
    Dq += __shfl_xor(Dq, 32, 64);
    float l2 = lse[(int64_t)(b * nh + h) * seq + qr] * kLog2e;          // +inf: no visible key -> every p = 0
    if (query >= seq) l2 = INFINITY;
    if (query < seq && hh == 0) Dout[(int64_t)(b * nh + h) * seq + query] = Dq;
    // every tracked vector-memory operation done before the loop (no compiler vmcnt wait inside a step: attn_mfma.hip)
#pragma unroll
    for (int kk = 0; kk < DK; ++kk) asm volatile("" : "+v"(qf[kk]), "+v"(dof[kk]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x16 init_s, init_d;
#pragma unroll
    for (int r = 0; r < 16; ++r) { init_s[r] = -l2; init_d[r] = -Dq; }

    // ---- staging: K and V tiles share the layout (attn_mfma.hip) ----
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t s_voff[NI];
    int s_row[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int byte = (w + 4 * i) * 1024 + lane * 16;
        const int row = byte / RB, p = (byte % RB) >> 4;
        s_row[i] = row;
        s_voff[i] = (uint32_t)(row * RB + ((p ^ swz_g<DP>(row)) << 4));
    }
    auto stage = [&](int buf, int it) {
        const int kb = it * 64;
        const uint32_t sb = lds0 + buf * STAGE;
        const char* kt = (const char*)(kbase + (int64_t)kb * DP);
        const char* vt = (const char*)(vbase + (int64_t)kb * DP);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const uint32_t vo = kb + s_row[i] < seq ? s_voff[i] : (uint32_t)((seq - 1 - kb) * RB) + (s_voff[i] - (uint32_t)(s_row[i] * RB));
            dma16(vo, kt, sb + (w + 4 * i) * 1024);
            dma16(vo, vt, sb + (w + 4 * i) * 1024 + T_BYTES);
        }
    };
    int k_off[2][DK];                          // row fragments: tile row 32 t + perm23(lq), chunk 2 kk + hh
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = t * 32 + perm23(lq);
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) k_off[t][kk] = row * RB + (((kk * 2 + hh) ^ swz_g<DP>(row)) << 4);
    }
    int t_off[DT][4][2];                       // transposed fragments (ds_read_b64_tr_b16): channels dt*32.., rows ss*16 + 8 hh + 4 r + qq
    {
        const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int ss = 0; ss < 4; ++ss)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int row = ss * 16 + 8 * hh + 4 * r + qq, col = dt * 32 + 16 * g1 + 4 * pp;
                    t_off[dt][ss][r] = row * RB + ((((col >> 3)) ^ swz_g<DP>(row)) << 4) + (col & 7) * 2;
                }
    }
    f32x16 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;

    auto step = [&](auto bufc, int it) {
        constexpr int BUF = decltype(bufc)::value;
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // this wave's pieces of tile `it` have landed ...
        __builtin_amdgcn_s_barrier();                       // ... every wave's have, and all are done with the other buffer
        if (it + 1 < n_it) stage(BUF ^ 1, it + 1);
        const char* sb = smem + BUF * STAGE;
        const int kb = it * 64;
        f32x16 st[2], dp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sb + k_off[t][kk]);
                st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], kk == 0 ? init_s : st[t], 0, 0, 0);
            }
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sb + T_BYTES + k_off[t][kk]);
                dp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[kk], kk == 0 ? init_d : dp[t], 0, 0, 0);
            }
        }
        const bool need_mask = (kb + 64 > end) || (causal && kb + 63 > q0) || !prefix;
        if (need_mask) {
            const int lim = (causal ? min(end, query + 1) : end) - kb - 8 * hh;
            const uint8_t* mrow = key_mask + (int64_t)b * seq + kb + 8 * hh;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int off = 32 * t + 16 * (r >> 3) + (r & 7);
                    bool ok = off < lim;
                    if (!prefix) ok = ok && mrow[min(off, seq - 1 - kb - 8 * hh)] != 0;
                    st[t][r] = ok ? st[t][r] : -INFINITY;
                }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[t][r] = __builtin_amdgcn_exp2f(st[t][r]) * dp[t][r];        // dS = P o (dP - D)
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            const bf16x8 dsf = pack8(st, ss);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sb + t_off[dt][ss][0]));
                const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sb + t_off[dt][ss][1]));
                const short8v k8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k8), dsf, acc[dt], 0, 0, 0);
            }
        }
    };
    if (n_it > 0) stage(0, 0);
    for (int it = 0; it < n_it; it += 2) {
        step(std::integral_constant<int, 0>{}, it);
        if (it + 1 < n_it) step(std::integral_constant<int, 1>{}, it + 1);
    }
    // dQ^T rows = channels (r & 3) + 8 (r >> 2) + 4 hh, column = query
    if (query < seq) {
        float* row = dq + ((int64_t)(b * nh + h) * seq + query) * DP;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float vv[4] = {acc[dt][4 * rg] * c_out, acc[dt][4 * rg + 1] * c_out, acc[dt][4 * rg + 2] * c_out, acc[dt][4 * rg + 3] * c_out};
                store4(row + dt * 32 + 8 * rg + 4 * hh, vv);
            }
    }
}

// ---------------------------------------------------------------------------------------------
template <int DP>
__global__ void __launch_bounds__(256, 1) attn_bwd_dkv_mfma_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                   const bf16_t* __restrict__ d_o, int64_t ld_do, const float* __restrict__ lse,
                                                                   const float* __restrict__ D, const uint8_t* __restrict__ key_mask,
                                                                   float* __restrict__ dk, float* __restrict__ dv, int B, int seq, int nh, int nkv,
                                                                   int causal, float c_out) {
    constexpr int RB = DP * 2, T_BYTES = 64 * RB, STAT = 2 * T_BYTES, STAGE = 2 * T_BYTES + 512, NI = DP / 32, DK = DP / 16, DT = DP / 32;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_kt = (seq + 127) >> 7;
    const int kt = __builtin_amdgcn_readfirstlane(blockIdx.x % n_kt), hk = __builtin_amdgcn_readfirstlane((blockIdx.x / n_kt) % nkv),
              b = __builtin_amdgcn_readfirstlane(blockIdx.x / (n_kt * nkv));
    const int rep = nh / nkv;
    const int k0 = kt * 128 + w * 32, lq = lane & 31, hh = lane >> 5;
    const int key = k0 + lq, kr = key < seq ? key : seq - 1;
    const bool kvalid = key < seq && key_mask[(int64_t)b * seq + kr] != 0;

    // ---- K and V fragments of this lane's key (B operands), kept for the whole kernel ----
    bf16x8 kf[DK], vf[DK];
    {
        const bf16_t* krow = k + ((int64_t)(b * nkv + hk) * seq + kr) * DP + 8 * hh;
        const bf16_t* vrow = v + ((int64_t)(b * nkv + hk) * seq + kr) * DP + 8 * hh;
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) {
            kf[kk] = *reinterpret_cast<const bf16x8*>(krow + kk * 16);
            vf[kk] = *reinterpret_cast<const bf16x8*>(vrow + kk * 16);
        }
    }
#pragma unroll
    for (int kk = 0; kk < DK; ++kk) asm volatile("" : "+v"(kf[kk]), "+v"(vf[kk]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- the stream of steps: for every query head of the group, the 64-query tiles from the first one a key of this block can
    // see (causal) to the end ----
    const int n_qt = (seq + 63) >> 6, it0 = causal ? (kt * 128) >> 6 : 0, per_head = n_qt - it0, n_steps = rep * per_head;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t q_voff[NI], o_voff[NI];
    int s_row[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int byte = (w + 4 * i) * 1024 + lane * 16;
        const int row = byte / RB, p = (byte % RB) >> 4;
        s_row[i] = row;
        q_voff[i] = (uint32_t)(((p ^ swz_g<DP>(row)) << 4));                  // + row * row stride, added per step (rows are clamped)
        o_voff[i] = q_voff[i];
    }
    auto stage = [&](int buf, int sidx) {
        // (integer division is VALU code: readfirstlane makes the results scalars again, as the DMA's base / M0 operands need)
        const int hi = __builtin_amdgcn_readfirstlane(sidx / per_head);
        const int h = hk * rep + hi, qb = (it0 + sidx - hi * per_head) * 64;
        const uint32_t sb = lds0 + buf * STAGE;
        const char* qt = (const char*)(q + ((int64_t)(b * nh + h) * seq + qb) * DP);
        const char* ot = (const char*)(d_o + ((int64_t)b * seq + qb) * ld_do + (int64_t)h * DP);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = qb + s_row[i] < seq ? s_row[i] : seq - 1 - qb;     // rows past the end re-read the last row (masked below)
            dma16(q_voff[i] + (uint32_t)(row * RB), qt, sb + (w + 4 * i) * 1024);
            dma16(o_voff[i] + (uint32_t)row * (uint32_t)(ld_do * 2), ot, sb + (w + 4 * i) * 1024 + T_BYTES);
        }
        if (w < 2) {                                       // the tile's 64 lse (wave 0) / D (wave 1) values: one 256-byte DMA each
            const float* src = (w == 0 ? lse : D) + (int64_t)(b * nh + h) * seq;
            const int qi = qb + lane < seq ? qb + lane : seq - 1;
            dma4((uint32_t)qi * 4u, (const char*)src, sb + STAT + w * 256);
        }
    };
    int k_off[2][DK];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = t * 32 + perm23(lq);
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) k_off[t][kk] = row * RB + (((kk * 2 + hh) ^ swz_g<DP>(row)) << 4);
    }
    int t_off[DT][4][2];
    {
        const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int ss = 0; ss < 4; ++ss)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int row = ss * 16 + 8 * hh + 4 * r + qq, col = dt * 32 + 16 * g1 + 4 * pp;
                    t_off[dt][ss][r] = row * RB + ((((col >> 3)) ^ swz_g<DP>(row)) << 4) + (col & 7) * 2;
                }
    }
    f32x16 ak[DT], av[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ak[dt][r] = 0.f; av[dt][r] = 0.f; }

    auto step = [&](auto bufc, int sidx) {
        constexpr int BUF = decltype(bufc)::value;
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        if (sidx + 1 < n_steps) stage(BUF ^ 1, sidx + 1);
        const char* sb = smem + BUF * STAGE;
        const int qb = (it0 + sidx - __builtin_amdgcn_readfirstlane(sidx / per_head) * per_head) * 64;
        // row constants: accumulator register r of tile t is query qb + 32 t + 16 (r >> 3) + 8 hh + (r & 7)
        f32x16 st[2], dp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float* ls = reinterpret_cast<const float*>(sb + STAT) + 32 * t + 16 * g + 8 * hh;
                const float4 l0 = *reinterpret_cast<const float4*>(ls), l1 = *reinterpret_cast<const float4*>(ls + 4);
                const float4 d0 = *reinterpret_cast<const float4*>(ls + 64), d1 = *reinterpret_cast<const float4*>(ls + 68);
                st[t][8 * g + 0] = -l0.x * kLog2e; st[t][8 * g + 1] = -l0.y * kLog2e; st[t][8 * g + 2] = -l0.z * kLog2e; st[t][8 * g + 3] = -l0.w * kLog2e;
                st[t][8 * g + 4] = -l1.x * kLog2e; st[t][8 * g + 5] = -l1.y * kLog2e; st[t][8 * g + 6] = -l1.z * kLog2e; st[t][8 * g + 7] = -l1.w * kLog2e;
                dp[t][8 * g + 0] = -d0.x; dp[t][8 * g + 1] = -d0.y; dp[t][8 * g + 2] = -d0.z; dp[t][8 * g + 3] = -d0.w;
                dp[t][8 * g + 4] = -d1.x; dp[t][8 * g + 5] = -d1.y; dp[t][8 * g + 6] = -d1.z; dp[t][8 * g + 7] = -d1.w;
            }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) {
                const bf16x8 qf = *reinterpret_cast<const bf16x8*>(sb + k_off[t][kk]);
                st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[kk], st[t], 0, 0, 0);
            }
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) {
                const bf16x8 of = *reinterpret_cast<const bf16x8*>(sb + T_BYTES + k_off[t][kk]);
                dp[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of, vf[kk], dp[t], 0, 0, 0);
            }
        }
        // visibility: the key must be valid (lane), not after the query (causal), and the query inside the sequence
        const bool tail = qb + 64 > seq, diag = causal && qb < k0 + 32;
        if (tail || diag || __builtin_amdgcn_ballot_w64(!kvalid) != 0) {          // wave-uniform
            const int lo = (causal ? key : 0) - qb - 8 * hh, hi_ = seq - qb - 8 * hh;      // visible iff lo <= off < hi
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int off = 32 * t + 16 * (r >> 3) + (r & 7);
                    st[t][r] = (kvalid && off >= lo && off < hi_) ? st[t][r] : -INFINITY;
                }
        }
        f32x16 ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[t][r]);
                st[t][r] = p;
                ds[t][r] = p * dp[t][r];
            }
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            const bf16x8 pf = pack8(st, ss), dsf = pack8(ds, ss);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const short4v olo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sb + T_BYTES + t_off[dt][ss][0]));
                const short4v ohi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sb + T_BYTES + t_off[dt][ss][1]));
                const short8v o8 = {olo[0], olo[1], olo[2], olo[3], ohi[0], ohi[1], ohi[2], ohi[3]};
                av[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o8), pf, av[dt], 0, 0, 0);
                const short4v qlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sb + t_off[dt][ss][0]));
                const short4v qhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sb + t_off[dt][ss][1]));
                const short8v q8 = {qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
                ak[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, q8), dsf, ak[dt], 0, 0, 0);
            }
        }
    };
    if (n_steps > 0) stage(0, 0);
    for (int sidx = 0; sidx < n_steps; sidx += 2) {
        step(std::integral_constant<int, 0>{}, sidx);
        if (sidx + 1 < n_steps) step(std::integral_constant<int, 1>{}, sidx + 1);
    }
    if (key < seq) {
        float* rk = dk + ((int64_t)(b * nkv + hk) * seq + key) * DP;
        float* rv = dv + ((int64_t)(b * nkv + hk) * seq + key) * DP;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float kk4[4] = {ak[dt][4 * rg] * c_out, ak[dt][4 * rg + 1] * c_out, ak[dt][4 * rg + 2] * c_out, ak[dt][4 * rg + 3] * c_out};
                const float vv4[4] = {av[dt][4 * rg], av[dt][4 * rg + 1], av[dt][4 * rg + 2], av[dt][4 * rg + 3]};
                store4(rk + dt * 32 + 8 * rg + 4 * hh, kk4);
                store4(rv + dt * 32 + 8 * rg + 4 * hh, vv4);
            }
    }
}

// log2_scores form, bf16, head_dim == dp in {64, 128}; P2T_ERR_UNSUPPORTED otherwise (the caller runs the exact kernels).
int launch_attn_bwd_mfma(const void* q, const void* k, const void* v, const void* o, int64_t ld_o, const void* d_o, int64_t ld_do, const float* lse,
                         const uint8_t* key_mask, const int32_t* kv_info, float* dq, float* dk, float* dv, float* D, int B, int T, int nh, int nkv,
                         int d, int dp, int causal, hipStream_t s) {
    if (d != dp || (dp != 64 && dp != 128) || ld_o % 8 || ld_do % 8 || (int64_t)64 * ld_do * 2 >= ((int64_t)1 << 31)) return P2T_ERR_UNSUPPORTED;
    const unsigned gq = (unsigned)(ceil_div(T, 128) * nh * B), gk = (unsigned)(ceil_div(T, 128) * nkv * B);
#define P2T_BWD(DPV)                                                                                                                     \
    attn_bwd_dq_mfma_kernel<DPV><<<gq, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, ld_o, (const bf16_t*)d_o,  \
                                                    ld_do, lse, key_mask, kv_info, dq, D, B, T, nh, nkv, causal, kLn2);                      \
    attn_bwd_dkv_mfma_kernel<DPV><<<gk, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, ld_do, lse, D, key_mask, \
                                                     dk, dv, B, T, nh, nkv, causal, kLn2)
    if (dp == 64) { P2T_BWD(64); } else { P2T_BWD(128); }
#undef P2T_BWD
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

}  // namespace p2t
