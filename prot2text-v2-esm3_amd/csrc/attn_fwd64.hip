// bf16 flash attention forward for gfx950, head_dim padded to 64 (ESM2-3B: 40 heads x 64; Llama-3.2-1B: 32 / 8 heads x 64):
// the hand-placed form.  Same semantics as attn_mfma.hip (log2-scores q, key-padding / causal / GQA, optional log-sum-exp).
//
// One workgroup = 4 wavefronts = 256 queries of one (batch, head), ONE wave per SIMD with the whole 512-register file; a wave
// owns two 32-query tiles (A, B).  Per 64-key tile j the wave runs two segments of 16 MFMAs (v_mfma_f32_32x32x16_bf16): the
// softmax of one query tile (exp2 / row sum / bf16 pack, the next tile's row maximum) is issued on the vector pipe in the gaps
// of the OTHER tile's QK^T and PV MFMAs, so neither pipe waits for the other inside a wave (attn_mfma.hip leaves that overlap
// to three co-resident waves and the compiler's order: 0.30 of the MFMA peak, issue-bound).  K / V fragments
// live in AGPRs, double-buffered by tile parity, so both query tiles share every LDS fragment read; the K / V tiles arrive by
// LDS-DMA into two 4-slot rings (K five tiles ahead, V three), one barrier per tile with a counted vmcnt.
//
// The loop is one asm statement with literal registers, written by tools/gen_attn_fwd64.py (segment anatomy, register map and
// the rare paths -- reference rescale, hidden keys -- are documented there); this file sets up its pinned inputs and runs the
// epilogue (normalise, stage O through LDS, whole-row stores).
#include "common.h"
#include "kernels.h"

#include "attn_fwd64_body.inc"

namespace p2t {

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kA64Slot = 8192, kA64Slots = 4;
constexpr int kA64Ring = 2 * kA64Slots * kA64Slot;           // K ring + V ring
constexpr int kA64Lds = kA64Ring + 4 * 8192;                 // + one 64-query x 128-B output stage per wave

__device__ __forceinline__ int a64_perm23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
__device__ __forceinline__ int a64_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

// What one query block needs from the index arithmetic (wave-uniform unless noted).
struct A64Block {
    int b, h, q0, n_it, flags;
    uint64_t kptr, vptr, qptr, mptr;
    i32x4 misc;                   // per lane: Q row offsets of tiles A / B, causal limits of tiles A / B
};

template <bool CAUSAL>
__device__ __forceinline__ A64Block a64_block(int id, const bf16_t* q, const bf16_t* k, const bf16_t* v, const uint8_t* key_mask, const int32_t* kv_info,
                                              int B, int seq, int nh, int nkv, int w, int lane) {
    // XCD-aware block order (attn_mfma.hip): the query blocks of one (batch, head) share an L2
    const int n_qb = (seq + 255) >> 8, n_hb = nh * B, hb_full = n_hb & ~7;
    int qb, hb;
    if (id < hb_full * n_qb) {
        const int slot = id >> 3;
        hb = (slot / n_qb) * 8 + (id & 7);
        qb = slot % n_qb;
    } else {
        const int rid = id - hb_full * n_qb;
        hb = hb_full + rid / n_qb;
        qb = rid % n_qb;
    }
    A64Block r;
    r.h = hb % nh;
    r.b = hb / nh;
    const int hk = r.h / (nh / nkv);
    r.q0 = qb * 256 + w * 64;
    int end = kv_info[r.b];
    const int prefix = kv_info[B + r.b];
    if (CAUSAL) end = min(end, qb * 256 + 256);
    end = min(end, seq);
    r.n_it = (end + 63) >> 6;
    r.flags = ((end & 63) != 0 ? 1 : 0) | (prefix ? 0 : 2);
    r.kptr = (uint64_t)(k + ((int64_t)(r.b * nkv + hk) * seq) * 64);
    r.vptr = (uint64_t)(v + ((int64_t)(r.b * nkv + hk) * seq) * 64);
    r.qptr = (uint64_t)(q + ((int64_t)(r.b * nh + r.h) * seq) * 64);
    r.mptr = (uint64_t)(key_mask + (int64_t)r.b * seq);
    const int lq = lane & 31, hh = lane >> 5;
    r.misc[0] = min(r.q0 + lq, seq - 1) * 128 + 16 * hh;
    r.misc[1] = min(r.q0 + 32 + lq, seq - 1) * 128 + 16 * hh;
    r.misc[2] = r.q0 + lq + 1 - 8 * hh;
    r.misc[3] = r.q0 + 32 + lq + 1 - 8 * hh;
    return r;
}

template <bool CAUSAL, int DIAG = 0>
__global__ void __launch_bounds__(256, 1) attn_fwd64_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                             const bf16_t* __restrict__ v, const uint8_t* __restrict__ key_mask,
                                                             const int32_t* __restrict__ kv_info, bf16_t* __restrict__ out, int64_t ld_out,
                                                             int B, int seq, int nh, int nkv, int d, int out_cols, float* lse, int n_blocks) {
    __shared__ __attribute__((aligned(16))) char smem[kA64Lds];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, hh = lane >> 5;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    // ---- per-lane constants of the whole launch ----
    // fragment read addresses (slot 0): K row pi(lq) (+ 32 per tile half), 16-byte chunk (2 kk + hh) ^ g(row)
    i32x4 ka, va, voffs;
    {
        const int row = a64_perm23(lq);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ka[kk] = (int)(lds0 + row * 128 + (((kk * 2 + hh) ^ a64_swz(row)) << 4));
        // V^T operand: 16-lane group (lane >> 4) reads the 4 x 16 block rows 8 hh + 4 r + qq (+ 16 per key step), columns
        // dt * 32 + 16 g1 + 4 pp .. +3  (ds_read_b64_tr_b16)
        const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int vrow = 8 * hh + 4 * r + qq, col = dt * 32 + 16 * g1 + 4 * pp;
                va[dt * 2 + r] = (int)(lds0 + kA64Slots * kA64Slot + vrow * 128 + (((col >> 3) ^ a64_swz(vrow)) << 4) + (col & 7) * 2);
            }
        // DMA source offsets of this wave's two 1-KiB pieces (rows 8 (w + 4 i) .. + 7 of a tile; the LDS image is lane-linear,
        // so the swizzle is applied to the source address), and the same with rows past the sequence end re-reading its last row
        const int tail0 = (seq >> 6) << 6;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int byte = (w + 4 * i) * 1024 + lane * 16;
            const int row = byte >> 7, p = (byte & 127) >> 4;
            const int chunk = (p ^ a64_swz(row)) << 4;
            voffs[i] = row * 128 + chunk;
            voffs[2 + i] = min(row, max(seq - 1 - tail0, 0)) * 128 + chunk;
        }
    }
    const int s_ldsk = (int)(lds0 + w * 1024), s_seq = seq;
    char* stage = smem + kA64Ring + w * 8192;                 // [64 queries][128 B], 16-byte chunk c of row r at c ^ (r & 7)

    // Persistent: one workgroup per CU walks the query blocks id = blockIdx.x, + gridDim.x, ... (a new workgroup of this size
    // starts ~2 us after its predecessor ends; gridDim.x is a multiple of 8, so a workgroup's blocks stay on its XCD's residue).
    // MAIN(block) is the K / V loop; it leaves O^T, the row sums and the reference exponents in registers for the epilogue.  Its
    // K / V stream is CONTINUOUS across blocks: the last iterations of a block, whose ring slots would sit idle, request the next
    // block's tiles K'0..K'3 / V'0..V'2 (the twelve-piece burst a block used to open with cost its waves ~4 000 cycles of DMA issue:
    // profiles/r04_attn64_v4_diag.log), so the next MAIN finds them in the ring at phase c0 = (tiles so far) & 3.  PREQ(block) then
    // only requests the block's Q fragments; it runs BEFORE the previous block's epilogue, whose ~1.7 us cover their flight.  A
    // workgroup's first block, and the successor of a block with fewer than five tiles, opens with PRE (Q + the first six tiles).
    i32x4 dv;                     // DMA source offsets in use (asm state carried between the statements)
    f32x16 qa, qb;                // Q fragments, in flight between PRE / PREQ and MAIN: nothing but those statements touches them
#define P2T_ATTN64_RUN_PRE(BK, C0V)                                                                                                     \
    asm volatile(P2T_ATTN64_PRE                                                                                                         \
                 : "={v[146:149]}"(dv), "={a[64:79]}"(qa), "={a[80:95]}"(qb), "+{s[36:37]}"(BK.kptr), "+{s[38:39]}"(BK.vptr)          \
                 : "{s[40:41]}"(BK.qptr), "{s44}"(BK.n_it), "{s45}"(s_ldsk), "{s46}"(s_seq), "{s91}"(C0V), "{v[150:153]}"(voffs),     \
                   "{v[154:157]}"(BK.misc)                                                                                              \
                 : P2T_ATTN64_PRE_CLOBBERS)
#define P2T_ATTN64_RUN_PREQ(BK)                                                                                                         \
    asm volatile(P2T_ATTN64_PREQ : "={a[64:79]}"(qa), "={a[80:95]}"(qb) : "{s[40:41]}"(BK.qptr), "{v[154:157]}"(BK.misc) : "memory")
    int id = blockIdx.x;
    int c0 = 0, pref = 0;          // ring phase of the block's tile 0; 1 = its first tiles came through the predecessor's stream
    A64Block cur = a64_block<CAUSAL>(id, q, k, v, key_mask, kv_info, B, seq, nh, nkv, w, lane);
    P2T_ATTN64_RUN_PRE(cur, c0);
    while (true) {
        const int next = id + (int)gridDim.x;
        const bool has_next = next < n_blocks;
        A64Block nxt = cur;
        if (has_next) nxt = a64_block<CAUSAL>(next, q, k, v, key_mask, kv_info, B, seq, nh, nkv, w, lane);
        // the stream hands tiles over only from a block long enough to have idle slots for all of them (K'0 rides in iteration n_it - 5)
        const int s_nitn = (has_next && cur.n_it >= 5) ? nxt.n_it : 0;
        uint64_t knext = nxt.kptr, vnext = nxt.vptr;
        f32x16 oA0, oA1, oB0, oB1;
        float lA, lB;                 // row sums (the ones-row of the PV product: every lane holds its query's whole sum)
        f32x2 mref;
#define P2T_ATTN64_OUTS "={a[0:15]}"(oA0), "={a[16:31]}"(oA1), "={a[32:47]}"(oB0), "={a[48:63]}"(oB1), "={a224}"(lA), "={a240}"(lB), "={v[132:133]}"(mref), \
                        "+{s[36:37]}"(cur.kptr), "+{s[38:39]}"(cur.vptr), "+{v[146:149]}"(dv)
#define P2T_ATTN64_INS "{s[42:43]}"(cur.mptr), "{s44}"(cur.n_it), "{s45}"(s_ldsk), "{s46}"(s_seq), "{s47}"(cur.flags), "{s49}"(cur.q0), \
                       "{s[86:87]}"(knext), "{s[88:89]}"(vnext), "{s90}"(s_nitn), "{s91}"(c0), "{s92}"(pref),                           \
                       "{v[138:141]}"(ka), "{v[142:145]}"(va), "{v[150:153]}"(voffs), "{v[154:157]}"(cur.misc), "{a[64:79]}"(qa), "{a[80:95]}"(qb)
#ifdef P2T_LAB
        i32x4 cyc0, cyc1;
        long long t_in = 0, r_in = 0;
        if constexpr (DIAG != 0) {
            // diagnostic build: cycles per phase of this wave (tools/gen_attn_fwd64.py, stamp()), written where the log-sum-exps would go
            t_in = __builtin_amdgcn_s_memtime();
            r_in = __builtin_amdgcn_s_memrealtime();
#define P2T_ATTN64_DIAG(N)                                                                                                                    \
            if constexpr (DIAG == N)                                                                                                              \
                asm volatile(P2T_ATTN64_BODY_DIAG##N : P2T_ATTN64_OUTS, "={s[72:75]}"(cyc0), "={s[76:79]}"(cyc1) : P2T_ATTN64_INS : P2T_ATTN64_CLOBBERS_DIAG);
            P2T_ATTN64_DIAG(1) P2T_ATTN64_DIAG(2) P2T_ATTN64_DIAG(3) P2T_ATTN64_DIAG(4) P2T_ATTN64_DIAG(5) P2T_ATTN64_DIAG(6) P2T_ATTN64_DIAG(7) P2T_ATTN64_DIAG(8) P2T_ATTN64_DIAG(9)
#undef P2T_ATTN64_DIAG
        } else
#endif
        if constexpr (CAUSAL)
            asm volatile(P2T_ATTN64_BODY_1 : P2T_ATTN64_OUTS : P2T_ATTN64_INS : P2T_ATTN64_CLOBBERS);
        else
            asm volatile(P2T_ATTN64_BODY_0 : P2T_ATTN64_OUTS : P2T_ATTN64_INS : P2T_ATTN64_CLOBBERS);
#undef P2T_ATTN64_OUTS
#undef P2T_ATTN64_INS
        const A64Block done = cur;
        c0 = (c0 + done.n_it) & 3;
        if (has_next) {
            cur = nxt;
            if (s_nitn > 0) {
                // the first tiles of `cur` are in the ring (K'0..K'3, V'0..V'2 as far as they exist): its streams resume behind them
                pref = 1;
                cur.kptr += 4 * (uint64_t)kA64Slot;
                cur.vptr += 3 * (uint64_t)kA64Slot;
                P2T_ATTN64_RUN_PREQ(cur);
            } else {
                pref = 0;
                __builtin_amdgcn_s_barrier();          // every wave is done with the previous block's K / V ring
                P2T_ATTN64_RUN_PRE(cur, c0);
            }
        }
#ifdef P2T_LAB
        if constexpr (DIAG != 0) {
            if (lane == 0) {
                float* dst = lse + ((int64_t)id * 4 + w) * 16;
                for (int i = 0; i < 4; ++i) { dst[i] = (float)(unsigned)cyc0[i]; dst[4 + i] = (float)(unsigned)cyc1[i]; }
                dst[8] = (float)(unsigned)(__builtin_amdgcn_s_memtime() - t_in);
                dst[9] = (float)done.n_it;
                // wall clock (100 MHz) at entry / exit and where the block ran: the gaps between the blocks of one CU and the clock it held
                const long long r_out = __builtin_amdgcn_s_memrealtime();
                dst[10] = (float)(unsigned)(r_in & 0xFFFFFF);
                dst[11] = (float)(unsigned)(r_out & 0xFFFFFF);
                dst[12] = (float)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFFFF);          // HW_ID (wave / simd / cu / sh / se ...)
                dst[13] = (float)__builtin_amdgcn_s_getreg((31 << 11) | 20);                      // XCC_ID
            }
            t_in = __builtin_amdgcn_s_memtime();          // (re-used: start of the epilogue)
        }
#endif
        // ---- epilogue: O^T rows = channels (r & 3) + 8 (r >> 2) + 4 hh (+ 32 per d-tile), column = query ----
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const float l = x == 0 ? lA : lB;
            const float m = mref[x];
            const float inv = l > 0.f ? __builtin_amdgcn_rcpf(l) : 0.f;
            const int query = done.q0 + 32 * x + lq;
            if (DIAG == 0 && lse && hh == 0 && query < seq)
                lse[(int64_t)(done.b * nh + done.h) * seq + query] = l > 0.f ? kLn2 * (m + __log2f(l)) : INFINITY;
            const int row = 32 * x + lq;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const f32x16& o = x == 0 ? (dt == 0 ? oA0 : oA1) : (dt == 0 ? oB0 : oB1);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int c16 = dt * 4 + rg;
                    *reinterpret_cast<uint2*>(stage + row * 128 + ((c16 ^ (row & 7)) << 4) + 8 * hh) =
                        make_uint2(pack_bf16x2(o[4 * rg] * inv, o[4 * rg + 1] * inv), pack_bf16x2(o[4 * rg + 2] * inv, o[4 * rg + 3] * inv));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the stage is private to the wave: program order + this wait is enough
        const int c = lane & 7;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 8 + (lane >> 3), query = done.q0 + row;
            const uint4 val = *reinterpret_cast<const uint4*>(stage + row * 128 + ((c ^ (row & 7)) << 4));
            if (query < seq) {
                bf16_t* orow = out + ((int64_t)done.b * seq + query) * ld_out;
                if (c * 8 < d) *reinterpret_cast<uint4*>(orow + done.h * d + c * 8) = val;
                if (done.h == nh - 1)
                    for (int cc = nh * d + c * 8; cc < out_cols; cc += 64) *reinterpret_cast<uint4*>(orow + cc) = make_uint4(0, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the stage is read out before the next block's epilogue rewrites it
#ifdef P2T_LAB
        if constexpr (DIAG != 0) {
            if (lane == 0) lse[((int64_t)id * 4 + w) * 16 + 14] = (float)(unsigned)(__builtin_amdgcn_s_memtime() - t_in);      // cycles of the epilogue
        }
#endif
        if (!has_next) break;
        id = next;
    }
#undef P2T_ATTN64_RUN_PRE
#undef P2T_ATTN64_RUN_PREQ
}

bool attn_fwd64_eligible(int64_t ld_out, int T, int nh, int nkv, int d, int dp, int log2_scores) {
    return dp == 64 && log2_scores && d % 8 == 0 && d <= 64 && nh % nkv == 0 && ld_out % 8 == 0 && (nh * d) % 8 == 0 && T >= 1;
}

int launch_attn_fwd64(const void* q, const void* k, const void* v, const uint8_t* key_mask, const int32_t* kv_info, void* out, int64_t ld_out,
                      int B, int T, int nh, int nkv, int d, int causal, float* lse, hipStream_t s) {
    P2T_REQUIRE(attn_fwd64_eligible(ld_out, T, nh, nkv, d, 64, 1), "attention(fwd64): unsupported shape d=%d heads %d/%d", d, nh, nkv);
    const int n_blocks = (int)(ceil_div(T, 256) * nh * B);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n & ~7 ? n & ~7 : 8;
    }
    const dim3 grid((unsigned)(n_blocks < cus ? n_blocks : cus));
    const int out_cols = (int)(round_up((int64_t)nh * d, 64) < ld_out ? round_up((int64_t)nh * d, 64) : ld_out);
#ifdef P2T_LAB
    if (causal >= 2) {          // lab build: stamped kernels (non-causal), variant causal - 1; lse = the stamp buffer, f32 [grid * 4 waves * 16]
#define P2T_ATTN64_DIAG(N)                                                                                                                     \
        if (causal - 1 == N)                                                                                                                   \
            attn_fwd64_kernel<false, N><<<grid, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_mask, kv_info, (bf16_t*)out, \
                                                             ld_out, B, T, nh, nkv, d, out_cols, lse, n_blocks);
        P2T_ATTN64_DIAG(1) P2T_ATTN64_DIAG(2) P2T_ATTN64_DIAG(3) P2T_ATTN64_DIAG(4) P2T_ATTN64_DIAG(5) P2T_ATTN64_DIAG(6) P2T_ATTN64_DIAG(7) P2T_ATTN64_DIAG(8) P2T_ATTN64_DIAG(9)
#undef P2T_ATTN64_DIAG
        P2T_LAUNCH_CHECK();
        return P2T_OK;
    }
#endif
    if (causal)
        attn_fwd64_kernel<true><<<grid, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_mask, kv_info, (bf16_t*)out, ld_out,
                                                     B, T, nh, nkv, d, out_cols, lse, n_blocks);
    else
        attn_fwd64_kernel<false><<<grid, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_mask, kv_info, (bf16_t*)out, ld_out,
                                                      B, T, nh, nkv, d, out_cols, lse, n_blocks);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

}  // namespace p2t
