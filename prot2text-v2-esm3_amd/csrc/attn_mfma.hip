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
// K / V tiles are staged with global_load_lds (three LDS buffers, two tiles in flight), the bank swizzle applied on the
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

// L2S ("log2 scores"): the caller stored q pre-multiplied by scale * log2(e) (the towers fold it into the q-scale of the QKV
// epilogue, where q is rounded to bf16 once either way), so K . Q^T already IS the exponent in base 2.  The running reference m
// of the exponentials is then folded into the QK^T MFMA as its accumulator INPUT (16 registers holding -m, rewritten only when
// the lazy rescale moves m), and the softmax per score shrinks from fma + exp2 + add to exp2 + add: the vector pipe, not the matrix
// pipe, bounds this kernel.
template <int DP, bool L2S>
// dp = 64: three waves per SIMD (three 48 KiB blocks per CU); the L2S form needs 16 registers more for -m and is held to that
// budget (168 registers: two scalar-like values per step reload from scratch) -- at two waves per SIMD it measured 9 % slower
// (profiles/r03_attn_ab.log)
__global__ void __launch_bounds__(256, DP == 64 ? 3 : 2) attn_mfma_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, const uint8_t* __restrict__ key_mask,
                                                        const int32_t* __restrict__ kv_info, bf16_t* __restrict__ out,
                                                        int64_t ld_out, int B, int seq, int nh, int nkv, int d,
                                                        float scale_log2e, int causal, int out_cols, float* __restrict__ lse) {
    constexpr int RB = DP * 2;                 // tile row bytes
    constexpr int T_BYTES = 64 * RB;           // one tile: 64 keys
    constexpr int STAGE = 2 * T_BYTES;         // K tile + V tile
    constexpr int NI = DP / 32;                // glds instructions per wave per tile
    constexpr int DK = DP / 16;                // QK^T k-steps
    constexpr int DT = DP / 32;                // O^T row tiles
    constexpr int NBUF = DP == 128 ? 2 : 3;    // K/V tiles in LDS: the one being read + NBUF-1 in flight (96 KiB would
                                               // leave one block per CU at dp = 128)
    __shared__ __attribute__((aligned(16))) char smem[NBUF * STAGE];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware block order (1-D grid; consecutive ids go round-robin over the 8 XCDs): the q-tiles of one (batch, head)
    // re-read the same K / V, so they must share an L2 -- heads are dealt to the XCDs in groups of eight and an XCD walks
    // the q-tiles of its head back to back.  (With q-tile as the fastest grid index every q-tile of a head landed on a
    // different XCD and K / V crossed the fabric eight times: 1.07 GB per launch measured against 0.25 GB of operands.)
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
    const int h = hb % nh, b = hb / nh;
    const int hk = h / (nh / nkv);
    const int q0 = qt * 128 + w * 32;
    const int lq = lane & 31, hh = lane >> 5;
    const int query = q0 + lq;

    int end = kv_info[b];
    const int prefix = kv_info[B + b];
    if (causal) end = min(end, qt * 128 + 128);
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
    // the Q loads are the only vector-memory operations the compiler tracks: waited for here, ahead of the loop, so that
    // no vmcnt wait of its own lands inside a step (there it would also wait for the untracked DMA of the next tile);
    // passing the fragments through an empty asm makes the compiler place that wait here
#pragma unroll
    for (int kk = 0; kk < DK; ++kk) asm volatile("" : "+v"(qf[kk]));

    // ---- staging (LDS DMA, 1 KiB per wave instruction); K and V tiles share the layout ----
    // Issued from inline asm (saddr form: uniform tile base + 32-bit lane offset, M0 = LDS address) rather than through
    // the builtin: while a builtin LDS DMA is pending LLVM's waitcnt pass turns every wait it inserts into a wait for
    // zero on BOTH counters -- each K fragment read was waited for individually and a vmcnt(0) landed in the middle of
    // the step, i.e. the "prefetch" of the next tile was waited for before the softmax.  The explicit vmcnt(0) at the
    // top of each step is what orders the DMA against the reads.
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
            // rows past the end of the sequence re-read its last row (their scores are masked)
            const uint32_t vo = kb + s_row[i] < seq ? s_voff[i] : (uint32_t)((seq - 1 - kb) * RB) + (s_voff[i] - (uint32_t)(s_row[i] * RB));
            const uint32_t lk = sb + (w + 4 * i) * 1024, lv = lk + T_BYTES;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(kt), "s"(lk) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(vt), "s"(lv) : "memory");
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
    f32x16 negm = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // L2S: -m_run on every register (0 while m_run = -inf)

    // One 64-key step on LDS buffer BUF (a compile-time constant: every ds_read address is then a per-lane VGPR
    // plus an immediate, no per-read address arithmetic on the VALU, which is the busier pipe in this kernel).
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto step = [&](auto bufc, int it) {
        constexpr int BUF = decltype(bufc)::value;
        // tile `it` has landed (this wave's pieces: counted wait, with three buffers the next tile's 2 NI pieces may stay in
        // flight; all waves': the barrier), and every wave is done reading the buffer the new tile goes into (it held tile it-1)
        if (NBUF == 3 && it + 1 < n_it) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * NI));
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        if (it + NBUF - 1 < n_it) stage((BUF + NBUF - 1) % NBUF, it + NBUF - 1);
        const char* sb = smem + BUF * STAGE;
        const int kb = it * 64;

        // S^T tiles: rows = keys (permuted), cols = queries.  L2S: the accumulator starts at -m_run (0 before the first visible key)
        f32x16 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sb + k_off[t][kk]);
                st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], kk == 0 ? (L2S ? negm : zero16) : st[t], 0, 0, 0);
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
        // Lazy rescale: the reference point m_run of the exponentials only moves when some query of this wave saw a score
        // more than kSlack (in log2 units of the scaled scores) above it; otherwise p = exp2(scale (s - m_run)) <= 2^kSlack,
        // harmless in fp32 / bf16, and the 32 accumulator multiplies + the alpha exponential are skipped (wave-uniform
        // branch).  The result is unchanged up to rounding: numerator and denominator carry the same reference.
        constexpr float kSlack = 8.0f;
        float lsum = 0.f;
        if constexpr (L2S) {
            // st and mloc are RELATIVE to m_run (absolute while m_run = -inf: negm = 0), in log2 units
            const bool first = m_run == -INFINITY;
            const bool grow = mloc != -INFINITY && (first || mloc > kSlack);
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {
                const float delta = mloc == -INFINITY ? 0.f : (first ? mloc : fmaxf(mloc, 0.f));     // m_new - (first ? 0 : m_run)
                const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);                    // first: l_run = 0 and ot = 0 anyway
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[t][r] -= delta;
                if (mloc != -INFINITY || !first) m_run = (first ? 0.f : m_run) + delta;
                const float nm = m_run == -INFINITY ? 0.f : -m_run;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[r] = nm;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(st[t][r]);                              // exp(scale * (s - m))
                    st[t][r] = p;
                    lsum += p;
                }
        } else {
            const bool grow = mloc != -INFINITY && (m_run == -INFINITY || (mloc - m_run) * scale_log2e > kSlack);
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {
                const float m_new = fmaxf(m_run, mloc);                  // running reference of the RAW scores
                const float m_use = m_new == -INFINITY ? 0.f : m_new;   // fully masked so far: p = exp2(-inf) = 0
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * scale_log2e);   // m_run = -inf -> 0
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
                m_run = m_new;
            }
            const float mc = (m_run == -INFINITY ? 0.f : m_run) * scale_log2e;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(st[t][r], scale_log2e, -mc));   // exp(scale * (s - m))
                    st[t][r] = p;
                    lsum += p;
                }
        }
        lsum += __shfl_xor(lsum, 32, 64);
        l_run += lsum;
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
    if (NBUF == 3 && n_it > 1) stage(1, 1);
    for (int it = 0; it < n_it; it += NBUF) {
        step(std::integral_constant<int, 0>{}, it);
        if (it + 1 < n_it) step(std::integral_constant<int, 1>{}, it + 1);
        if constexpr (NBUF == 3) {
            if (it + 2 < n_it) step(std::integral_constant<int, 2>{}, it + 2);
        }
    }

    // ---- epilogue: O^T rows = channels (r&3) + 8(r>>2) + 4hh, col = query ----
    if (query < seq) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        // natural-log sum-exp of the effective logits (kernels.h attention()): the exponent reference is m_run in log2 units of the
        // stored scores (L2S) or in raw-score units (scale_log2e converts); both lanes of a query hold the same value
        if (lse && hh == 0)
            lse[(int64_t)(b * nh + h) * seq + query] = l_run > 0.f ? kLn2 * ((L2S ? m_run : m_run * scale_log2e) + __log2f(l_run)) : INFINITY;
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
                     int log2_scores, float* lse, hipStream_t s) {
    P2T_REQUIRE(d % 4 == 0 && (dp == 32 || dp == 64 || dp == 128) && d <= dp && nh % nkv == 0 && (nh * d) % 4 == 0 && ld_out % 4 == 0,
                "attention(mfma): unsupported shape d=%d dp=%d heads %d/%d", d, dp, nh, nkv);
    const dim3 grid((unsigned)(ceil_div(T, 128) * nh * B));
    const int out_cols = (int)(round_up((int64_t)nh * d, 64) < ld_out ? round_up((int64_t)nh * d, 64) : ld_out);
    const float sl = scale * 1.4426950408889634f;
#define P2T_ATTN(DPV)                                                                                                        \
    if (log2_scores)                                                                                                         \
        attn_mfma_kernel<DPV, true><<<grid, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_mask, kv_info, \
                                                         (bf16_t*)out, ld_out, B, T, nh, nkv, d, sl, causal, out_cols, lse); \
    else                                                                                                                     \
        attn_mfma_kernel<DPV, false><<<grid, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_mask, kv_info, \
                                                          (bf16_t*)out, ld_out, B, T, nh, nkv, d, sl, causal, out_cols, lse)
    if (dp == 32) { P2T_ATTN(32); }
    else if (dp == 64) { P2T_ATTN(64); }
    else { P2T_ATTN(128); }
#undef P2T_ATTN
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

}  // namespace p2t
