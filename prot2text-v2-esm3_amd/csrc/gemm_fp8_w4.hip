// fp8 (OCP e4m3) MFMA GEMM, four-wave persistent form of the 256 x 256 tile (gemm_w4.hip's structure on the block-scaled
// instruction of gemm_fp8.hip):  C[M,N] = (A8 . 2^ea)[M,K] * (W8 . 2^ew)[N,K]^T, fp32 accumulate, fused epilogue.
//
// A K step of 128 e4m3 bytes is the same 64 KiB of operands and the same 2 048 matrix-pipe cycles (64 x
// v_mfma_scale_f32_16x16x128_f8f6f4 per wave) as a 64-deep bf16 stage, so the LDS image (128-byte rows = whole cache lines per
// DMA piece, chunk swizzle c ^ ((r >> 1) & 7)), the two-buffer ring, the persistent tile stream and the hand-ordered volatile
// asm K loop carry over.  What differs: an operand fragment is 8 registers and both halves are needed by ONE instruction, so the
// 8 + 8 fragments (128 registers) cannot be double-buffered -- they are reloaded IN PLACE for step s+1 as soon as step s has
// issued their last MFMA.  The 64 MFMAs of a step run in two phases so that this happens early:
//   phase A: W fragments 0..3 x all activation fragments  ->  W 0..3 are free: reloaded at the start of phase B;
//   phase B: W fragments 4..7 x all activation fragments, j-major  ->  activation fragment j is free after MFMA (7, j);
//   W 4..7 are free at the end of the step: reloaded at the start of the next step's phase A, which does not use them.
// Buffer protocol (B = s & 1): once every wave has its W 4..7 of step s (start of the step, barrier 1) buffer B is idle and the
// DMA of step s+2 starts into it, 16 pieces spread over the rest of the step; step s+1 must have landed before the first
// reload at the start of phase B (counted vmcnt + barrier 2).  Fragment reads are ordinary loads (the compiler places the
// counted lgkmcnt waits in front of the asm MFMAs that consume them; the LDS DMA is asm, so it never widens those waits).
// Shapes: M % 256 == 0, N % 256 == 0 == n_cover, K % 128 == 0, K >= 512 (callers fall back to gemm_fp8.hip otherwise).
#include <type_traits>

#include "common.h"
#include "epilogue.h"
#include "gemm_tile_common.h"
#include "kernels.h"

#ifndef P2T_F8_EPI_P
#define P2T_F8_EPI_P 4      // row groups of a read-modify-write operand in flight ahead of the stores (rotary operands: two rows)
#endif

namespace p2t {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

// DIAG (lab build only; 0 in the product): 1 = s_memtime stamps around the two barriers of every step, summed per workgroup into
// ep.z ([loop cycles, barrier-1 cycles, barrier-2 cycles, steps, loop realtime ticks, epilogue cycles, cycles between epilogue and
// the next K loop, cycles of the first steps] as uint64); 2 = no DMA in the steady state
// (garbage results, timing only).  Measured with them (profiles/r04_fp8w4_diag_*.log): a step takes ~2 680 cycles for 2 048 of matrix
// work; the two barrier waits are ~80 cycles each, removing the 16 DMA pieces is worth 23-27 % -- the issue cost of the pieces, not
// the latency of their data, is what the loop pays.  A wave-staggered issue order (slot m of a step belongs to wave (m - 8) & 3: one
// piece per MFMA slot CU-wide, selected by scalar branches) was bit-identical and 7-19 % SLOWER: a taken branch per slot costs a
// one-wave SIMD more than the queueing it avoids.
template <typename Epi, int DIAG = 0>
__global__ void __launch_bounds__(256)
    gemm_nt_fp8_w4_kernel(const uint8_t* __restrict__ A, int64_t lda, const uint8_t* __restrict__ a_scale, const uint8_t* __restrict__ W,
                          int64_t ldw, const uint8_t* __restrict__ w_scale, int64_t M, int N, int K, int tiles_m, int tiles_n, int n_items,
                          EpiParams ep) {
    constexpr int MT = 8, NT = 8, BUF = 512 * 128, WOFF = 256 * 128, NL = 16;
    constexpr int kIssuedBeforeWait = 10;           // DMA pieces of step s+2 issued when the wave waits for step s+1 (schedule below)
    constexpr int kEpiOps = 2 * MT * Epi::kMinOps;
    constexpr int kExtCount = kIssuedBeforeWait + kEpiOps > 63 ? 63 : kIssuedBeforeWait + kEpiOps;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int nk = K >> 7;

    int item = blockIdx.x;
    int tm, tn;
    tile_coords(item, n_items, tiles_m, tiles_n, tm, tn);
    int64_t m0 = (int64_t)tm * 256;
    int n0 = tn * 256;

    // ---- staging (as gemm_w4.hip, byte addressed): wave w owns rows [64 w, 64 w + 64) of both operand tiles ----
    const char* a_ptr = (const char*)A + m0 * lda;
    const char* w_ptr = (const char*)W + (int64_t)n0 * ldw;
    uint32_t a_voff[8], w_voff[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int R = w * 64 + t * 8 + (lane >> 3), r = R & 63;
        const int c = (lane & 7) ^ ((R >> 1) & 7);
        const int wr = (R & ~63) + ((r >> 5) & 1) * 32 + ((r >> 2) & 3) * 8 + ((r >> 4) & 1) * 4 + (r & 3);      // permuted weight row
        a_voff[t] = (uint32_t)((int64_t)R * lda + c * 16);
        w_voff[t] = (uint32_t)((int64_t)wr * ldw + c * 16);
    }
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

    uint64_t d_loop = 0, d_b1 = 0, d_b2 = 0, d_steps = 0, d_real = 0, d_epi = 0, d_gap = 0, d_first = 0, t_epi_end = 0;
    auto now = [&]() -> uint64_t {
        uint64_t t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };
    auto now_real = [&]() -> uint64_t {
        uint64_t t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };

    const int fr = lane & 15, kg = lane >> 4;
    const uint32_t f_off = fr * 128 + ((kg ^ ((fr >> 1) & 7)) << 4);        // K bytes 16 g ..; K bytes 64 + 16 g .. at the same address ^ 64
    const uint32_t x_off = wm * 128 * 128 + f_off, w_off = WOFF + wn * 128 * 128 + f_off;

    // row scales of a tile (E8M0 bytes, one per operand row): byte (j & 3) of sx[j >> 2] belongs to activation fragment j of this
    // lane's row fr, byte (i & 3) of sw[i >> 2] to W fragment i (the instruction's op_sel picks the byte)
    // (fetch and combine are separate: the bytes of the NEXT tile are requested before the current tile's last two steps and
    // only combined behind its epilogue, so their latency is never waited for in the K loop)
    auto fetch_scales = [&](int64_t tm0, int tn0, uint8_t (&raw)[16]) {
#pragma unroll
        for (int j = 0; j < MT; ++j) raw[j] = a_scale[tm0 + wm * 128 + j * 16 + fr];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int r = (i & 3) * 16 + fr;
            const int nl = ((r >> 5) & 1) * 32 + ((r >> 2) & 3) * 8 + ((r >> 4) & 1) * 4 + (r & 3);
            raw[8 + i] = w_scale[tn0 + wn * 128 + (i >> 2) * 64 + nl];
        }
    };
    auto combine_scales = [&](const uint8_t (&raw)[16], int (&sx)[2], int (&sw)[2]) {
        sx[0] = sx[1] = sw[0] = sw[1] = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sx[j >> 2] |= (int)raw[j] << (8 * (j & 3));
            sw[j >> 2] |= (int)raw[8 + j] << (8 * (j & 3));
        }
    };

    auto frag = [&](int buf, uint32_t off) -> v8i {
        const v4i lo = *reinterpret_cast<const v4i*>(smem + buf * BUF + off);
        const v4i hi = *reinterpret_cast<const v4i*>(smem + buf * BUF + (off ^ 64u));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    using T = std::true_type;
    using F = std::false_type;
    // MFMA (i, j): W fragment i (operand A, scale byte i & 3 of swv), activation fragment j (operand B, byte j & 3 of sxv)
    auto mm = [&](auto first, auto ic, auto jc, f32x4& c, const v8i& a, const v8i& b, int swv, int sxv) {
        constexpr int I = decltype(ic)::value, J = decltype(jc)::value;
        if constexpr (decltype(first)::value)
            asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel:[%5,%6,0] op_sel_hi:[%7,%8,0]"
                         : "=a"(c) : "v"(a), "v"(b), "v"(swv), "v"(sxv), "n"(I & 1), "n"(J & 1), "n"((I >> 1) & 1), "n"((J >> 1) & 1));
        else
            asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[%5,%6,0] op_sel_hi:[%7,%8,0]"
                         : "+a"(c) : "v"(a), "v"(b), "v"(swv), "v"(sxv), "n"(I & 1), "n"(J & 1), "n"((I >> 1) & 1), "n"((J >> 1) & 1));
    };
    // LDS DMA in the saddr form (see gemm_mfma.hip); piece 0..7: activation rows, 8..15: weight rows
    auto dma1 = [&](int piece, int buf) {
        const char* sb = piece < 8 ? a_ptr : w_ptr;
        const uint32_t vo = piece < 8 ? a_voff[piece & 7] : w_voff[piece & 7];
        const uint32_t lds = lds0 + buf * BUF + (piece < 8 ? 0 : WOFF) + (w * 8 + (piece & 7)) * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(sb), "s"(lds) : "memory");
    };

    v8i X[MT], Wf[NT];
    int sx[2], sw[2];

    // One K step s (buffer B = s & 1): the activation fragments and W 0..3 of the step are in registers.  FIRST: the tile's
    // accumulators start here (C = 0).  RT: the DMA of step s+2 is conditional (`more`: last two steps of a tile).
    auto step = [&](f32x4 (&acc)[2][4][MT], auto bufc, auto first, auto rt, bool ext, bool more, auto last) {
        constexpr int B = decltype(bufc)::value;
        constexpr bool LAST = decltype(last)::value;
        constexpr bool RT = decltype(rt)::value;
        using FI = decltype(first);
        auto piece = [&](int q) { if (DIAG != 2 && (!RT || more)) dma1(q, B); };
        // MFMA m of the step: phase A (m < 32): i = m & 3, j = m >> 2; phase B: i = 4 + (m & 3), j = (m - 32) >> 2
#define P2T_F8_MM(Mi)                                                                                                              \
        mm(FI{}, std::integral_constant<int, ((Mi) < 32 ? ((Mi) & 3) : 4 + ((Mi) & 3))>{},            \
           std::integral_constant<int, ((Mi) < 32 ? ((Mi) >> 2) : (((Mi) - 32) >> 2))>{},                                                       \
           acc[((Mi) < 32 ? 0 : 1)][(Mi) & 3][((Mi) < 32 ? ((Mi) >> 2) : (((Mi) - 32) >> 2))], Wf[((Mi) < 32 ? ((Mi) & 3) : 4 + ((Mi) & 3))],      \
           X[((Mi) < 32 ? ((Mi) >> 2) : (((Mi) - 32) >> 2))], sw[((Mi) < 32 ? 0 : 1)], sx[((Mi) < 32 ? ((Mi) >> 2) : (((Mi) - 32) >> 2)) >> 2]);
#define P2T_F8_SB __builtin_amdgcn_sched_barrier(0);
#define P2T_F8_NL(stmt) if constexpr (!LAST) { stmt }      // the last step of a tile leaves the next tile's fragments to load_first_fragments()
        // ---- phase A: W 4..7 of this step arrive (their registers were free since the previous step's last MFMA) ----
        Wf[4] = frag(B, w_off + 4 * 2048); P2T_F8_SB P2T_F8_MM(0) P2T_F8_SB
        Wf[5] = frag(B, w_off + 5 * 2048); P2T_F8_SB P2T_F8_MM(1) P2T_F8_SB
        Wf[6] = frag(B, w_off + 6 * 2048); P2T_F8_SB P2T_F8_MM(2) P2T_F8_SB
        Wf[7] = frag(B, w_off + 7 * 2048); P2T_F8_SB P2T_F8_MM(3) P2T_F8_SB
        P2T_F8_MM(4) P2T_F8_MM(5) P2T_F8_MM(6) P2T_F8_MM(7)
        // every wave's reads of buffer B are done: it may be refilled (barrier 1)
        if constexpr (DIAG == 1) {
            const uint64_t t0 = now();
            asm volatile("s_barrier" ::: "memory");
            d_b1 += now() - t0;
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        piece(0); P2T_F8_MM(8) P2T_F8_MM(9) piece(1); P2T_F8_MM(10) P2T_F8_MM(11) P2T_F8_MM(12)
        piece(2); P2T_F8_MM(13) P2T_F8_MM(14) piece(3); P2T_F8_MM(15) P2T_F8_MM(16) P2T_F8_MM(17)
        piece(4); P2T_F8_MM(18) P2T_F8_MM(19) piece(5); P2T_F8_MM(20) P2T_F8_MM(21) P2T_F8_MM(22)
        piece(6); P2T_F8_MM(23) P2T_F8_MM(24) piece(7); P2T_F8_MM(25) P2T_F8_MM(26) P2T_F8_MM(27)
        piece(8); P2T_F8_MM(28) P2T_F8_MM(29) piece(9); P2T_F8_MM(30) P2T_F8_MM(31)
        // step s+1 has landed in every wave (barrier 2); `ext`: the previous tile's epilogue operations sit between it and this step's DMA
        uint64_t tb2 = 0;
        if constexpr (DIAG == 1) tb2 = now();
        if (RT && !more) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else if (FI::value && ext) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kExtCount) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kIssuedBeforeWait) : "memory");
        if constexpr (DIAG == 1) { d_b2 += now() - tb2; d_steps += 1; }
        // ---- phase B: W 0..3 of step s+1 first, then activation fragment j of step s+1 behind MFMA (7, j) ----
        P2T_F8_SB P2T_F8_NL(Wf[0] = frag(B ^ 1, w_off + 0 * 2048);) P2T_F8_SB P2T_F8_MM(32) P2T_F8_SB
        piece(10); P2T_F8_NL(Wf[1] = frag(B ^ 1, w_off + 1 * 2048);) P2T_F8_SB P2T_F8_MM(33) P2T_F8_SB
        P2T_F8_NL(Wf[2] = frag(B ^ 1, w_off + 2 * 2048);) P2T_F8_SB P2T_F8_MM(34) P2T_F8_SB
        P2T_F8_NL(Wf[3] = frag(B ^ 1, w_off + 3 * 2048);) P2T_F8_SB P2T_F8_MM(35) P2T_F8_SB
        P2T_F8_NL(X[0] = frag(B ^ 1, x_off + 0 * 2048);) P2T_F8_SB
        piece(11); P2T_F8_MM(36) P2T_F8_MM(37) P2T_F8_MM(38) P2T_F8_MM(39) P2T_F8_SB
        P2T_F8_NL(X[1] = frag(B ^ 1, x_off + 1 * 2048);) P2T_F8_SB
        piece(12); P2T_F8_MM(40) P2T_F8_MM(41) P2T_F8_MM(42) P2T_F8_MM(43) P2T_F8_SB
        P2T_F8_NL(X[2] = frag(B ^ 1, x_off + 2 * 2048);) P2T_F8_SB
        piece(13); P2T_F8_MM(44) P2T_F8_MM(45) piece(14); P2T_F8_MM(46) P2T_F8_MM(47) P2T_F8_SB
        P2T_F8_NL(X[3] = frag(B ^ 1, x_off + 3 * 2048);) P2T_F8_SB
        piece(15); P2T_F8_MM(48) P2T_F8_MM(49) P2T_F8_MM(50) P2T_F8_MM(51) P2T_F8_SB
        P2T_F8_NL(X[4] = frag(B ^ 1, x_off + 4 * 2048);) P2T_F8_SB
        P2T_F8_MM(52) P2T_F8_MM(53) P2T_F8_MM(54) P2T_F8_MM(55) P2T_F8_SB
        P2T_F8_NL(X[5] = frag(B ^ 1, x_off + 5 * 2048);) P2T_F8_SB
        P2T_F8_MM(56) P2T_F8_MM(57) P2T_F8_MM(58) P2T_F8_MM(59) P2T_F8_SB
        P2T_F8_NL(X[6] = frag(B ^ 1, x_off + 6 * 2048);) P2T_F8_SB
        P2T_F8_MM(60) P2T_F8_MM(61) P2T_F8_MM(62) P2T_F8_MM(63) P2T_F8_SB
        P2T_F8_NL(X[7] = frag(B ^ 1, x_off + 7 * 2048);) P2T_F8_SB
#undef P2T_F8_MM
#undef P2T_F8_SB
#undef P2T_F8_NL
        a_ptr += 128; w_ptr += 128;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    if constexpr (DIAG == 4 || DIAG == 5) {          // lab: start-phase skew of half of the XCDs (4) / of every other CU (5) by half a tile
        const bool late = DIAG == 4 ? (blockIdx.x & 4) != 0 : ((blockIdx.x >> 3) & 1) != 0;
        if (late) {
            const uint64_t t0 = now();
            const uint64_t wait_cycles = (uint64_t)nk * 1340;
            while (now() - t0 < wait_cycles) __builtin_amdgcn_s_sleep(32);
        }
    }
    // ---- prologue: steps 0 and 1 of the first tile, its scales, its first fragments ----
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int q = 0; q < NL; ++q) dma1(q, b);
        a_ptr += 128; w_ptr += 128;
    }
    {
        uint8_t raw[16];
        fetch_scales(m0, n0, raw);
        combine_scales(raw, sx, sw);
    }
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NL) : "memory");
    // W fragments 0..3 and all activation fragments of a tile's step 0 (buffer 0).  Also run BEHIND every epilogue: held across it,
    // these 96 registers leave the epilogue too few (spills whose reloads drain its stores; no room to prefetch its operands)
    auto load_first_fragments = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) Wf[i] = frag(0, w_off + i * 2048);
#pragma unroll
        for (int j = 0; j < MT; ++j) X[j] = frag(0, x_off + j * 2048);
    };
    load_first_fragments();

    f32x4 acc[2][4][MT];                // [W fragments 0..3 / 4..7][W fragment & 3][activation fragment]
    bool ext = false;
    for (;;) {
        const int nxt = item + (int)gridDim.x;
        const bool has_next = nxt < n_items;
        uint64_t tl0 = 0, tr0 = 0;
        if constexpr (DIAG == 1) { tl0 = now(); tr0 = now_real(); if (ext) d_gap += tl0 - t_epi_end; }
        step(acc, I0{}, T{}, F{}, ext, true, F{});
        if constexpr (DIAG == 1) d_first += now() - tl0;
        step(acc, I1{}, F{}, F{}, false, true, F{});
        for (int s = 2; s + 2 < nk; s += 2) {          // nk is even (K % 256 == 0) -- see the launcher
            step(acc, I0{}, F{}, F{}, false, true, F{});
            step(acc, I1{}, F{}, F{}, false, true, F{});
        }
        int64_t nm0 = 0;
        int nn0 = 0;
        uint8_t nraw[16] = {0};
        if (has_next) {                                 // the last two steps issue the next tile's first two; its scales are fetched now
            tile_coords(nxt, n_items, tiles_m, tiles_n, tm, tn);
            nm0 = (int64_t)tm * 256;
            nn0 = tn * 256;
            a_ptr = (const char*)A + nm0 * lda;
            w_ptr = (const char*)W + (int64_t)nn0 * ldw;
            fetch_scales(nm0, nn0, nraw);
        }
        step(acc, I0{}, F{}, T{}, false, has_next, F{});
        step(acc, I1{}, F{}, T{}, false, has_next, T{});     // the next tile's first fragments are read behind the epilogue
        uint64_t te0 = 0;
        if constexpr (DIAG == 1) { te0 = now(); d_loop += te0 - tl0; d_real += now_real() - tr0; }
        // the last MFMAs retire before the epilogue reads accumulators: the compiler tracks no hazards across inline asm, and it
        // would hoist the epilogue's v_accvgpr_read above a bare s_nop -- so every quad is re-defined (no code) BEHIND the wait
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) asm volatile("" : "+a"(acc[h][i][j]));
        // the next tile's scale bytes are combined BEFORE the first store of this tile: behind the stores, the wait for these sixteen
        // loads (or for their spill slots) would drain every store of the epilogue -- vmcnt retires in order
        int nsx[2] = {0, 0}, nsw[2] = {0, 0};
        if (has_next) combine_scales(nraw, nsx, nsw);
        {
            int fr_e = fr, kg_e = kg;
            asm volatile("" : "+v"(fr_e), "+v"(kg_e), "+v"(nsx[0]), "+v"(nsx[1]), "+v"(nsw[0]), "+v"(nsw[1]));
            tile_epilogue_pair<MT, Epi, (epi_has_fetch<Epi>::value && !Epi::kRmw) ? 2 : P2T_F8_EPI_P>(acc, ep, m0, n0, wm, wn, fr_e, kg_e);
        }
        if constexpr (DIAG == 1) {
            t_epi_end = now();
            d_epi += t_epi_end - te0;
            if (!has_next && threadIdx.x == 0) {
                uint64_t* d = (uint64_t*)ep.z + (size_t)blockIdx.x * 8;
                d[0] = d_loop; d[1] = d_b1; d[2] = d_b2; d[3] = d_steps; d[4] = d_real; d[5] = d_epi; d[6] = d_gap; d[7] = d_first;
            }
        }
        if (!has_next) break;
        ext = true;
        item = nxt;
        m0 = nm0;
        n0 = nn0;
        sx[0] = nsx[0]; sx[1] = nsx[1]; sw[0] = nsw[0]; sw[1] = nsw[1];
        load_first_fragments();
    }
}

// Persistent four-wave fp8 GEMM.  P2T_ERR_UNSUPPORTED when the shape is not eligible (the caller falls back to gemm_fp8.hip).
template <typename Epi>
int launch_gemm_fp8_w4(const void* A, int64_t lda, const uint8_t* a_scale, const void* W, int64_t ldw, const uint8_t* w_scale, int64_t M, int N,
                       int K, int n_cover, int grid, const EpiParams& ep, hipStream_t s) {
    const int64_t items = (M / 256) * (N / 256);
    if (M % 256 || N % 256 || n_cover != N || K % 256 || K < 512 || items < grid || (int64_t)256 * (lda > ldw ? lda : ldw) >= ((int64_t)1 << 32))
        return P2T_ERR_UNSUPPORTED;
    gemm_nt_fp8_w4_kernel<Epi><<<dim3((unsigned)grid), 256, 0, s>>>((const uint8_t*)A, lda, a_scale, (const uint8_t*)W, ldw, w_scale, M, N, K,
                                                                  (int)(M / 256), N / 256, (int)items, ep);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}
#ifdef P2T_LAB
// lab build: the diagnostic forms (plain bf16 store epilogue only; ep.z = uint64[8 * grid] for DIAG 1)
int launch_gemm_fp8_w4_diag(int diag, const void* A, int64_t lda, const uint8_t* a_scale, const void* W, int64_t ldw, const uint8_t* w_scale, int64_t M,
                            int N, int K, int n_cover, int grid, const EpiParams& ep, hipStream_t s) {
    const int64_t items = (M / 256) * (N / 256);
    if (M % 256 || N % 256 || n_cover != N || K % 256 || K < 512 || items < grid) return P2T_ERR_UNSUPPORTED;
    using E = EpiStore<bf16_t>;
    if (diag == 4)
        gemm_nt_fp8_w4_kernel<E, 4><<<dim3((unsigned)grid), 256, 0, s>>>((const uint8_t*)A, lda, a_scale, (const uint8_t*)W, ldw, w_scale, M, N, K, (int)(M / 256), N / 256, (int)items, ep);
    else if (diag == 5)
        gemm_nt_fp8_w4_kernel<E, 5><<<dim3((unsigned)grid), 256, 0, s>>>((const uint8_t*)A, lda, a_scale, (const uint8_t*)W, ldw, w_scale, M, N, K, (int)(M / 256), N / 256, (int)items, ep);
    else if (diag == 1)
        gemm_nt_fp8_w4_kernel<E, 1><<<dim3((unsigned)grid), 256, 0, s>>>((const uint8_t*)A, lda, a_scale, (const uint8_t*)W, ldw, w_scale, M, N, K, (int)(M / 256), N / 256, (int)items, ep);
    else
        gemm_nt_fp8_w4_kernel<E, 2><<<dim3((unsigned)grid), 256, 0, s>>>((const uint8_t*)A, lda, a_scale, (const uint8_t*)W, ldw, w_scale, M, N, K, (int)(M / 256), N / 256, (int)items, ep);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}
#endif
#define P2T_F8W4_INST(E) template int launch_gemm_fp8_w4<E>(const void*, int64_t, const uint8_t*, const void*, int64_t, const uint8_t*, int64_t, int, int, int, int, const EpiParams&, hipStream_t);
P2T_F8W4_INST(EpiStore<bf16_t>)
P2T_F8W4_INST(EpiResid)
P2T_F8W4_INST(EpiSwiglu<bf16_t>)
P2T_F8W4_INST(EpiQkvRope<bf16_t>)
P2T_F8W4_INST(EpiGeluFp8)
#undef P2T_F8W4_INST

}  // namespace p2t
