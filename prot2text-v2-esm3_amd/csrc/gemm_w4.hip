// bf16 MFMA GEMM, four-wave form of the 256 x 256 tile:  C[M,N] = A[M,K] * W[N,K]^T, fp32 accumulate, fused epilogue.
//
// Same LDS image, staging, swizzle, W-row permutation, tile order and epilogue functors as gemm_mfma.hip; what differs
// is the wave layout: 4 wavefronts as 2 (M) x 2 (N), each owning 128 x 128 outputs = 64 accumulator quads (256
// registers, the AGPR half of a 512-register wave), one wave per SIMD.  Per 32-deep stage a wave reads 8 + 8 fragments
// for 64 MFMAs (the 2 x 4 layout of eight waves reads 8 + 4 for 32): a third fewer LDS fragment bytes per flop, which
// is LDS-port time and power the matrix pipe gets back.  With a single wave per SIMD
// nothing hides a wave's own stalls, so every memory operation is placed by hand between MFMA pairs: the 16 fragment
// reads of stage s+1 (register double buffer) behind the first 16 pairs of stage s, the 8 LDS-DMA pieces of stage s+4
// behind the next 8, and the last 8 pairs cover the tail of the LDS latency before the step's only wait.
#include <type_traits>

#include "common.h"
#include "epilogue.h"
#include "gemm_tile_common.h"
#include "kernels.h"

namespace p2t {
#ifdef P2T_LAB
__device__ uint64_t* g_lab_stamp_ptr = nullptr;          // lab build: see p2t_lab_set_stamp_buffer below
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// PERSIST: one block per CU walks the items blockIdx.x, + gridDim.x, ... < n_items of the tile order over n_items + n_tail tiles and
// treats the K loops of consecutive tiles as ONE stream of stages: the last two stages of a tile issue the DMA of the next
// tile's first two, and the epilogue's stores are never drained -- vmcnt completes in issue order, so the first stage after
// an epilogue waits with a count that leaves the epilogue's operations in flight (Epi::kMinOps is a lower bound of them: the
// persistent form only takes shapes without edge tiles, where every wave issues all of them).  Requires M % 256 == 0,
// N % 256 == 0 == n_cover, K % 128 == 0, K >= 256.  !PERSIST: one tile per block, any M / N (rows clamped, edge epilogue).
template <typename Epi, bool PERSIST, int SCHED = 1, bool PAIRS_ONLY = false>
__global__ void __launch_bounds__(256)
    gemm_nt_w4_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw, int64_t M, int N, int K,
                      int tiles_m, int tiles_n, int n_items, int n_tail, int n_cover, EpiParams ep, SplitFix fix) {
    constexpr int MT = 8, NT = 8, BUF = 512 * 128, WOFF = 256 * 128, NL = 16;
#ifndef P2T_LAB
    static_assert(SCHED == 1, "the product library carries one instruction order of the K loop; order 0 is in the lab build");
#endif
    constexpr int kEpiOps = 2 * MT * Epi::kMinOps;
    constexpr int kIssuedBeforeWait = SCHED == 0 ? 6 : 9;    // DMA pieces of stage s+2 a wave has issued when it waits for stage s+1 (tools/gen_w4_schedule.py)
    constexpr int kExtCount = kIssuedBeforeWait + kEpiOps > 63 ? 63 : kIssuedBeforeWait + kEpiOps;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int nk = K >> 6;

    int item = blockIdx.x;
    int tm, tn;
    const int n_order = n_items + n_tail;
    tile_coords(item, n_order, tiles_m, tiles_n, tm, tn);
    int64_t m0 = (int64_t)tm * 256;
    int n0 = tn * 256;

    // ---- staging: one DMA instruction = 8 rows x 128 B (whole cache lines); wave w owns rows [64 w, 64 w + 64) of both
    // operand tiles = 8 + 8 pieces per 64-deep stage.  LDS image: 128-byte rows, 16-byte chunk c of row r at chunk
    // c ^ ((r >> 1) & 7) (conflict-free for the fragment reads below); the image is lane-linear, so the swizzle is applied to
    // the SOURCE address. ----
    const char* a_ptr = (const char*)(A + m0 * lda);          // DMA source of the next stage to issue (advances 128 B per stage)
    const char* w_ptr = (const char*)(W + (int64_t)n0 * ldw);
    uint32_t a_voff[8], w_voff[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int R = w * 64 + t * 8 + (lane >> 3), r = R & 63;
        const int c = (lane & 7) ^ ((R >> 1) & 7);
        int64_t ar = R;
        int wr = (R & ~63) + ((r >> 5) & 1) * 32 + ((r >> 2) & 3) * 8 + ((r >> 4) & 1) * 4 + (r & 3);      // permuted weight row
        if (!PERSIST) {
            ar = m0 + R < M ? R : (int)(M - 1 - m0);
            wr = n0 + wr < N ? wr : N - 1 - n0;
        }
        a_voff[t] = (uint32_t)(ar * lda + c * 8) * 2u;
        w_voff[t] = (uint32_t)((int64_t)wr * ldw + c * 8) * 2u;
    }
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int fr = lane & 15, kg = lane >> 4;
    const uint32_t f_off = fr * 128 + ((kg ^ ((fr >> 1) & 7)) << 4);        // K half 0; K half 1 is the same address ^ 64
    const uint32_t x_addr = lds0 + wm * 128 * 128 + f_off, w_addr = lds0 + WOFF + wn * 128 * 128 + f_off;

    bf16x8 xa[MT], wa[NT], xb[MT], wb[NT];           // fragments of K half 0 / K half 1 of the current stage

    // The whole K loop is volatile inline asm, so the issue order below IS the program order: the compiler allocates
    // registers and forms addresses, nothing else ("a" pins every accumulator quad to AGPRs, in place; left to itself the
    // allocator shuffles accumulators between the register files with v_accvgpr copies and s_nop bubbles).
    auto rd = [&](bf16x8& f, uint32_t addr, auto off) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(decltype(off)::value));
    };
    auto mm = [&](auto first, f32x4& c, const bf16x8& a, const bf16x8& b) {
        if constexpr (decltype(first)::value) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    };
    // LDS DMA in the saddr form (see gemm_mfma.hip: the builtin makes LLVM turn counted waits into waits for zero)
    auto dma1 = [&](int piece, int buf) {                               // piece 0..7: activation rows, 8..15: weight rows
        const char* sb = piece < 8 ? a_ptr : w_ptr;
        const uint32_t vo = piece < 8 ? a_voff[piece & 7] : w_voff[piece & 7];
        const uint32_t lds = lds0 + buf * BUF + (piece < 8 ? 0 : WOFF) + (w * 8 + (piece & 7)) * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(sb), "s"(lds) : "memory");
    };
    auto mm_dma = [&](auto first, f32x4& c, const bf16x8& a, const bf16x8& b, int piece, int buf) {
        const char* sb = piece < 8 ? a_ptr : w_ptr;
        const uint32_t vo = piece < 8 ? a_voff[piece & 7] : w_voff[piece & 7];
        const uint32_t lds = lds0 + buf * BUF + (piece < 8 ? 0 : WOFF) + (w * 8 + (piece & 7)) * 1024;
        if constexpr (decltype(first)::value)
            asm volatile("s_mov_b32 m0, %5\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\tglobal_load_lds_dwordx4 %3, %4"
                         : "=a"(c) : "v"(a), "v"(b), "v"(vo), "s"(sb), "s"(lds) : "memory");
        else
            asm volatile("s_mov_b32 m0, %5\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tglobal_load_lds_dwordx4 %3, %4"
                         : "+a"(c) : "v"(a), "v"(b), "v"(vo), "s"(sb), "s"(lds) : "memory");
    };
    using T = std::true_type;
    using F = std::false_type;
    // One 64-deep stage s (buffer B = s & 1); its K-half-0 fragments are in (xa, wa).
    //   first half : 64 MFMAs on K half 0 (FIRST: C = 0, the tile's accumulators start here); the 16 reads of K half 1
    //                behind the first 16 pairs; once they have landed in every wave (lgkmcnt(0) + barrier) buffer B is free
    //                and the DMA of stage s+2 starts into it (RT: only if `more`);
    //   second half: 64 MFMAs on K half 1; stage s+1 has landed (vmcnt + barrier; it was issued a whole stage ago; `ext`:
    //                the previous tile's epilogue operations sit between it and this stage's DMA), its K-half-0 fragments
    //                are read into (xa, wa) behind pairs 4..19.
    auto stage = [&](f32x4 (&acc)[2][4][MT], auto bufc, auto first, auto rt, bool ext, bool more) {
        constexpr int B = decltype(bufc)::value;
        constexpr bool RT = decltype(rt)::value;
        using FI = decltype(first);
        const uint32_t xs1 = (x_addr + B * BUF) ^ 64u, ws1 = (w_addr + B * BUF) ^ 64u;                 // K half 1 of this stage
        const uint32_t xs0 = x_addr + (B ^ 1) * BUF, ws0 = w_addr + (B ^ 1) * BUF;                     // K half 0 of the next one
#define P2T_W4_PAIR(FF, WF, XF, P)                                                                            \
        mm(FF{}, acc[(((P) & 3) * 2) >> 2][(((P) & 3) * 2) & 3][(P) >> 2], WF[((P) & 3) * 2], XF[(P) >> 2]);   \
        mm(FF{}, acc[(((P) & 3) * 2 + 1) >> 2][(((P) & 3) * 2 + 1) & 3][(P) >> 2], WF[((P) & 3) * 2 + 1], XF[(P) >> 2]);
#define P2T_W4_R1W(I) rd(wb[I], ws1, std::integral_constant<int, (I) * 2048>{});
#define P2T_W4_R1X(J) rd(xb[J], xs1, std::integral_constant<int, (J) * 2048>{});
#define P2T_W4_R0W(I) rd(wa[I], ws0, std::integral_constant<int, (I) * 2048>{});
#define P2T_W4_R0X(J) rd(xa[J], xs0, std::integral_constant<int, (J) * 2048>{});
#define P2T_W4_G(Q) if (!RT || more) dma1(Q, B);
// a pair that carries DMA piece Q: M0 is set before the pair's first MFMA and the DMA goes behind it (the MFMA is the wait
// state the M0 write needs: no s_nop); stages whose DMA is conditional (RT) keep the separate form
#define P2T_W4_GPAIR(Q, FF, WF, XF, P)                                                                                           \
        if constexpr (!RT) {                                                                                                     \
            mm_dma(FF{}, acc[(((P) & 3) * 2) >> 2][(((P) & 3) * 2) & 3][(P) >> 2], WF[((P) & 3) * 2], XF[(P) >> 2], Q, B);       \
            mm(FF{}, acc[(((P) & 3) * 2 + 1) >> 2][(((P) & 3) * 2 + 1) & 3][(P) >> 2], WF[((P) & 3) * 2 + 1], XF[(P) >> 2]);     \
        } else {                                                                                                                 \
            P2T_W4_G(Q) P2T_W4_PAIR(FF, WF, XF, P)                                                                               \
        }
#define P2T_W4_WAIT_NEXT_STAGE                                                                                                  \
        if (RT && !more) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                                          \
        else if (FI::value && ext) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kExtCount) : "memory");                 \
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kIssuedBeforeWait) : "memory");
        // GENERATED (tools/gen_w4_schedule.py) BEGIN
        if constexpr (SCHED == 1) {
            P2T_W4_R1W(0) P2T_W4_R1W(1) P2T_W4_PAIR(FI, wa, xa, 0) P2T_W4_R1X(0) P2T_W4_R1W(2) P2T_W4_PAIR(FI, wa, xa, 1) P2T_W4_R1W(3) P2T_W4_R1X(1) P2T_W4_PAIR(FI, wa, xa, 2) P2T_W4_R1W(4) P2T_W4_R1W(5) P2T_W4_PAIR(FI, wa, xa, 3)
            P2T_W4_R1X(2) P2T_W4_R1W(6) P2T_W4_PAIR(FI, wa, xa, 4) P2T_W4_R1W(7) P2T_W4_R1X(3) P2T_W4_PAIR(FI, wa, xa, 5) P2T_W4_R1X(4) P2T_W4_R1X(5) P2T_W4_PAIR(FI, wa, xa, 6) P2T_W4_R1X(6) P2T_W4_R1X(7) P2T_W4_PAIR(FI, wa, xa, 7)
            P2T_W4_PAIR(FI, wa, xa, 8) P2T_W4_PAIR(FI, wa, xa, 9) P2T_W4_PAIR(FI, wa, xa, 10)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            P2T_W4_GPAIR(0, FI, wa, xa, 11)
            P2T_W4_PAIR(FI, wa, xa, 12) P2T_W4_PAIR(FI, wa, xa, 13) P2T_W4_GPAIR(1, FI, wa, xa, 14) P2T_W4_PAIR(FI, wa, xa, 15)
            P2T_W4_PAIR(FI, wa, xa, 16) P2T_W4_GPAIR(2, FI, wa, xa, 17) P2T_W4_PAIR(FI, wa, xa, 18) P2T_W4_PAIR(FI, wa, xa, 19)
            P2T_W4_GPAIR(3, FI, wa, xa, 20) P2T_W4_PAIR(FI, wa, xa, 21) P2T_W4_PAIR(FI, wa, xa, 22) P2T_W4_GPAIR(4, FI, wa, xa, 23)
            P2T_W4_PAIR(FI, wa, xa, 24) P2T_W4_PAIR(FI, wa, xa, 25) P2T_W4_GPAIR(5, FI, wa, xa, 26) P2T_W4_PAIR(FI, wa, xa, 27)
            P2T_W4_PAIR(FI, wa, xa, 28) P2T_W4_GPAIR(6, FI, wa, xa, 29) P2T_W4_PAIR(FI, wa, xa, 30) P2T_W4_PAIR(FI, wa, xa, 31)
            P2T_W4_GPAIR(7, F, wb, xb, 0) P2T_W4_PAIR(F, wb, xb, 1) P2T_W4_PAIR(F, wb, xb, 2) P2T_W4_GPAIR(8, F, wb, xb, 3)
            P2T_W4_WAIT_NEXT_STAGE
            P2T_W4_R0W(0) P2T_W4_PAIR(F, wb, xb, 4) P2T_W4_R0W(1) P2T_W4_PAIR(F, wb, xb, 5) P2T_W4_R0X(0) P2T_W4_GPAIR(9, F, wb, xb, 6) P2T_W4_R0W(2) P2T_W4_PAIR(F, wb, xb, 7)
            P2T_W4_R0W(3) P2T_W4_PAIR(F, wb, xb, 8) P2T_W4_R0X(1) P2T_W4_GPAIR(10, F, wb, xb, 9) P2T_W4_R0W(4) P2T_W4_PAIR(F, wb, xb, 10) P2T_W4_R0W(5) P2T_W4_PAIR(F, wb, xb, 11)
            P2T_W4_R0X(2) P2T_W4_GPAIR(11, F, wb, xb, 12) P2T_W4_R0W(6) P2T_W4_PAIR(F, wb, xb, 13) P2T_W4_R0W(7) P2T_W4_PAIR(F, wb, xb, 14) P2T_W4_R0X(3) P2T_W4_GPAIR(12, F, wb, xb, 15)
            P2T_W4_R0X(4) P2T_W4_PAIR(F, wb, xb, 16) P2T_W4_R0X(5) P2T_W4_PAIR(F, wb, xb, 17) P2T_W4_R0X(6) P2T_W4_GPAIR(13, F, wb, xb, 18) P2T_W4_R0X(7) P2T_W4_PAIR(F, wb, xb, 19)
            P2T_W4_PAIR(F, wb, xb, 20) P2T_W4_GPAIR(14, F, wb, xb, 21) P2T_W4_PAIR(F, wb, xb, 22) P2T_W4_PAIR(F, wb, xb, 23)
            P2T_W4_GPAIR(15, F, wb, xb, 24) P2T_W4_PAIR(F, wb, xb, 25) P2T_W4_PAIR(F, wb, xb, 26) P2T_W4_PAIR(F, wb, xb, 27)
            P2T_W4_PAIR(F, wb, xb, 28) P2T_W4_PAIR(F, wb, xb, 29) P2T_W4_PAIR(F, wb, xb, 30) P2T_W4_PAIR(F, wb, xb, 31)
        }
#ifdef P2T_LAB
        else if constexpr (SCHED == 0) {
            P2T_W4_R1W(0) P2T_W4_PAIR(FI, wa, xa, 0) P2T_W4_R1W(1) P2T_W4_PAIR(FI, wa, xa, 1) P2T_W4_R1X(0) P2T_W4_PAIR(FI, wa, xa, 2) P2T_W4_R1W(2) P2T_W4_PAIR(FI, wa, xa, 3)
            P2T_W4_R1W(3) P2T_W4_PAIR(FI, wa, xa, 4) P2T_W4_R1X(1) P2T_W4_PAIR(FI, wa, xa, 5) P2T_W4_R1W(4) P2T_W4_PAIR(FI, wa, xa, 6) P2T_W4_R1W(5) P2T_W4_PAIR(FI, wa, xa, 7)
            P2T_W4_R1X(2) P2T_W4_PAIR(FI, wa, xa, 8) P2T_W4_R1W(6) P2T_W4_PAIR(FI, wa, xa, 9) P2T_W4_R1W(7) P2T_W4_PAIR(FI, wa, xa, 10) P2T_W4_R1X(3) P2T_W4_PAIR(FI, wa, xa, 11)
            P2T_W4_R1X(4) P2T_W4_PAIR(FI, wa, xa, 12) P2T_W4_R1X(5) P2T_W4_PAIR(FI, wa, xa, 13) P2T_W4_R1X(6) P2T_W4_PAIR(FI, wa, xa, 14) P2T_W4_R1X(7) P2T_W4_PAIR(FI, wa, xa, 15)
            P2T_W4_PAIR(FI, wa, xa, 16) P2T_W4_PAIR(FI, wa, xa, 17) P2T_W4_PAIR(FI, wa, xa, 18) P2T_W4_PAIR(FI, wa, xa, 19)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            P2T_W4_GPAIR(0, FI, wa, xa, 20) P2T_W4_PAIR(FI, wa, xa, 21) P2T_W4_GPAIR(1, FI, wa, xa, 22) P2T_W4_PAIR(FI, wa, xa, 23)
            P2T_W4_PAIR(FI, wa, xa, 24) P2T_W4_GPAIR(2, FI, wa, xa, 25) P2T_W4_PAIR(FI, wa, xa, 26) P2T_W4_PAIR(FI, wa, xa, 27)
            P2T_W4_GPAIR(3, FI, wa, xa, 28) P2T_W4_PAIR(FI, wa, xa, 29) P2T_W4_PAIR(FI, wa, xa, 30) P2T_W4_GPAIR(4, FI, wa, xa, 31)
            P2T_W4_PAIR(F, wb, xb, 0) P2T_W4_GPAIR(5, F, wb, xb, 1) P2T_W4_PAIR(F, wb, xb, 2) P2T_W4_PAIR(F, wb, xb, 3)
            P2T_W4_WAIT_NEXT_STAGE
            P2T_W4_R0W(0) P2T_W4_GPAIR(6, F, wb, xb, 4) P2T_W4_R0W(1) P2T_W4_PAIR(F, wb, xb, 5) P2T_W4_R0X(0) P2T_W4_PAIR(F, wb, xb, 6) P2T_W4_R0W(2) P2T_W4_GPAIR(7, F, wb, xb, 7)
            P2T_W4_R0W(3) P2T_W4_PAIR(F, wb, xb, 8) P2T_W4_R0X(1) P2T_W4_PAIR(F, wb, xb, 9) P2T_W4_R0W(4) P2T_W4_GPAIR(8, F, wb, xb, 10) P2T_W4_R0W(5) P2T_W4_PAIR(F, wb, xb, 11)
            P2T_W4_R0X(2) P2T_W4_PAIR(F, wb, xb, 12) P2T_W4_R0W(6) P2T_W4_GPAIR(9, F, wb, xb, 13) P2T_W4_R0W(7) P2T_W4_PAIR(F, wb, xb, 14) P2T_W4_R0X(3) P2T_W4_PAIR(F, wb, xb, 15)
            P2T_W4_R0X(4) P2T_W4_GPAIR(10, F, wb, xb, 16) P2T_W4_R0X(5) P2T_W4_PAIR(F, wb, xb, 17) P2T_W4_R0X(6) P2T_W4_PAIR(F, wb, xb, 18) P2T_W4_R0X(7) P2T_W4_GPAIR(11, F, wb, xb, 19)
            P2T_W4_PAIR(F, wb, xb, 20) P2T_W4_PAIR(F, wb, xb, 21) P2T_W4_GPAIR(12, F, wb, xb, 22) P2T_W4_PAIR(F, wb, xb, 23)
            P2T_W4_PAIR(F, wb, xb, 24) P2T_W4_GPAIR(13, F, wb, xb, 25) P2T_W4_PAIR(F, wb, xb, 26) P2T_W4_PAIR(F, wb, xb, 27)
            P2T_W4_GPAIR(14, F, wb, xb, 28) P2T_W4_PAIR(F, wb, xb, 29) P2T_W4_PAIR(F, wb, xb, 30) P2T_W4_GPAIR(15, F, wb, xb, 31)
        }
#endif
        // GENERATED END
        a_ptr += 128; w_ptr += 128;
#undef P2T_W4_WAIT_NEXT_STAGE
#undef P2T_W4_PAIR
#undef P2T_W4_R1W
#undef P2T_W4_R1X
#undef P2T_W4_R0W
#undef P2T_W4_R0X
#undef P2T_W4_G
#undef P2T_W4_GPAIR
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the next stage's first fragments are in registers
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ring fill (stages 0 and 1 of the job a_ptr / w_ptr point at) + the first fragments
    auto prologue = [&]() {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int q = 0; q < NL; ++q) dma1(q, b);
        a_ptr += 128; w_ptr += 128;
    }
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NL) : "memory");
    rd(wa[0], w_addr, std::integral_constant<int, 0>{}); rd(wa[1], w_addr, std::integral_constant<int, 2048>{});
    rd(wa[2], w_addr, std::integral_constant<int, 4096>{}); rd(wa[3], w_addr, std::integral_constant<int, 6144>{});
    rd(wa[4], w_addr, std::integral_constant<int, 8192>{}); rd(wa[5], w_addr, std::integral_constant<int, 10240>{});
    rd(wa[6], w_addr, std::integral_constant<int, 12288>{}); rd(wa[7], w_addr, std::integral_constant<int, 14336>{});
    rd(xa[0], x_addr, std::integral_constant<int, 0>{}); rd(xa[1], x_addr, std::integral_constant<int, 2048>{});
    rd(xa[2], x_addr, std::integral_constant<int, 4096>{}); rd(xa[3], x_addr, std::integral_constant<int, 6144>{});
    rd(xa[4], x_addr, std::integral_constant<int, 8192>{}); rd(xa[5], x_addr, std::integral_constant<int, 10240>{});
    rd(xa[6], x_addr, std::integral_constant<int, 12288>{}); rd(xa[7], x_addr, std::integral_constant<int, 14336>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // Whole tiles: items blockIdx.x, + gridDim.x, ... < n_items.
    auto k_loop = [&](f32x4 (&acc)[2][4][MT], int nk_l, bool ext, bool has_next, int64_t nm0, int nn0) {
        stage(acc, I0{}, T{}, F{}, ext, true);
        stage(acc, I1{}, F{}, F{}, false, true);
        for (int s = 2; s + 2 < nk_l; s += 2) {        // nk_l is even
            stage(acc, I0{}, F{}, F{}, false, true);
            stage(acc, I1{}, F{}, F{}, false, true);
        }
        if (has_next) {                                 // the last two stages issue the next tile's first two
            a_ptr = (const char*)(A + nm0 * lda);
            w_ptr = (const char*)(W + (int64_t)nn0 * ldw);
        }
        stage(acc, I0{}, F{}, T{}, false, has_next);
        stage(acc, I1{}, F{}, T{}, false, has_next);   // (without a next tile its fragment reads fetch stale LDS: unused)
        // the last MFMAs retire before the epilogue reads accumulators: the compiler tracks no hazards across inline asm, and it
        // would hoist the epilogue's v_accvgpr_read above a bare s_nop -- so every quad is re-defined (no code) BEHIND the wait
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) asm volatile("" : "+a"(acc[h][i][j]));
    };
    // PAIRS_ONLY (a grid of at most half a round of tiles with a long K: o-proj / FFN-down of the text tower): no whole tiles,
    // every tile runs as a split-K pair below (n_items == 0, grid = 2 n_tail)
    if constexpr (!PAIRS_ONLY) {
    prologue();
    {
        f32x4 acc[2][4][MT];            // [64-column group of the wave][W fragment][activation fragment]
        bool ext = false;
#ifdef P2T_LAB
        // lab build: with ep.z set on an epilogue that does not use it, s_memtime stamps around the K loop and the epilogue of every
        // tile, summed per workgroup into ep.z as uint64 [K-loop cycles, epilogue cycles, epilogue end -> next K loop, tiles, loop
        // realtime ticks] (tools/w4_diag.py)
        constexpr bool kDiagEpi = std::is_same<Epi, EpiStore<bf16_t>>::value || std::is_same<Epi, EpiGelu<bf16_t, false>>::value || std::is_same<Epi, EpiResid>::value;
        uint64_t* dbgp = kDiagEpi ? (uint64_t*)ep.z : nullptr;                  // stamps go to ep.z where the epilogue never touches it,
        if (!dbgp) dbgp = g_lab_stamp_ptr;                                      // else to the pointer of p2t_lab_set_stamp_buffer (any epilogue)
        const bool diag = PERSIST && dbgp != nullptr;
        uint64_t d_loop = 0, d_epi = 0, d_gap = 0, d_tiles = 0, d_real = 0, t_end = 0;
        auto now = [&]() -> uint64_t { uint64_t t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; };
        auto now_real = [&]() -> uint64_t { uint64_t t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; };
#endif
        for (;;) {
            const int nxt = item + (int)gridDim.x;
            const bool has_next = PERSIST && nxt < n_items;
            int64_t nm0 = 0;
            int nn0 = 0;
            if (has_next) {
                tile_coords(nxt, n_order, tiles_m, tiles_n, tm, tn);
                nm0 = (int64_t)tm * 256;
                nn0 = tn * 256;
            }
#ifdef P2T_LAB
            uint64_t t0 = 0, r0 = 0, t1 = 0;
            if (diag) { t0 = now(); r0 = now_real(); if (ext) d_gap += t0 - t_end; }
#endif
            k_loop(acc, nk, ext, has_next, nm0, nn0);
#ifdef P2T_LAB
            if (diag) { t1 = now(); d_loop += t1 - t0; d_real += now_real() - r0; d_tiles += 1; }
#endif
            if (PERSIST) {
                // launder the lane coordinates so the per-row output addresses are rebuilt per tile instead of being held across the K loop
                int fr_e = fr, kg_e = kg;
                asm volatile("" : "+v"(fr_e), "+v"(kg_e));
                tile_epilogue_pair<MT, Epi, (epi_has_fetch<Epi>::value && !Epi::kRmw) ? 2 : 4>(acc, ep, m0, n0, wm, wn, fr_e, kg_e);      // rotary operands: two rows ahead (16 registers each; four cost spills)
            } else {
                const bool interior = m0 + 256 <= M && n0 + 256 <= N && n0 + 256 <= n_cover;
                if (interior) {
                    tile_epilogue<MT, Epi, true>(acc[0], ep, M, N, n_cover, m0, n0, wm, 2 * wn, fr, kg);
                    tile_epilogue<MT, Epi, true>(acc[1], ep, M, N, n_cover, m0, n0, wm, 2 * wn + 1, fr, kg);
                } else {
                    tile_epilogue<MT, Epi, false>(acc[0], ep, M, N, n_cover, m0, n0, wm, 2 * wn, fr, kg);
                    tile_epilogue<MT, Epi, false>(acc[1], ep, M, N, n_cover, m0, n0, wm, 2 * wn + 1, fr, kg);
                }
            }
#ifdef P2T_LAB
            if (diag) {
                t_end = now();
                d_epi += t_end - t1;
                if (!has_next && threadIdx.x == 0) {
                    uint64_t* d = dbgp + (size_t)blockIdx.x * 8;
                    d[0] = d_loop; d[1] = d_epi; d[2] = d_gap; d[3] = d_tiles; d[4] = d_real;
                }
            }
#endif
            if (!has_next) break;
            ext = true;                                 // every wave issued at least kEpiOps operations in that epilogue
            item = nxt;
            m0 = nm0;
            n0 = nn0;
        }
    }
    }
    // The partial last round: blocks [0, 2 n_tail) take one half of tile n_items + t each -- the SECOND K half as producer
    // (blocks [0, n_tail): publishes the raw accumulators, never waits) or the FIRST K half as consumer (blocks [n_tail,
    // 2 n_tail): adds the partner's slab on the way into the epilogue).  Every block of the grid is resident (one per CU), so a
    // consumer's partner is always running.  Own accumulator variables: the tile loop's register allocation is untouched.
    if (PERSIST && (int)blockIdx.x < 2 * n_tail) {
        const bool producer = (int)blockIdx.x < n_tail;
        const int t = producer ? (int)blockIdx.x : (int)blockIdx.x - n_tail;
        const int nk0 = (nk >> 1) & ~1;                 // consumer: stages [0, nk0), producer: [nk0, nk)
        tile_coords(n_items + t, n_order, tiles_m, tiles_n, tm, tn);
        m0 = (int64_t)tm * 256;
        n0 = tn * 256;
        a_ptr = (const char*)(A + m0 * lda + (producer ? nk0 * 64 : 0));
        w_ptr = (const char*)(W + (int64_t)n0 * ldw + (producer ? nk0 * 64 : 0));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the ring is idle in every wave
        prologue();
        f32x4 acc2[2][4][MT];
        k_loop(acc2, producer ? nk - nk0 : nk0, false, false, 0, 0);
        if (producer) {
            // register order: quad q = (h * 4 + i) * MT + j of thread t at slab[q * 256 + t]; drain, then release at agent scope
            float4* slab = reinterpret_cast<float4*>(fix.slab) + (int64_t)t * (64 * 256) + threadIdx.x;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        slab[((h * 4 + i) * MT + j) * 256] = make_float4(acc2[h][i][j][0], acc2[h][i][j][1], acc2[h][i][j][2], acc2[h][i][j][3]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (w == 0) {               // wave-uniform (every lane stores the same word): no divergent region around the hand-off
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(fix.flag + t, fix.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            // wait for the partner (bounded spin; giving up is LOUD: sticky time-out word + NaN tiles, see gemm_mfma.hip).  The
            // whole wait is ONE asm statement executed by every wave (agent-scope load of the flag, scalar compare, sleep,
            // then the acquire's cache invalidate): no control flow the compiler can see, so the 256 accumulators stay where
            // the K loop left them -- with a C++ spin loop the register allocator moved them across the loop and a quad came
            // back wrong (tests/test_gpu_kernels.py::test_gemm_persistent_forms_fuzz[8])
            {
                const unsigned* fp = fix.flag + t;
                unsigned seen, spins, vtmp;
                const unsigned zero = 0;
                asm volatile(
                    "s_mov_b32 %1, 0\n"
                    "1:\n\t"
                    "global_load_dword %2, %5, %3 sc1\n\t"
                    "s_waitcnt vmcnt(0)\n\t"
                    "v_readfirstlane_b32 %0, %2\n\t"
                    "s_cmp_eq_u32 %0, %4\n\t"
                    "s_cbranch_scc1 2f\n\t"
                    "s_sleep 8\n\t"
                    "s_add_u32 %1, %1, 1\n\t"
                    "s_cmp_lt_u32 %1, 0x400000\n\t"
                    "s_cbranch_scc1 1b\n"
                    "2:\n\t"
                    "buffer_inv sc1"
                    : "=&s"(seen), "=&s"(spins), "=&v"(vtmp)
                    : "s"(fp), "s"(fix.epoch), "v"(zero)
                    : "memory", "scc");
                if (seen != fix.epoch) __hip_atomic_store(fix.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const bool timed_out = __hip_atomic_load(fix.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            const float poison = timed_out ? __builtin_nanf("") : 0.f;
            const float4* slab = reinterpret_cast<const float4*>(fix.slab) + (int64_t)t * (64 * 256) + threadIdx.x;
            tile_epilogue<MT, Epi, true, true>(acc2[0], ep, M, N, n_cover, m0, n0, wm, 2 * wn, fr, kg, slab, 256, poison);
            tile_epilogue<MT, Epi, true, true>(acc2[1], ep, M, N, n_cover, m0, n0, wm, 2 * wn + 1, fr, kg, slab + 4 * MT * 256, 256, poison);
        }
    }
}

template <typename Epi>
static int launch_w4(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover, const EpiParams& ep,
                     hipStream_t s) {
    const int tiles_m = (int)ceil_div(M, 256), tiles_n = (int)ceil_div(n_cover, 256);
    gemm_nt_w4_kernel<Epi, false><<<dim3((unsigned)(tiles_m * tiles_n)), 256, 0, s>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, tiles_m,
                                                                                     tiles_n, tiles_m * tiles_n, 0, n_cover, ep, SplitFix{});
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

// Persistent form: n_items whole tiles + n_tail tiles (the partial last round, n_tail <= grid / 2) as split-K pairs, `fix` the
// hand-off workspace (unused when n_tail == 0).  Caller guarantees: M % 256 == 0, N % 256 == 0 == n_cover, K % 128 == 0,
// K >= 256 (K >= 1024 with a tail), n_items >= grid when there is a tail.
template <typename Epi>
int launch_gemm_w4_persist(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_items, int n_tail, int grid,
                           const EpiParams& ep, const SplitFix& fix, hipStream_t s, int sched) {
    const dim3 g((unsigned)(grid < n_items ? grid : n_items));
#ifdef P2T_LAB
    if (sched == 0) {
        gemm_nt_w4_kernel<Epi, true, 0><<<g, 256, 0, s>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, (int)(M / 256), N / 256, n_items, n_tail, N, ep, fix);
        P2T_LAUNCH_CHECK();
        return P2T_OK;
    }
#endif
    P2T_REQUIRE(sched == 1, "gemm (four-wave): instruction order %d is not in this build", sched);
    gemm_nt_w4_kernel<Epi, true, 1><<<g, 256, 0, s>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, (int)(M / 256), N / 256, n_items, n_tail, N, ep, fix);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}
// Every tile as a split-K pair (no whole tiles): n_tail tiles, grid 2 n_tail <= CUs.  Caller guarantees the persistent form's
// shape conditions and K >= 1024.
template <typename Epi>
int launch_gemm_w4_pairs(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_tail, const EpiParams& ep,
                         const SplitFix& fix, hipStream_t s) {
    gemm_nt_w4_kernel<Epi, true, 1, true><<<dim3((unsigned)(2 * n_tail)), 256, 0, s>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, (int)(M / 256),
                                                                                      N / 256, 0, n_tail, N, ep, fix);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}
template int launch_gemm_w4_pairs<EpiResid>(const void*, int64_t, const void*, int64_t, int64_t, int, int, int, const EpiParams&, const SplitFix&, hipStream_t);
template int launch_gemm_w4_pairs<EpiStore<bf16_t>>(const void*, int64_t, const void*, int64_t, int64_t, int, int, int, const EpiParams&, const SplitFix&, hipStream_t);
template int launch_gemm_w4_pairs<EpiStore<float>>(const void*, int64_t, const void*, int64_t, int64_t, int, int, int, const EpiParams&, const SplitFix&, hipStream_t);

template int launch_gemm_w4_persist<EpiStore<bf16_t>>(const void*, int64_t, const void*, int64_t, int64_t, int, int, int, int, int, const EpiParams&, const SplitFix&, hipStream_t, int);
template int launch_gemm_w4_persist<EpiStore<float>>(const void*, int64_t, const void*, int64_t, int64_t, int, int, int, int, int, const EpiParams&, const SplitFix&, hipStream_t, int);
template int launch_gemm_w4_persist<EpiResid>(const void*, int64_t, const void*, int64_t, int64_t, int, int, int, int, int, const EpiParams&, const SplitFix&, hipStream_t, int);
template int launch_gemm_w4_persist<EpiQkvRope<bf16_t>>(const void*, int64_t, const void*, int64_t, int64_t, int, int, int, int, int, const EpiParams&, const SplitFix&, hipStream_t, int);
template int launch_gemm_w4_persist<EpiGelu<bf16_t>>(const void*, int64_t, const void*, int64_t, int64_t, int, int, int, int, int, const EpiParams&, const SplitFix&, hipStream_t, int);
template int launch_gemm_w4_persist<EpiSwiglu<bf16_t>>(const void*, int64_t, const void*, int64_t, int64_t, int, int, int, int, int, const EpiParams&, const SplitFix&, hipStream_t, int);

// Four-wave form for the epilogues it is built for; P2T_ERR_UNSUPPORTED otherwise (the caller falls back to gemm_mfma.hip).
int launch_gemm_w4(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover, int out_dtype, int epilogue,
                   const EpiParams& ep, hipStream_t s) {
    if (K % 128 != 0 || K < 256 || M < 1 || (int64_t)256 * (lda > ldw ? lda : ldw) * 2 >= (int64_t)1 << 32) return P2T_ERR_UNSUPPORTED;
    const bool ob = out_dtype == P2T_BF16;
    switch (epilogue) {
        case P2T_EPI_STORE: return ob ? launch_w4<EpiStore<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, s) : launch_w4<EpiStore<float>>(A, lda, W, ldw, M, N, K, n_cover, ep, s);
        case P2T_EPI_GELU: if (ob) return launch_w4<EpiGelu<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, s); break;
        case P2T_EPI_RESID: return launch_w4<EpiResid>(A, lda, W, ldw, M, N, K, n_cover, ep, s);
        case P2T_EPI_QKV_ROPE: if (ob) return launch_w4<EpiQkvRope<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, s); break;
        default: break;
    }
    return P2T_ERR_UNSUPPORTED;
}

}  // namespace p2t

#ifdef P2T_LAB
// lab build only (tools/w4_diag.py): where the bf16 four-wave kernel writes its per-workgroup stamps when the epilogue's own z pointer cannot
// carry them (QKV + RoPE, GELU, SwiGLU); nullptr switches the stamps off again.  uint64 [8 x workgroups].
extern "C" int p2t_lab_set_stamp_buffer(void* ptr) {
    uint64_t* p = (uint64_t*)ptr;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(p2t::g_lab_stamp_ptr), &p, sizeof(p));
}
#endif
