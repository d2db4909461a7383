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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int w4_vm_imm(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }
}

template <typename Epi>
__global__ void __launch_bounds__(256)
    gemm_nt_w4_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw, int64_t M, int N, int K,
                      int tiles_m, int tiles_n, int n_cover, EpiParams ep) {
    constexpr int MT = 8, NT = 8, SLOT = 512 * 64, NL = 8;
    __shared__ __attribute__((aligned(16))) char smem[4 * SLOT];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int ns = K >> 5;

    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
    const int64_t m0 = (int64_t)tm * 256;
    const int n0 = tn * 256;

    // ---- staging: wave w owns rows [64 w, 64 w + 64) of both operand tiles: 4 + 4 pieces of 16 rows x 64 B per stage ----
    const int schunk = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
    const char* a_base = (const char*)(A + m0 * lda);
    const char* w_base = (const char*)(W + (int64_t)n0 * ldw);
    uint32_t a_voff[4], w_voff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int R = w * 64 + t * 16 + (lane >> 2), r = R & 63;
        int64_t ar = m0 + R < M ? R : (int)(M - 1 - m0);
        int wr = (R & ~63) + ((r >> 5) & 1) * 32 + ((r >> 2) & 3) * 8 + ((r >> 4) & 1) * 4 + (r & 3);      // permuted weight row
        wr = n0 + wr < N ? wr : N - 1 - n0;
        a_voff[t] = (uint32_t)(ar * lda + schunk * 8) * 2u;
        w_voff[t] = (uint32_t)((int64_t)wr * ldw + schunk * 8) * 2u;
    }
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    // LDS DMA in the saddr form from inline asm (see gemm_mfma.hip: the builtin makes LLVM turn counted waits into waits for zero)
    auto dma = [&](int piece, int slot, int koff) {                  // piece 0..3: activation rows, 4..7: weight rows
        const char* sb = (piece < 4 ? a_base : w_base) + koff * 2;
        const uint32_t vo = piece < 4 ? a_voff[piece & 3] : w_voff[piece & 3];
        const uint32_t lds = lds0 + slot * SLOT + (piece < 4 ? 0 : 256 * 64) + (w * 4 + (piece & 3)) * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(sb), "s"(lds) : "memory");
    };

    const int fr = lane & 15, kg = lane >> 4;
    const int sw = (kg ^ ((4 - ((fr >> 2) & 3)) & 3)) << 4;
    const int x_off = (wm * 128 + fr) * 64 + sw;
    const int w_off = 256 * 64 + (wn * 128 + fr) * 64 + sw;

    f32x4 acc[2][4][MT];                // [64-column group of the wave][W fragment][activation fragment]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 xa[MT], wa[NT], xb[MT], wb[NT];

    // one 32-deep stage; FULL: stages s+1 .. s+4 all exist (branch-free)
    auto step = [&](auto full, int s, const bf16x8 (&xc)[MT], const bf16x8 (&wc)[NT], bf16x8 (&xn)[MT], bf16x8 (&wnx)[NT]) {
        constexpr bool FULL = decltype(full)::value;
        const bool rd = FULL || s + 1 < ns, is = FULL || s + 4 < ns;
        if (FULL) {
            __builtin_amdgcn_s_waitcnt(w4_vm_imm(2 * NL));              // stage s+1 landed; s+2, s+3 may be in flight
        } else if (s + 1 < ns) {
            const int inflight = (ns - 1 < s + 3 ? ns - 1 : s + 3) - (s + 1);
            if (inflight >= 2) __builtin_amdgcn_s_waitcnt(w4_vm_imm(2 * NL));
            else if (inflight == 1) __builtin_amdgcn_s_waitcnt(w4_vm_imm(NL));
            else __builtin_amdgcn_s_waitcnt(w4_vm_imm(0));
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* xs = smem + ((s + 1) & 3) * SLOT + x_off;
        const char* ws = smem + ((s + 1) & 3) * SLOT + w_off;
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            if (p < 8) {
                if (rd) wnx[p] = *reinterpret_cast<const bf16x8*>(ws + p * 1024);
            } else if (p < 16) {
                if (rd) xn[p - 8] = *reinterpret_cast<const bf16x8*>(xs + (p - 8) * 1024);
            } else if (p < 24) {
                if (is) dma(p - 16, s & 3, (s + 4) * 32);
            }
            const int j = p >> 2, i0 = (p & 3) * 2;
            acc[i0 >> 2][i0 & 3][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[i0], xc[j], acc[i0 >> 2][i0 & 3][j], 0, 0, 0);
            acc[(i0 + 1) >> 2][(i0 + 1) & 3][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[i0 + 1], xc[j], acc[(i0 + 1) >> 2][(i0 + 1) & 3][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                             // lgkmcnt(0): the next step's fragments are in registers
    };

#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < ns) {
#pragma unroll
            for (int q = 0; q < NL; ++q) dma(q, s, s * 32);
        }
    {
        const int inflight = ns - 1 < 3 ? ns - 1 : 3;
        if (inflight >= 3) __builtin_amdgcn_s_waitcnt(w4_vm_imm(3 * NL));
        else if (inflight == 2) __builtin_amdgcn_s_waitcnt(w4_vm_imm(2 * NL));
        else if (inflight == 1) __builtin_amdgcn_s_waitcnt(w4_vm_imm(NL));
        else __builtin_amdgcn_s_waitcnt(w4_vm_imm(0));
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < NT; ++i) wa[i] = *reinterpret_cast<const bf16x8*>(smem + w_off + i * 1024);
#pragma unroll
    for (int j = 0; j < MT; ++j) xa[j] = *reinterpret_cast<const bf16x8*>(smem + x_off + j * 1024);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    int s = 0;
    for (; s + 5 < ns; s += 2) {
        step(std::true_type{}, s, xa, wa, xb, wb);
        step(std::true_type{}, s + 1, xb, wb, xa, wa);
    }
    for (; s < ns; s += 2) {                                            // K % 64 == 0: an even number of stages
        step(std::false_type{}, s, xa, wa, xb, wb);
        step(std::false_type{}, s + 1, xb, wb, xa, wa);
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool interior = m0 + 256 <= M && n0 + 256 <= N && n0 + 256 <= n_cover;
    if (interior) {
        tile_epilogue<MT, Epi, true>(acc[0], ep, M, N, n_cover, m0, n0, wm, 2 * wn, fr, kg);
        tile_epilogue<MT, Epi, true>(acc[1], ep, M, N, n_cover, m0, n0, wm, 2 * wn + 1, fr, kg);
    } else {
        tile_epilogue<MT, Epi, false>(acc[0], ep, M, N, n_cover, m0, n0, wm, 2 * wn, fr, kg);
        tile_epilogue<MT, Epi, false>(acc[1], ep, M, N, n_cover, m0, n0, wm, 2 * wn + 1, fr, kg);
    }
}

template <typename Epi>
static int launch_w4(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover, const EpiParams& ep,
                     hipStream_t s) {
    const int tiles_m = (int)ceil_div(M, 256), tiles_n = (int)ceil_div(n_cover, 256);
    gemm_nt_w4_kernel<Epi><<<dim3((unsigned)(tiles_m * tiles_n)), 256, 0, s>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, tiles_m,
                                                                              tiles_n, n_cover, ep);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

// Four-wave form for the epilogues it is built for; P2T_ERR_UNSUPPORTED otherwise (the caller falls back to gemm_mfma.hip).
int launch_gemm_w4(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover, int out_dtype, int epilogue,
                   const EpiParams& ep, hipStream_t s) {
    if (K % 64 != 0 || M < 1 || (int64_t)256 * (lda > ldw ? lda : ldw) * 2 >= (int64_t)1 << 32) return P2T_ERR_UNSUPPORTED;
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
