// fp8 (OCP e4m3) MFMA GEMM for gfx950:  C[M,N] = (A8 . 2^ea)[M,K] * (W8 . 2^ew)[N,K]^T, fp32 accumulate, fused epilogue.
//
// Operands are e4m3 bytes with ONE power-of-two scale per row (E8M0 byte: a token's activations, an output channel's
// weights; quant.hip); the block-scaled matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4 multiplies at twice the bf16
// rate and applies both scales itself (its scale operands, constant along K here).  Lane maps of the instruction, found
// with one-hot operands on the hardware (tools/mx_probe2.hip, profiles/r02_mx_probe.log):
//   operand registers 0..3 of lane (r = lane & 15, g = lane >> 4) hold K bytes 16 g .. 16 g + 15 of row r, registers 4..7
//   hold K bytes 64 + 16 g .. 64 + 16 g + 15 -- two K = 64 halves, each laid out exactly like the bf16 16x16x32 operand
//   (16 bytes per lane, lane group g = k chunk g); the scale byte of lane 16 b + r applies to (row r, K block b of 32).
// So the staging scheme of the bf16 kernels carries over (LDS DMA, source-side bank swizzle, W-row permutation): the LDS image
// of a K step is (BM + 256) rows x 128 bytes (whole cache lines per DMA instruction, as in gemm_w4.hip), the low / high half of
// the operand registers are the two 64-byte halves of a row.  What differs is the register budget: an operand
// fragment is 8 registers, so 8 + 4 fragments (96) + 128 accumulators leave no room for double buffering; fragments are
// reloaded in place as soon as their last MFMA of the step has issued (W fragment i after row i of the 4 x 8 MFMA grid,
// activation fragment j after MFMA (3, j)) from the OTHER ring slot, and the ring is two K steps deep:
//   top of step s:  wait own DMAs (step s+1) + own LDS reads, barrier  ->  DMA of step s+2 into the slots of step s,
//   interleaved with the 32 MFMAs of step s and the 24 fragment reads of step s+1.
// Per-tile launch form (one block per tile, XCD-aware grouped tile order), epilogues shared with the bf16 kernel.
// A persistent form (K loops of consecutive tiles as one stream, undrained epilogue stores, row scales staged through LDS)
// was built and measured at +6 % .. -2 % of this kernel on the tower shapes (profiles/r02_microbench_fp8_persistent.log): the
// loop is bound by the global->LDS path (ablation: +26 .. 62 % without the DMA, profiles/r02_microbench_fp8abl.log), not by
// the tile boundaries, and with 96 fragment registers the tile loop's own state spills -- dropped.
// Requirements: K % 128 == 0 (producers pad K with zero bytes), lda / ldw % 16 == 0, 16-byte aligned bases.
#include <type_traits>

#include "common.h"
#include "epilogue.h"
#include "gemm_tile_common.h"
#include "kernels.h"

namespace p2t {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int OA, int OB>
__device__ __forceinline__ f32x4 mfma_fp8(const v8i& a, const v8i& b, const f32x4& c, int sa, int sb) {
    // cbsz = blgp = 0: both operands e4m3; op_sel picks the byte of the scale registers
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OA, sa, OB, sb);
}

typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
// the same skeleton on bf16 operands (experiment, p2t_set_gemm_policy(6)): a K step of 128 BYTES is 64 bf16 elements, the two
// 64-byte halves of a fragment feed two v_mfma_f32_16x16x32_bf16
__device__ __forceinline__ f32x4 mfma_bf16_pair(const v8i& a, const v8i& b, f32x4 c) {
    const v4i alo = __builtin_shufflevector(a, a, 0, 1, 2, 3), ahi = __builtin_shufflevector(a, a, 4, 5, 6, 7);
    const v4i blo = __builtin_shufflevector(b, b, 0, 1, 2, 3), bhi = __builtin_shufflevector(b, b, 4, 5, 6, 7);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, alo), __builtin_bit_cast(bf16x8v, blo), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, ahi), __builtin_bit_cast(bf16x8v, bhi), c, 0, 0, 0);
}

// ABL (lab only, results are garbage): 1 = no DMA in the K loop, 2 = no fragment reloads, 3 = neither -- what the loop costs
// without its global->LDS traffic / its LDS reads (tools/microbench.py fp8abl)
template <int MT, typename Epi, bool F8 = true, int ABL = 0>
__global__ void __launch_bounds__(512)
    gemm_nt_fp8_kernel(const uint8_t* __restrict__ A, int64_t lda, const uint8_t* __restrict__ a_scale, const uint8_t* __restrict__ W,
                       int64_t ldw, const uint8_t* __restrict__ w_scale, int64_t M, int N, int K, int tiles_m, int tiles_n, int n_cover,
                       EpiParams ep) {
    constexpr int WN = 4, NT = 4, BM = 2 * MT * 16;
    constexpr int AI = BM / 64;                             // activation DMA pieces per wave per K step (1 KiB each)
    constexpr int SLOT = (BM + 256) * 128;                  // bytes per K step (128 K bytes of every row)
    constexpr int NP = AI + 4;                              // DMA pieces per wave per K step
    __shared__ __attribute__((aligned(16))) char smem[2 * SLOT];

    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * 256;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w / WN, wn = w % WN;
    const int ns = K >> 7;                                  // K steps of 128

    // ---- staging: one wave instruction = 8 rows x 128 B -- WHOLE cache lines (half lines, 16 rows x 64 B, cost the loop a
    // second L2 request for every line: the loop is bound by this path, profiles/r02_microbench_fp8abl.log); lane -> (row =
    // lane >> 3, 16-byte chunk = lane & 7).  LDS image: 128-byte rows, chunk c of row r at chunk c ^ ((r >> 1) & 7)
    // (conflict-free fragment reads); the image is lane-linear, so the swizzle is applied to the SOURCE address. ----
    const char* a_base = (const char*)A + m0 * lda;         // uniform: tile origin; lane offsets below are the same for every step
    const char* w_base = (const char*)W + (int64_t)n0 * ldw;
    uint32_t a_voff[AI], w_voff[4];
#pragma unroll
    for (int t = 0; t < AI; ++t) {
        const int R = (w * AI + t) * 8 + (lane >> 3);
        int64_t am = m0 + R;
        am = (am < M ? am : M - 1) - m0;                    // rows past the edge re-read the last row (results discarded)
        a_voff[t] = (uint32_t)(am * lda + (((lane & 7) ^ ((R >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int R = (w * 4 + t) * 8 + (lane >> 3), r = R & 63;
        const int nl = ((r >> 5) & 1) * 32 + ((r >> 2) & 3) * 8 + ((r >> 4) & 1) * 4 + (r & 3);     // permuted weight row
        int wr = n0 + (R & ~63) + nl;
        wr = (wr < N ? wr : N - 1) - n0;
        w_voff[t] = (uint32_t)((int64_t)wr * ldw + (((lane & 7) ^ ((R >> 1) & 7)) << 4));
    }
    // LDS DMA as inline asm in the saddr form (see gemm_mfma.hip: the builtin makes LLVM drain every counted wait)
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto piece = [&](auto which, int s) {                   // DMA piece `which` (0 .. NP-1) of K step s: Q < AI activation, else weight
        constexpr int Q = decltype(which)::value;
        const char* sb = (Q < AI ? a_base : w_base) + (int64_t)s * 128;
        const uint32_t vo = Q < AI ? a_voff[Q < AI ? Q : 0] : w_voff[Q < AI ? 0 : Q - AI];
        const uint32_t lds = lds0 + (s & 1) * SLOT + (Q < AI ? (w * AI + Q) * 1024 : BM * 128 + (w * 4 + (Q - AI)) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(sb), "s"(lds) : "memory");
    };
    auto issue_all = [&](int s) {
        piece(std::integral_constant<int, 0>{}, s); piece(std::integral_constant<int, 1>{}, s);
        piece(std::integral_constant<int, 2>{}, s); piece(std::integral_constant<int, 3>{}, s);
        piece(std::integral_constant<int, 4>{}, s); piece(std::integral_constant<int, 5>{}, s);
        if constexpr (NP == 8) { piece(std::integral_constant<int, 6>{}, s); piece(std::integral_constant<int, 7>{}, s); }
    };

    // ---- fragment read offsets: K bytes 16 g .. of row fr (low registers) and 64 + 16 g .. (high registers) ----
    const int fr = lane & 15, kg = lane >> 4;
    const int sw = (kg ^ ((fr >> 1) & 7)) << 4;
    const int x_off = (wm * MT * 16 + fr) * 128 + sw;
    const int w_off = BM * 128 + (wn * NT * 16 + fr) * 128 + sw;
    constexpr int FS = 2048;                                // bytes between consecutive fragments (16 rows)
    auto frag = [&](int s, int off) -> v8i {
        const v4i lo = *reinterpret_cast<const v4i*>(smem + (s & 1) * SLOT + off);
        const v4i hi = *reinterpret_cast<const v4i*>(smem + (s & 1) * SLOT + (off ^ 64));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    // ---- prologue: DMA of K steps 0 and 1, then the row scales (E8M0 bytes, one per operand row, packed per fragment) ----
    issue_all(0);
    if (ns > 1) issue_all(1);
    int sx[MT / 4 > 0 ? MT / 4 : 1] = {0}, sw8 = 0;
#pragma unroll
    for (int j = 0; F8 && j < MT; ++j) {
        int64_t m = m0 + wm * MT * 16 + j * 16 + fr;
        m = m < M ? m : M - 1;
        const int e = a_scale[m];
        if ((j & 3) == 0) sx[j >> 2] = e; else sx[j >> 2] |= e << (8 * (j & 3));
    }
#pragma unroll
    for (int i = 0; F8 && i < NT; ++i) {
        const int r = i * 16 + fr;
        const int nl = ((r >> 5) & 1) * 32 + ((r >> 2) & 3) * 8 + ((r >> 4) & 1) * 4 + (r & 3);
        int n = n0 + wn * 64 + nl;
        n = n < N ? n : N - 1;
        sw8 |= (int)w_scale[n] << (8 * i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every prologue DMA (and the scale bytes) landed
    __builtin_amdgcn_s_barrier();

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    v8i X[MT], Wf[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) Wf[i] = frag(0, w_off + i * FS);
#pragma unroll
    for (int j = 0; j < MT; ++j) X[j] = frag(0, x_off + j * FS);

    // One K step.  MORE: a step s + 1 exists (reload the fragments); ISSUE: a step s + 2 exists (start its DMA).  Both are
    // compile-time so the steady-state body is branch-free; the last two steps are peeled below.
    auto step = [&](auto more_c, auto issue_c, int s) {
        constexpr bool MORE = decltype(more_c)::value && !(ABL & 2), ISSUE = decltype(issue_c)::value && !(ABL & 1);
        // own fragment reads of step s complete, own DMAs of step s+1 landed; after the barrier: every wave's are, so the
        // slots of step s may be overwritten and the slots of step s+1 may be read
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        auto row = [&](auto ii) {
            constexpr int I = decltype(ii)::value;
            auto two = [&](auto jj) {                       // MFMAs (I, J), (I, J + 1)
                constexpr int J = decltype(jj)::value;
                if constexpr (F8) {
                    acc[I][J] = mfma_fp8<I, J & 3>(Wf[I], X[J], acc[I][J], sw8, sx[J >> 2]);
                    acc[I][J + 1] = mfma_fp8<I, (J + 1) & 3>(Wf[I], X[J + 1], acc[I][J + 1], sw8, sx[(J + 1) >> 2]);
                } else {
                    acc[I][J] = mfma_bf16_pair(Wf[I], X[J], acc[I][J]);
                    acc[I][J + 1] = mfma_bf16_pair(Wf[I], X[J + 1], acc[I][J + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto dma2 = [&](auto pp) {                      // DMA piece pp of step s + 2
                constexpr int P = decltype(pp)::value;
                if constexpr (ISSUE && P < NP) piece(std::integral_constant<int, P>{}, s + 2);
            };
            auto reloadx = [&](auto jj) {
                constexpr int J = decltype(jj)::value;
                if constexpr (MORE && I == NT - 1) {        // last use of the activation fragments J, J + 1 was just issued
                    X[J] = frag(s + 1, x_off + J * FS);
                    X[J + 1] = frag(s + 1, x_off + (J + 1) * FS);
                }
            };
            // DMA of step s + 2 as EARLY in the step as the issue slots allow (rows 0 and 1, one piece per MFMA pair): its data
            // is waited for at the top of step s + 1, so every cycle it is issued earlier is latency hidden (spreading the pieces
            // evenly over the four rows, which pays in the one-wave-per-SIMD kernel of gemm_w4.hip, measured -2.8 % here: with two
            // waves per SIMD the partner covers a queued piece's issue stall)
            constexpr int PPR = MT >= 8 ? 4 : 2;            // pieces per row
            two(std::integral_constant<int, 0>{});
            dma2(std::integral_constant<int, (I < 2 ? PPR * I : NP)>{});
            reloadx(std::integral_constant<int, 0>{});
            two(std::integral_constant<int, 2>{});
            dma2(std::integral_constant<int, (I < 2 ? PPR * I + 1 : NP)>{});
            reloadx(std::integral_constant<int, 2>{});
            if constexpr (MT >= 8) {
                two(std::integral_constant<int, 4>{});
                dma2(std::integral_constant<int, (I < 2 ? PPR * I + 2 : NP)>{});
                reloadx(std::integral_constant<int, 4>{});
                two(std::integral_constant<int, 6>{});
                dma2(std::integral_constant<int, (I < 2 ? PPR * I + 3 : NP)>{});
                reloadx(std::integral_constant<int, 6>{});
            }
            if constexpr (MT < 8 && I < 2) dma2(std::integral_constant<int, 4 + I>{});   // 128-row tiles: 6 pieces, 3 per row
            if constexpr (MORE) Wf[I] = frag(s + 1, w_off + I * FS);   // row I done: its W fragment is free
            __builtin_amdgcn_sched_barrier(0);
        };
        row(std::integral_constant<int, 0>{});
        row(std::integral_constant<int, 1>{});
        row(std::integral_constant<int, 2>{});
        row(std::integral_constant<int, 3>{});
    };
    using T = std::true_type;
    using F = std::false_type;
    int s = 0;
    for (; s + 2 < ns; ++s) step(T{}, T{}, s);
    if (s + 1 < ns) { step(T{}, F{}, s); ++s; }
    step(F{}, F{}, s);
    tile_epilogue<MT, Epi>(acc, ep, M, N, n_cover, m0, n0, wm, wn, fr, kg);
}

template <int MT, typename Epi>
static int launch_cfg8(const void* A, int64_t lda, const uint8_t* a_scale, const void* W, int64_t ldw, const uint8_t* w_scale, int64_t M,
                       int N, int K, int n_cover, const EpiParams& ep, hipStream_t s) {
    constexpr int BM = 2 * MT * 16;
    const int tiles_m = (int)ceil_div(M, BM), tiles_n = (int)ceil_div(n_cover, 256);
    gemm_nt_fp8_kernel<MT, Epi><<<dim3((unsigned)(tiles_m * tiles_n)), 512, 0, s>>>((const uint8_t*)A, lda, a_scale, (const uint8_t*)W, ldw, w_scale,
                                                                                   M, N, K, tiles_m, tiles_n, n_cover, ep);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

// gemm_fp8_w4.hip: persistent four-wave form (P2T_ERR_UNSUPPORTED when the shape is not eligible)
template <typename Epi>
int launch_gemm_fp8_w4(const void* A, int64_t lda, const uint8_t* a_scale, const void* W, int64_t ldw, const uint8_t* w_scale, int64_t M, int N,
                       int K, int n_cover, int grid, const EpiParams& ep, hipStream_t s);
#ifdef P2T_LAB
int launch_gemm_fp8_w4_diag(int diag, const void* A, int64_t lda, const uint8_t* a_scale, const void* W, int64_t ldw, const uint8_t* w_scale, int64_t M,
                            int N, int K, int n_cover, int grid, const EpiParams& ep, hipStream_t s);
#endif
template <typename Epi>
constexpr bool kHasFp8W4 = std::is_same<Epi, EpiStore<bf16_t>>::value || std::is_same<Epi, EpiResid>::value || std::is_same<Epi, EpiSwiglu<bf16_t>>::value ||
                           std::is_same<Epi, EpiQkvRope<bf16_t>>::value || std::is_same<Epi, EpiGeluFp8>::value;

template <typename Epi>
static int launch_shape8(const void* A, int64_t lda, const uint8_t* a_scale, const void* W, int64_t ldw, const uint8_t* w_scale, int64_t M,
                         int N, int K, int n_cover, const EpiParams& ep, int tile, hipStream_t s) {
    // 256-row tiles unless 128-row tiles fill the chip better (same rule as the bf16 per-tile kernels)
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    // tile: 0 = four-wave persistent kernel when the shape allows, else the per-tile kernel below; 4 = four-wave kernel or
    // error; 128 / 256 = per-tile eight-wave kernel of that height
    if ((tile == 0 && get_gemm_policy() != 9) || tile == 4) {      // p2t_set_gemm_policy(9): without the four-wave forms (A/B, identity tests)
        if constexpr (kHasFp8W4<Epi>) {
            const int rc = launch_gemm_fp8_w4<Epi>(A, lda, a_scale, W, ldw, w_scale, M, N, K, n_cover, cus, ep, s);
            if (rc != P2T_ERR_UNSUPPORTED) return rc;
        }
        if (tile == 4) {
            set_error("gemm (fp8): the four-wave kernel needs M, N %% 256 == 0, K %% 256 == 0, K >= 512, at least one tile per CU and a bf16 / fp32-residual / e4m3 epilogue");
            return P2T_ERR_UNSUPPORTED;
        }
    }
    const int64_t tn = ceil_div(n_cover, 256), tm256 = ceil_div(M, 256), tm128 = ceil_div(M, 128);
    const double cost256 = (double)ceil_div(tm256 * tn, cus);
    const double cost128 = (double)ceil_div(tm128 * tn, cus) * 0.625 * 1.08;
#ifdef P2T_LAB
    if constexpr (std::is_same<Epi, EpiStore<bf16_t>>::value) {
        if (tile >= 2001 && tile <= 2005) return launch_gemm_fp8_w4_diag(tile - 2000, A, lda, a_scale, W, ldw, w_scale, M, N, K, n_cover, cus, ep, s);
        if (tile == 2011) return launch_gemm_fp8_w4_diag(1, A, lda, a_scale, W, ldw, w_scale, M, N, K, n_cover, 64, ep, s);      // stamps, 64 workgroups only
    }
#endif
    if constexpr (std::is_same<Epi, EpiStore<bf16_t>>::value) {
        if (tile >= 1001 && tile <= 1003) {
            const int tiles_m = (int)tm256, tiles_n = (int)tn;
            const dim3 grid((unsigned)(tiles_m * tiles_n));
            if (tile == 1001) gemm_nt_fp8_kernel<8, Epi, true, 1><<<grid, 512, 0, s>>>((const uint8_t*)A, lda, a_scale, (const uint8_t*)W, ldw, w_scale, M, N, K, tiles_m, tiles_n, n_cover, ep);
            if (tile == 1002) gemm_nt_fp8_kernel<8, Epi, true, 2><<<grid, 512, 0, s>>>((const uint8_t*)A, lda, a_scale, (const uint8_t*)W, ldw, w_scale, M, N, K, tiles_m, tiles_n, n_cover, ep);
            if (tile == 1003) gemm_nt_fp8_kernel<8, Epi, true, 3><<<grid, 512, 0, s>>>((const uint8_t*)A, lda, a_scale, (const uint8_t*)W, ldw, w_scale, M, N, K, tiles_m, tiles_n, n_cover, ep);
            P2T_LAUNCH_CHECK();
            return P2T_OK;
        }
    }
    if (tile == 256 || (tile != 128 && cost256 <= cost128)) return launch_cfg8<8, Epi>(A, lda, a_scale, W, ldw, w_scale, M, N, K, n_cover, ep, s);
    return launch_cfg8<4, Epi>(A, lda, a_scale, W, ldw, w_scale, M, N, K, n_cover, ep, s);
}

int launch_gemm_fp8(const void* A, int64_t lda, const uint8_t* a_scale, const void* W, int64_t ldw, const uint8_t* w_scale, int64_t M, int N,
                    int K, int n_cover, int out_dtype, int epilogue, const EpiParams& ep, int tile, hipStream_t s) {
    const bool ob = out_dtype == P2T_BF16;
#define P2T_FP8_CASE(E) return launch_shape8<E>(A, lda, a_scale, W, ldw, w_scale, M, N, K, n_cover, ep, tile, s)
    switch (epilogue) {
        case P2T_EPI_STORE: if (ob) P2T_FP8_CASE(EpiStore<bf16_t>); else P2T_FP8_CASE(EpiStore<float>);
        case P2T_EPI_GELU: { using GeluB = EpiGelu<bf16_t, true>; using GeluF = EpiGelu<float, true>; if (ob) P2T_FP8_CASE(GeluB); else P2T_FP8_CASE(GeluF); }
        case P2T_EPI_RESID: P2T_FP8_CASE(EpiResid);
        case P2T_EPI_SWIGLU: if (ob) P2T_FP8_CASE(EpiSwiglu<bf16_t>); else P2T_FP8_CASE(EpiSwiglu<float>);
        case P2T_EPI_STORE_F32: P2T_FP8_CASE(EpiF32);
        case P2T_EPI_QKV_ROPE: if (ob) P2T_FP8_CASE(EpiQkvRope<bf16_t>); else P2T_FP8_CASE(EpiQkvRope<float>);
        case P2T_EPI_GELU_FP8: P2T_FP8_CASE(EpiGeluFp8);
    }
#undef P2T_FP8_CASE
    set_error("gemm (fp8): unsupported epilogue %d", epilogue);
    return P2T_ERR_ARG;
}

#ifdef P2T_LAB
// lab build only: bf16 operands through the 64-deep single-barrier skeleton (policy 6).  K in elements, strides in elements.
template <typename Epi>
static int launch_k64(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover, const EpiParams& ep,
                      hipStream_t s) {
    const int tiles_m = (int)ceil_div(M, 256), tiles_n = (int)ceil_div(n_cover, 256);
    gemm_nt_fp8_kernel<8, Epi, false><<<dim3((unsigned)(tiles_m * tiles_n)), 512, 0, s>>>((const uint8_t*)A, lda * 2, nullptr, (const uint8_t*)W, ldw * 2,
                                                                                         nullptr, M, N, K * 2, tiles_m, tiles_n, n_cover, ep);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}
int launch_gemm_bf16_k64(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover, int out_dtype,
                         int epilogue, const EpiParams& ep, hipStream_t s) {
    if (K % 64) return P2T_ERR_UNSUPPORTED;
    const bool ob = out_dtype == P2T_BF16;
    switch (epilogue) {
        case P2T_EPI_STORE: return ob ? launch_k64<EpiStore<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, s) : launch_k64<EpiStore<float>>(A, lda, W, ldw, M, N, K, n_cover, ep, s);
        case P2T_EPI_GELU: return ob ? launch_k64<EpiGelu<bf16_t, true>>(A, lda, W, ldw, M, N, K, n_cover, ep, s) : launch_k64<EpiGelu<float, true>>(A, lda, W, ldw, M, N, K, n_cover, ep, s);
        case P2T_EPI_RESID: return launch_k64<EpiResid>(A, lda, W, ldw, M, N, K, n_cover, ep, s);
    }
    return P2T_ERR_UNSUPPORTED;
}
#endif  // P2T_LAB

}  // namespace p2t
