// Pieces shared by the bf16 (gemm_mfma.hip) and fp8 (gemm_fp8.hip) MFMA GEMM kernels: both use 16x16 accumulator tiles in
// the same wave layout (8 waves as 2 (M) x 4 (N), W rows permuted at staging), so the tile order and the epilogue -- which
// only sees the C/D fragment, dtype independent on gfx950 -- are common.
#pragma once
#include "common.h"
#include <type_traits>

#include "epilogue.h"

namespace p2t {

// Split-K fix-up of the tail round (gemm_mfma.hip gemm_nt_mfma_tail_kernel, gemm_w4.hip): one fp32 accumulator slab per tail tile in the
// register order of the block (thread t, register quad i -> float4 slab[i * 512 + t]) and one flag word per tile.
struct SplitFix {
    float* slab;            // [n_tail][32][512] float4 = 256 KiB per tile
    unsigned* flag;         // [n_tail], holds the epoch of the last completed producer
    unsigned* timeout;      // set to 1 if a consumer ever gave up waiting (never expected; bounded spin)
    unsigned epoch;         // unique per launch within one zeroing of `flag`
};
enum { TILE_FULL = 0, TILE_PRODUCE = 1, TILE_CONSUME = 2 };


// (tm, tn) of work item `id` out of `n_items`: the 8 XCDs get contiguous chunks (bijective for any count),
// inside a chunk GM row-tiles are walked column-major so neighbouring CUs share activation panels.
__device__ __forceinline__ void tile_coords(int id, int n_items, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int q8 = n_items >> 3, r8 = n_items & 7, xcd = id & 7;
    const int swz_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    constexpr int GM = 4;
    const int band = swz_id / (GM * tiles_n), first_m = band * GM;
    const int gm = min(GM, tiles_m - first_m);
    const int in_band = swz_id - band * GM * tiles_n;
    tm = first_m + in_band % gm;
    tn = in_band / gm;
}

// Epilogue of one tile: lane owns row m, columns nb .. nb+7 (n-tiles 0,1) and nb+32 .. nb+39 (n-tiles 2,3).
// INTERIOR: the tile lies fully inside the output (no row / column checks, one basic block).  Read-modify-write
// epilogues (Epi::kRmw) then fetch their operand for four rows at a time before the first add, so the HBM latency of
// the residual read is paid twice per tile instead of once per row.
// ADD: `addend` holds a second set of accumulators of the same tile (split-K partner), quad (i, j) of this thread at
// addend[(i * MT + j) * add_stride] -- added on the way into the epilogue, so the accumulator registers themselves are never
// redefined under control flow (which costs the register allocator hundreds of spills); `poison` (0 or NaN) is added too.
template <int MT, typename Epi, bool INTERIOR = false, bool ADD = false>
__device__ __forceinline__ void tile_epilogue(const f32x4 (&acc)[4][MT], const EpiParams& ep, int64_t M, int N, int n_cover,
                                              int64_t m0, int n0, int wm, int wn, int fr, int kg, const float4* addend = nullptr,
                                              int add_stride = 0, float poison = 0.f) {
    const int nb = n0 + wn * 64 + kg * 8;
    if (!INTERIOR && nb >= n_cover) return;
    float b0[8], b1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) b0[e] = b1[e] = 0.f;
    if (ep.bias) {
        if (INTERIOR || nb < N) loadW<8>(ep.bias + nb, b0);
        if (INTERIOR || nb + 32 < N) loadW<8>(ep.bias + nb + 32, b1);
    }
    if constexpr (INTERIOR && Epi::kRmw && MT % 4 == 0 && !ADD) {
#pragma unroll
        for (int jb = 0; jb < MT; jb += 4) {
            float r0[4][8], r1[4][8];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) Epi::template fetch2<8>(ep, m0 + wm * MT * 16 + (jb + jj) * 16 + fr, nb, r0[jj], r1[jj]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = jb + jj;
                const float v0[8] = {acc[0][j][0], acc[0][j][1], acc[0][j][2], acc[0][j][3], acc[1][j][0], acc[1][j][1], acc[1][j][2], acc[1][j][3]};
                const float v1[8] = {acc[2][j][0], acc[2][j][1], acc[2][j][2], acc[2][j][3], acc[3][j][0], acc[3][j][1], acc[3][j][2], acc[3][j][3]};
                Epi::template apply2_fetched<8>(ep, m0 + wm * MT * 16 + j * 16 + fr, nb, v0, v1, b0, b1, r0[jj], r1[jj]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int64_t m = m0 + wm * MT * 16 + j * 16 + fr;
            if (!INTERIOR && m >= M) continue;
            float v0[8] = {acc[0][j][0], acc[0][j][1], acc[0][j][2], acc[0][j][3], acc[1][j][0], acc[1][j][1], acc[1][j][2], acc[1][j][3]};
            float v1[8] = {acc[2][j][0], acc[2][j][1], acc[2][j][2], acc[2][j][3], acc[3][j][0], acc[3][j][1], acc[3][j][2], acc[3][j][3]};
            if constexpr (ADD) {
                const float4 p0 = addend[(0 * MT + j) * add_stride], p1 = addend[(1 * MT + j) * add_stride];
                const float4 p2 = addend[(2 * MT + j) * add_stride], p3 = addend[(3 * MT + j) * add_stride];
                v0[0] += p0.x + poison; v0[1] += p0.y + poison; v0[2] += p0.z + poison; v0[3] += p0.w + poison;
                v0[4] += p1.x + poison; v0[5] += p1.y + poison; v0[6] += p1.z + poison; v0[7] += p1.w + poison;
                v1[0] += p2.x + poison; v1[1] += p2.y + poison; v1[2] += p2.z + poison; v1[3] += p2.w + poison;
                v1[4] += p3.x + poison; v1[5] += p3.y + poison; v1[6] += p3.z + poison; v1[7] += p3.w + poison;
            }
            Epi::template apply2<8, INTERIOR>(ep, m, nb, v0, v1, b0, b1);
        }
    }
}

// Epilogue of BOTH 64-column halves of a four-wave kernel's 128 x 128 wave tile (tiles fully inside the output only), written so
// that no wait inside it -- and none behind it -- drains the stores it has already issued.  vmcnt retires in order, so a wait for
// a load that was issued AFTER a store also waits for that store's write to complete: tile_epilogue above pays that once per
// 64-column half for the bias and once per four rows for a read-modify-write operand (measured with in-kernel stamps,
// tools/w4_diag.py: 10 K cycles for the 32 stores of a plain bf16 tile, 47 K for the fp32 residual update, of a 86 K-cycle K loop).
//   * the bias of both halves is fetched before the first store (one wait, in front of everything);
//   * a read-modify-write operand is software-pipelined: the operand of group g + P is requested BEFORE the stores of group g are
//     issued, so the wait for it (counted by the compiler: only younger operations stay outstanding) never covers a store.
// Group g = (half h, row fragment j): the lane's row m0 + wm 128 + 16 j + fr, columns nb .. nb+7 and nb+32 .. nb+39 with
// nb = n0 + (2 wn + h) 64 + 8 kg.  Results are bit-identical to tile_epilogue (same arithmetic per element).
// The four accumulator quads of group (h, j) stay in their AGPRs until this point: the empty statement re-defines them (no code), so
// the compiler cannot place their v_accvgpr_read copies any earlier.  Left alone it reads all 256 accumulators into VGPRs in front of
// the epilogue, spills ~60 live registers (the next tile's fragments, the DMA offsets) to make room, and reloads them behind the
// last store -- a wait that drains every store of the tile.
template <typename Epi, typename = void> struct epi_has_fetch_shared : std::false_type {};
template <typename Epi> struct epi_has_fetch_shared<Epi, std::void_t<decltype(&Epi::fetch_shared)>> : std::true_type {};
template <typename Epi, typename = void> struct epi_has_fetch : std::false_type {};
template <typename Epi> struct epi_has_fetch<Epi, std::void_t<decltype(Epi::kFetch)>> : std::true_type {};
#define P2T_EPI_PIN_GROUP(h, j) \
    asm volatile("" : "+a"(acc[h][0][j]), "+a"(acc[h][1][j]), "+a"(acc[h][2][j]), "+a"(acc[h][3][j])::"memory");
template <int MT, typename Epi, int P = 4>
__device__ __forceinline__ void tile_epilogue_pair(f32x4 (&acc)[2][4][MT], const EpiParams& ep, int64_t m0, int n0, int wm, int wn, int fr, int kg) {
    const int nb0 = n0 + wn * 128 + kg * 8;
    float b[2][2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) b[h][0][e] = b[h][1][e] = 0.f;
    if (ep.bias) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            loadW<8>(ep.bias + nb0 + h * 64, b[h][0]);
            loadW<8>(ep.bias + nb0 + h * 64 + 32, b[h][1]);
        }
    }
    const int64_t mrow = m0 + wm * MT * 16 + fr;
    if constexpr (epi_has_fetch_shared<Epi>::value && P > 0) {
        // One fetch serves both 64-column halves of a row (EpiQkvRope at head_dim 64: the rotary channels of a lane repeat in every block): rows
        // outermost, half the table reads -- they, not the arithmetic, are what this epilogue waits for (256 KB per CU and tile otherwise against
        // 128 KB of stores; stamped: 20.1 K -> 17.7 K cycles per tile).  Same values in the same arithmetic: bit-identical.
        if (Epi::fetch_shared(ep)) {
            constexpr int PS = P < MT ? P : MT;
            float r[PS][2][8];
#pragma unroll
            for (int u = 0; u < PS; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) r[u][0][e] = r[u][1][e] = 0.f;
#pragma unroll
            for (int u = 0; u < PS; ++u) Epi::template fetch2<8, true>(ep, mrow + u * 16, nb0, r[u][0], r[u][1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < MT; ++u) {
                float s0[8], s1[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { s0[e] = r[u % PS][0][e]; s1[e] = r[u % PS][1][e]; }
                __builtin_amdgcn_sched_barrier(0);
                if (u + PS < MT) Epi::template fetch2<8, true>(ep, mrow + (u + PS) * 16, nb0, r[u % PS][0], r[u % PS][1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    P2T_EPI_PIN_GROUP(h, u)
                    const float v0[8] = {acc[h][0][u][0], acc[h][0][u][1], acc[h][0][u][2], acc[h][0][u][3], acc[h][1][u][0], acc[h][1][u][1], acc[h][1][u][2], acc[h][1][u][3]};
                    const float v1[8] = {acc[h][2][u][0], acc[h][2][u][1], acc[h][2][u][2], acc[h][2][u][3], acc[h][3][u][0], acc[h][3][u][1], acc[h][3][u][2], acc[h][3][u][3]};
                    Epi::template apply2_fetched<8, true>(ep, mrow + u * 16, nb0 + h * 64, v0, v1, b[h][0], b[h][1], s0, s1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            return;
        }
    }
    if constexpr ((Epi::kRmw || epi_has_fetch<Epi>::value) && P > 0) {
        constexpr int G = 2 * MT;           // P: groups in flight (16 registers each; the fp8 kernel, whose next-tile fragments hold 96 registers, uses fewer)
        float r[P][2][8];
        if constexpr (!Epi::kRmw) {         // a fetch may be skipped (V heads of the QKV epilogue): defined values either way
#pragma unroll
            for (int g = 0; g < P; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) r[g][0][e] = r[g][1][e] = 0.f;
        }
#pragma unroll
        for (int g = 0; g < P; ++g) Epi::template fetch2<8, true>(ep, mrow + (g % MT) * 16, nb0 + (g / MT) * 64, r[g][0], r[g][1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int h = g / MT, j = g % MT;
            P2T_EPI_PIN_GROUP(h, j)
            const float v0[8] = {acc[h][0][j][0], acc[h][0][j][1], acc[h][0][j][2], acc[h][0][j][3], acc[h][1][j][0], acc[h][1][j][1], acc[h][1][j][2], acc[h][1][j][3]};
            const float v1[8] = {acc[h][2][j][0], acc[h][2][j][1], acc[h][2][j][2], acc[h][2][j][3], acc[h][3][j][0], acc[h][3][j][1], acc[h][3][j][2], acc[h][3][j][3]};
            float s0[8], s1[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { s0[e] = r[g % P][0][e]; s1[e] = r[g % P][1][e]; }
            __builtin_amdgcn_sched_barrier(0);
            if (g + P < G) Epi::template fetch2<8, true>(ep, mrow + ((g + P) % MT) * 16, nb0 + ((g + P) / MT) * 64, r[g % P][0], r[g % P][1]);
            __builtin_amdgcn_sched_barrier(0);
            Epi::template apply2_fetched<8, true>(ep, mrow + j * 16, nb0 + h * 64, v0, v1, b[h][0], b[h][1], s0, s1);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                P2T_EPI_PIN_GROUP(h, j)
                const float v0[8] = {acc[h][0][j][0], acc[h][0][j][1], acc[h][0][j][2], acc[h][0][j][3], acc[h][1][j][0], acc[h][1][j][1], acc[h][1][j][2], acc[h][1][j][3]};
                const float v1[8] = {acc[h][2][j][0], acc[h][2][j][1], acc[h][2][j][2], acc[h][2][j][3], acc[h][3][j][0], acc[h][3][j][1], acc[h][3][j][2], acc[h][3][j][3]};
                Epi::template apply2<8, true>(ep, mrow + j * 16, nb0 + h * 64, v0, v1, b[h][0], b[h][1]);
                __builtin_amdgcn_sched_barrier(0);      // one group at a time: hoisting all 256 accumulator reads costs spills, and a reload behind the stores drains them
            }
    }
}

}  // namespace p2t
