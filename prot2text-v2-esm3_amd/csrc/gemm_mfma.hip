// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T, fp32 accumulate, fused epilogue.
//
// Both operands are K-contiguous (activations [M][K], PyTorch Linear weights [N][K]), so A and W tiles stage and
// read identically.  Structure (chosen by in-process A/B runs of tools/gemm_lab.*, profiles/r01_gemm_lab.log):
//   * block tile BM x 256 (BM = 256 or 128), 8 wavefronts as 2 (M) x 4 (N), each owning (BM/2) x 64 outputs as
//     v_mfma_f32_16x16x32_bf16 accumulators;
//   * K is consumed in 32-deep stages through a 4-slot LDS ring filled by `global_load_lds_dwordx4` (no VGPR
//     round trip): three stages stay in flight across raw s_barriers behind counted `s_waitcnt vmcnt(N)`;
//   * fragments are double-buffered in registers: the ds_read_b128 of stage s+1 and the DMA issue of stage s+4 are
//     interleaved between the 32 MFMAs of stage s (sched_group_barrier), so the matrix pipe restarts right after
//     each barrier instead of waiting out the LDS latency;
//   * the LDS image is lane-linear, so the bank swizzle (64-byte rows: 16-B chunk c of row r stored at
//     c ^ ((-(r >> 2)) & 3), conflict-free for the 16x16x32 fragment pattern) is applied to the per-lane SOURCE
//     address and again on the ds_read (rule 21);
//   * operands are passed swapped (W fragment as MFMA "A") and the W rows of each 64-row group are PERMUTED while
//     staging (LDS row 16 i + 4 q + e  <-  weight row 32 (i >> 1) + 8 q + 4 (i & 1) + e), so a lane ends up with
//     8 consecutive output columns of one row, twice (columns n..n+7 and n+32..n+39): 16-byte stores, vector bias
//     loads, and the pairwise epilogues (SwiGLU gate/up, rotate-half RoPE partners) finish inside one lane;
//   * 1-D grid with an XCD-aware, grouped tile order: the 8 XCDs get contiguous chunks of the tile list and
//     consecutive tiles share activation row-panels (L2 reuse, technique T1).
// Requirements: K % 64 == 0 (callers pad K with zeros), lda/ldw % 8 == 0, 16-byte aligned bases.
#include <atomic>
#include <type_traits>

#include "common.h"
#include "epilogue.h"
#include "gemm_tile_common.h"
#include "kernels.h"

namespace p2t {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

// One BM x 256 output tile at (m0, n0) over K range [k0, k0 + 32 ns).  smem: 4 * (BM + 256) * 64 bytes.
template <int MT, typename Epi, int MODE = TILE_FULL>
__device__ __forceinline__ void gemm_tile(char* smem, const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W,
                                          int64_t ldw, int64_t M, int N, int k0, int ns, int64_t m0, int n0, int n_cover,
                                          const EpiParams& ep, const SplitFix* fix = nullptr, int fix_tile = 0) {
    constexpr int WN = 4, NT = 4, BM = 2 * MT * 16, BN = 256;
    constexpr int AI = BM / 128;                            // A staging instructions per wave per stage (1 KiB each)
    constexpr int SLOT = (BM + BN) * 64;                    // bytes per 32-deep stage

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w / WN, wn = w % WN;

    // ---- staging: one wave instruction = 16 rows x 64 B; lane -> (row = lane >> 2, slot = lane & 3) ----
    const int schunk = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
    const bf16_t* a_src[AI];
    const bf16_t* w_src[2];
#pragma unroll
    for (int t = 0; t < AI; ++t) {
        int64_t am = m0 + (w * AI + t) * 16 + (lane >> 2);
        am = am < M ? am : M - 1;
        a_src[t] = A + am * lda + k0 + schunk * 8;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int R = (w * 2 + t) * 16 + (lane >> 2), r = R & 63;
        const int nl = ((r >> 5) & 1) * 32 + ((r >> 2) & 3) * 8 + ((r >> 4) & 1) * 4 + (r & 3);     // permuted weight row
        int wr = n0 + (R & ~63) + nl;
        wr = wr < N ? wr : N - 1;
        w_src[t] = W + (int64_t)wr * ldw + k0 + schunk * 8;
    }
    auto stage = [&](int s) {
        char* base = smem + (s & 3) * SLOT;
#pragma unroll
        for (int t = 0; t < AI; ++t)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[t] + (int64_t)s * 32), (lptr_t)(base + (w * AI + t) * 1024), 16, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
            __builtin_amdgcn_global_load_lds((gptr_t)(w_src[t] + (int64_t)s * 32), (lptr_t)(base + BM * 64 + (w * 2 + t) * 1024), 16, 0, 0);
    };
    constexpr int NL = AI + 2;                              // DMA instructions per wave per stage

    // ---- fragment read offsets ----
    const int fr = lane & 15, kg = lane >> 4;
    const int sw = (kg ^ ((4 - ((fr >> 2) & 3)) & 3)) << 4;
    const int x_off = (wm * MT * 16 + fr) * 64 + sw;
    const int w_off = BM * 64 + (wn * NT * 16 + fr) * 64 + sw;

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_frags = [&](int s, bf16x8 (&xf)[MT], bf16x8 (&wf)[NT]) {
        const char* sb = smem + (s & 3) * SLOT;
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 1024);
#pragma unroll
        for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(sb + x_off + j * 1024);
    };
    auto mfma_all = [&](const bf16x8 (&xf)[MT], const bf16x8 (&wf)[NT]) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    };
    // s_waitcnt vmcnt(n) only (lgkmcnt / expcnt untouched): simm16 = vmcnt[3:0] | 0x70 | 0xF00 | vmcnt[5:4] << 14
    auto wait_vm = [&](int n_stages) {                      // n_stages of DMA (NL loads each) may remain in flight
        if (n_stages >= 3) __builtin_amdgcn_s_waitcnt(0x0F70 | ((3 * NL) & 15) | (((3 * NL) >> 4) << 14));
        else if (n_stages == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * NL) & 15) | (((2 * NL) >> 4) << 14));
        else if (n_stages == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (NL & 15));
        else __builtin_amdgcn_s_waitcnt(0x0F70);
    };
    // one 32-deep stage: stage s+1 must have landed, its fragments and the DMA of stage s+4 are issued between the
    // MFMAs of stage s.  FULL = steady state (stages s+1 .. s+4 exist): branch-free so the scheduler can interleave.
    auto step = [&](auto full, int s, bf16x8 (&xc)[MT], bf16x8 (&wc)[NT], bf16x8 (&xn)[MT], bf16x8 (&wnx)[NT]) {
        constexpr bool FULL = decltype(full)::value;
        if (FULL) wait_vm(2);
        else if (s + 1 < ns) wait_vm(min(ns - 1, s + 3) - (s + 1));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (FULL || s + 1 < ns) load_frags(s + 1, xn, wnx);
        if (FULL || s + 4 < ns) stage(s + 4);
        mfma_all(xc, wc);
        if (FULL) {
            constexpr int PER = (MT * NT) / (MT + NT + NL);              // MFMAs between two memory issues
#pragma unroll
            for (int qq = 0; qq < MT + NT; ++qq) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);    // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
            }
#pragma unroll
            for (int qq = 0; qq < NL; ++qq) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read (LDS DMA)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                             // lgkmcnt(0): exact counts at the back edge
    };

    bf16x8 xa[MT], wa[NT], xb[MT], wb[NT];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < ns) stage(s);
    wait_vm(min(ns - 1, 3));                                            // stage 0 landed
    __builtin_amdgcn_s_barrier();
    load_frags(0, xa, wa);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    int s = 0;
    for (; s + 5 < ns; s += 2) {
        step(std::true_type{}, s, xa, wa, xb, wb);
        step(std::true_type{}, s + 1, xb, wb, xa, wa);
    }
    for (; s < ns; s += 2) {
        step(std::false_type{}, s, xa, wa, xb, wb);
        step(std::false_type{}, s + 1, xb, wb, xa, wa);
    }

    if constexpr (MODE == TILE_PRODUCE) {
        // second K half of a tail tile: publish the raw accumulators (plain 16-B stores, coalesced), then every wave
        // drains its stores, the block barriers, and one lane releases at agent scope and raises the tile's flag
        // (cdna_hip_programming.md Guideline 16, counter form).  This block never waits: it cannot deadlock.
        float4* slab = reinterpret_cast<float4*>(fix->slab) + (int64_t)fix_tile * (MT * NT * 512);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j)
                slab[(i * MT + j) * 512 + threadIdx.x] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(fix->flag + fix_tile, fix->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if constexpr (MODE == TILE_CONSUME) {
        // first K half: wait for the partner, acquire, add its slab.  Invariant that makes the wait finite: a producer
        // has a LOWER block id than its consumer, never waits on anything, and hardware dispatches workgroups in id order,
        // so by the time a consumer runs its producer has been dispatched (running or done) -- whatever else shares the
        // chip (other streams' grids only delay it).  The spin is bounded all the same; giving up is LOUD: the time-out
        // word stays set (sticky until the tower's next forward zeroes the header) and every consumer that sees it
        // writes NaN tiles, so the loss of the step is NaN (the reference loop aborts on that, train_contrast.py:476-480)
        // instead of a silently stale sum.
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(fix->flag + fix_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fix->epoch) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 22)) {                  // ~ seconds
                    __hip_atomic_store(fix->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const bool timed_out = __hip_atomic_load(fix->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        const float poison = timed_out ? __builtin_nanf("") : 0.f;
        const float4* slab = reinterpret_cast<const float4*>(fix->slab) + (int64_t)fix_tile * (MT * NT * 512);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const float4 p = slab[(i * MT + j) * 512 + threadIdx.x];
                acc[i][j][0] += p.x + poison; acc[i][j][1] += p.y + poison; acc[i][j][2] += p.z + poison; acc[i][j][3] += p.w + poison;
            }
    }

    tile_epilogue<MT, Epi>(acc, ep, M, N, n_cover, m0, n0, wm, wn, fr, kg);
}

template <int MT, typename Epi>
__global__ void __launch_bounds__(512)
    gemm_nt_mfma_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw, int64_t M,
                        int N, int K, int tiles_m, int tiles_n, int n_cover, EpiParams ep) {
    constexpr int BM = 2 * MT * 16;
    __shared__ __attribute__((aligned(16))) char smem[4 * (BM + 256) * 64];
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
    gemm_tile<MT, Epi>(smem, A, lda, W, ldw, M, N, 0, K >> 5, (int64_t)tm * BM, tn * 256, n_cover, ep);
}

// s_waitcnt vmcnt(n) only: simm16 = vmcnt[3:0] | expcnt 7 | lgkmcnt 15 | vmcnt[5:4] << 14
constexpr int vm_imm(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }

// Persistent form of the 256 x 256 tile: one block per CU walks work items blockIdx.x, + gridDim.x, ... and treats the
// K loops of consecutive tiles as ONE stream of stages -- the last three steps of a tile already issue the LDS DMA of
// the next tile's first three stages, so no tile after the first pays the ring fill, and the epilogue's stores are
// never drained: vmcnt completes in issue order (MI355X_MICROARCH.md, s_waitcnt), so the first two steps after an
// epilogue wait with a count that leaves the epilogue's operations in flight (Epi::kMinOps is a lower bound of them;
// the kernel only takes shapes without edge tiles, where every wave issues all of them).  The write burst of a round (all CUs finish together: 32 MB, ~6 us of HBM time) then overlaps the next tile's MFMAs.
// Requirements on top of the plain kernel: M % 256 == 0, N % 256 == 0 == n_cover (no edge tiles), (K / 32) % 4 == 0,
// K >= 384, n_items >= gridDim.x.
template <typename Epi>
__global__ void __launch_bounds__(512)
    gemm_nt_mfma_persist_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw, int64_t M,
                                int N, int K, int tiles_m, int tiles_n, int n_items, int n_tail, int half_tail, int n_cover,
                                EpiParams ep, SplitFix fix) {
    constexpr int MT = 8, NT = 4, WN = 4, SLOT = 512 * 64, NL = 4;
    constexpr int kEpiOps = MT * Epi::kMinOps;
    constexpr int kExtCount = NL + kEpiOps > 63 ? 63 : NL + kEpiOps;
    __shared__ __attribute__((aligned(16))) char smem[4 * SLOT];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w / WN, wn = w % WN;
    const int ns = K >> 5;

    const int schunk = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
    // DMA sources as a uniform 64-bit base (SGPRs: tile origin + K offset) plus a 32-bit per-lane byte offset that is
    // the same for every tile (the persistent kernel only takes shapes whose tiles all lie inside the matrices).
    const char* a_base;
    const char* w_base;
    uint32_t a_voff[2], w_voff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int R = (w * 2 + t) * 16 + (lane >> 2), r = R & 63;
        const int wr = (R & ~63) + ((r >> 5) & 1) * 32 + ((r >> 2) & 3) * 8 + ((r >> 4) & 1) * 4 + (r & 3);   // permuted weight row
        a_voff[t] = (uint32_t)(R * (int)lda + schunk * 8) * 2u;
        w_voff[t] = (uint32_t)(wr * (int)ldw + schunk * 8) * 2u;
    }
    auto set_src = [&](int64_t m0, int n0) {
        a_base = (const char*)(A + m0 * lda);
        w_base = (const char*)(W + (int64_t)n0 * ldw);
    };
    // LDS DMA as inline asm (saddr form: uniform base in SGPRs + 32-bit lane offset; M0 = LDS address of the wave's
    // 1 KiB piece).  Deliberately NOT the __builtin: LLVM's waitcnt pass treats an LDS DMA as a pending "flat" access
    // and then turns EVERY counted wait it inserts (the lgkmcnt of fragment reads carried across steps, the vmcnt of
    // epilogue loads) into a wait for zero for as long as a DMA is in flight -- which here is always.  Untracked, the
    // DMAs only ever make the compiler's own vmcnt waits conservative (they complete in issue order and none is issued
    // inside an epilogue); the ring's ordering is enforced by the explicit counted waits below.
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto dma = [&](auto which, int slot, int koff) {
        constexpr int WH = decltype(which)::value;                     // 0,1: activation rows; 2,3: weight rows
        const char* sb = (WH < 2 ? a_base : w_base) + koff * 2;
        const uint32_t vo = WH < 2 ? a_voff[WH & 1] : w_voff[WH & 1];
        const uint32_t lds = lds0 + slot * SLOT + (WH < 2 ? 0 : 256 * 64) + (w * 2 + (WH & 1)) * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(sb), "s"(lds) : "memory");
    };
    using D0 = std::integral_constant<int, 0>;
    using D1 = std::integral_constant<int, 1>;
    using D2 = std::integral_constant<int, 2>;
    using D3 = std::integral_constant<int, 3>;
    auto stage = [&](int slot, int koff) {
        dma(D0{}, slot, koff); dma(D1{}, slot, koff); dma(D2{}, slot, koff); dma(D3{}, slot, koff);
    };

    const int fr = lane & 15, kg = lane >> 4;
    const int sw = (kg ^ ((4 - ((fr >> 2) & 3)) & 3)) << 4;
    const int x_off = (wm * MT * 16 + fr) * 64 + sw;
    const int w_off = 256 * 64 + (wn * NT * 16 + fr) * 64 + sw;

    // Register plan (256 per lane at two waves per SIMD): 128 accumulators, the W fragments double-buffered (2 x 16),
    // the activation fragments SINGLE-buffered (32): the MFMAs run activation-fragment-major, and fragment j of the next
    // stage is re-read into the same registers once its four MFMAs are issued.  Those reads complete in the following
    // step, so a slot is recycled one step later than in the per-tile kernel: the DMA of stage s+3 (not s+4) goes into
    // the slot of stage s-1, whose fragments every wave consumed before this step's barrier -- no LDS wait at the barrier.
    f32x4 acc[NT][MT];
    bf16x8 x[MT], wa[NT], wb[NT];
    // One 32-deep stage s as 16 pairs of MFMAs (pair p: activation fragment p/2, W fragments 2(p&1), 2(p&1)+1), each
    // pair fenced with ONE memory operation so the matrix pipe never waits for an issue slot:
    //   W'0 G0 W'1 X'0 G1 W'2 X'1 G2 W'3 X'2 G3 X'3 X'4 X'5 X'6 -- X'7   (' = of stage s+1, G = DMA piece of stage s+3)
    // X'j follows the last MFMA of fragment j (X'7 trails the last pair; it is not needed before pair 14 of step s+1).
    // WAITK: 3 = steady state (one stage may stay in flight; plus the previous epilogue's operations when `ext`), 0..2 =
    // that many stages, 4 / 5 = the last two steps of a tile, 6 = 3 for the first step of a tile (C operand = 0).  LOADF: read the fragments of stage s+1 (slot fslot).  ISSUE (compile time) and
    // `more` (run time): start the DMA of stage s+3 into slot islot at element offset koff of the issue pointers.
    auto step = [&](auto waitk, auto loadf, auto issue, bool ext, bool more, int fslot, int islot, int koff, const bf16x8 (&wc)[NT],
                    bf16x8 (&wnx)[NT]) {
        constexpr int WAITK = decltype(waitk)::value;
        constexpr bool LOADF = decltype(loadf)::value, ISSUE = decltype(issue)::value;
        constexpr bool FIRST = WAITK == 6;              // first step of a tile: the accumulators start from the inline constant 0
        if constexpr (WAITK == 3 || WAITK == 6) {
            if (ext) __builtin_amdgcn_s_waitcnt(vm_imm(kExtCount));
            else __builtin_amdgcn_s_waitcnt(vm_imm(NL));
        } else if constexpr (WAITK == 4) {              // second to last step of a tile: steady-state wait, or drain
            if (more) __builtin_amdgcn_s_waitcnt(vm_imm(NL));
            else __builtin_amdgcn_s_waitcnt(vm_imm(0));
        } else if constexpr (WAITK == 5) {              // last step: steady-state wait, or nothing left to wait for
            if (more) __builtin_amdgcn_s_waitcnt(vm_imm(NL));
        } else if constexpr (WAITK >= 0) {
            __builtin_amdgcn_s_waitcnt(vm_imm(WAITK * NL));
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* xs = smem + fslot * SLOT + x_off;
        const char* ws = smem + fslot * SLOT + w_off;
        auto rw = [&](int i) { if constexpr (LOADF) wnx[i] = *reinterpret_cast<const bf16x8*>(ws + i * 1024); };
        auto rx = [&](int j) { if constexpr (LOADF) x[j] = *reinterpret_cast<const bf16x8*>(xs + j * 1024); };
        auto g = [&](auto which) { if constexpr (ISSUE) { if (more) dma(which, islot, koff); } };
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        auto pair = [&](int p) {
            const int j = p >> 1, i0 = (p & 1) * 2;
            acc[i0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[i0], x[j], FIRST ? zero4 : acc[i0][j], 0, 0, 0);
            acc[i0 + 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[i0 + 1], x[j], FIRST ? zero4 : acc[i0 + 1][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        rw(0); pair(0);
        g(D0{}); pair(1);
        rw(1); pair(2);
        rx(0); pair(3);
        g(D1{}); pair(4);
        rw(2); pair(5);
        rx(1); pair(6);
        g(D2{}); pair(7);
        rw(3); pair(8);
        rx(2); pair(9);
        g(D3{}); pair(10);
        rx(3); pair(11);
        rx(4); pair(12);
        rx(5); pair(13);
        rx(6); pair(14);
        pair(15);
        rx(7);
    };
    using I1 = std::integral_constant<int, 1>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>;
    using T = std::true_type;
    using F = std::false_type;

    int item = blockIdx.x;
    int tm, tn;
    tile_coords(item, n_items + n_tail, tiles_m, tiles_n, tm, tn);
    int64_t m0 = (int64_t)tm * 256;
    int n0 = tn * 256;
    set_src(m0, n0);
#pragma unroll
    for (int s = 0; s < 3; ++s) stage(s, s * 32);
    __builtin_amdgcn_s_waitcnt(vm_imm(2 * NL));
    __builtin_amdgcn_s_barrier();
    bool ext = false;
    for (;;) {
        // fragments of this tile's stage 0 (landed: prologue above, or the barrier of the previous tile's last step)
#pragma unroll
        for (int i = 0; i < NT; ++i) wa[i] = *reinterpret_cast<const bf16x8*>(smem + w_off + i * 1024);
#pragma unroll
        for (int j = 0; j < MT; ++j) x[j] = *reinterpret_cast<const bf16x8*>(smem + x_off + j * 1024);
        const int nxt = item + (int)gridDim.x;
        const bool has_next = nxt < n_items;
        // steps 0, 1: the epilogue of the previous tile may still be in flight behind stages 1 / 2
        step(I6{}, T{}, T{}, ext, true, 1, 3, 3 * 32, wa, wb);
        step(I3{}, T{}, T{}, ext, true, 2, 0, 4 * 32, wb, wa);
        int s = 2;
        for (; s + 4 < ns; s += 2) {
            step(I3{}, T{}, T{}, false, true, (s + 1) & 3, (s + 3) & 3, (s + 3) * 32, wa, wb);
            step(I3{}, T{}, T{}, false, true, (s + 2) & 3, s & 3, (s + 4) * 32, wb, wa);
        }
        // s == ns - 4 (a multiple of 4): stages s .. s+3 sit in slots 0 .. 3; one more step issues a stage of this tile
        int64_t nm0 = 0;
        int nn0 = 0;
        step(I3{}, T{}, T{}, false, true, 1, 3, (s + 3) * 32, wa, wb);
        if (has_next) {
            tile_coords(nxt, n_items + n_tail, tiles_m, tiles_n, tm, tn);
            nm0 = (int64_t)tm * 256;
            nn0 = tn * 256;
            set_src(nm0, nn0);
        }
        // last three steps: with a next tile they issue its stages 0..2 and wait like steady-state steps (a steady-state
        // wait is also right for the first of them without one); without, the last two drain the ring
        // (one code path: two arms that each define the accumulators cost ~330 spilled registers)
        step(I1{}, T{}, T{}, false, has_next, 2, 0, 0, wb, wa);
        step(I4{}, T{}, T{}, false, has_next, 3, 1, 32, wa, wb);
        step(I5{}, F{}, T{}, false, has_next, 0, 2, 64, wb, wa);      // with a next tile: its stage 0 has landed behind this barrier
        {
            // launder the lane coordinates so the per-row output addresses are rebuilt here, per tile, instead of being
            // hoisted out of the tile loop and held (then spilled) across the main loop
            int fr_e = fr, kg_e = kg;
            asm volatile("" : "+v"(fr_e), "+v"(kg_e));
            tile_epilogue<MT, Epi, true>(acc, ep, M, N, n_cover, m0, n0, wm, wn, fr_e, kg_e);
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the next fragment reads behind the epilogue (register pressure)
        if (!has_next) break;
        ext = true;                                 // every wave issued at least kEpiOps operations in that epilogue
        item = nxt;
        m0 = nm0;
        n0 = nn0;
    }
    // Split-K fix-up of a partial last round (see gemm_nt_mfma_tail_kernel): n_items counts the whole rounds only; each of
    // the n_tail leftover tiles runs as two K halves on two blocks -- producers on blocks [0, n_tail) (they never wait),
    // consumers on [n_tail, 2 n_tail): a producer has the lower block id and never waits (see gemm_tile, TILE_CONSUME).
    // half_tail: the same leftover tiles, each as two 128-row halves on two blocks over the whole K (no hand-off): the
    // partial round then costs one 128 x 256 tile (~0.62 of a tile time) instead of a whole one -- for K too short for
    // the split-K form to pay.
    if (n_tail > 0 && (int)blockIdx.x < 2 * n_tail) {
        const int ns0 = (ns >> 1) & ~1;
        __builtin_amdgcn_s_waitcnt(vm_imm(0));                 // every DMA of the tile loop has landed before the ring is reused
        __builtin_amdgcn_s_barrier();
        if (half_tail) {
            const int t = blockIdx.x >> 1, h = blockIdx.x & 1;
            tile_coords(n_items + t, n_items + n_tail, tiles_m, tiles_n, tm, tn);
            gemm_tile<4, Epi>(smem, A, lda, W, ldw, M, N, 0, ns, (int64_t)tm * 256 + h * 128, tn * 256, n_cover, ep);
        } else if ((int)blockIdx.x < n_tail) {
            const int t = blockIdx.x;
            tile_coords(n_items + t, n_items + n_tail, tiles_m, tiles_n, tm, tn);
            gemm_tile<8, Epi, TILE_PRODUCE>(smem, A, lda, W, ldw, M, N, ns0 * 32, ns - ns0, (int64_t)tm * 256, tn * 256, n_cover, ep, &fix, t);
        } else {
            const int t = blockIdx.x - n_tail;
            tile_coords(n_items + t, n_items + n_tail, tiles_m, tiles_n, tm, tn);
            gemm_tile<8, Epi, TILE_CONSUME>(smem, A, lda, W, ldw, M, N, 0, ns0, (int64_t)tm * 256, tn * 256, n_cover, ep, &fix, t);
        }
    }
}

// Wave-quantisation fix for grids that are not a whole number of "rounds" of the 256 CUs (o-proj / FFN-down of the
// encoder: 640 tiles = 2.5 rounds): the first n_full blocks take whole tiles; each of the n_tail leftover tiles is
// split in two K halves run by two CUs at the same time -- a producer block (second half; dispatched FIRST, never
// waits) and a consumer block (first half, then adds the producer's slab and runs the epilogue).  The last partial
// round then costs about half a round plus the 256 KiB slab hand-off instead of a full round.
template <typename Epi>
__global__ void __launch_bounds__(512)
    gemm_nt_mfma_tail_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw, int64_t M,
                             int N, int K, int tiles_m, int tiles_n, int n_full, int n_tail, int n_cover, EpiParams ep,
                             SplitFix fix) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 512 * 64];
    const int bid = blockIdx.x, ns = K >> 5, ns0 = (ns >> 1) & ~1;     // consumer: stages [0, ns0), producer: [ns0, ns)
    int tm, tn;
    if (bid < n_full) {
        tile_coords(bid, n_full + n_tail, tiles_m, tiles_n, tm, tn);   // tail tiles = the last ids of the same order
        gemm_tile<8, Epi>(smem, A, lda, W, ldw, M, N, 0, ns, (int64_t)tm * 256, tn * 256, n_cover, ep);
    } else if (bid < n_full + n_tail) {
        const int t = bid - n_full;
        tile_coords(n_full + t, n_full + n_tail, tiles_m, tiles_n, tm, tn);
        gemm_tile<8, Epi, TILE_PRODUCE>(smem, A, lda, W, ldw, M, N, ns0 * 32, ns - ns0, (int64_t)tm * 256, tn * 256, n_cover, ep, &fix, t);
    } else {
        const int t = bid - n_full - n_tail;
        tile_coords(n_full + t, n_full + n_tail, tiles_m, tiles_n, tm, tn);
        gemm_tile<8, Epi, TILE_CONSUME>(smem, A, lda, W, ldw, M, N, 0, ns0, (int64_t)tm * 256, tn * 256, n_cover, ep, &fix, t);
    }
}

// gemm_w4.hip: persistent four-wave form over the first n_items tiles of the order of n_order tiles
template <typename Epi>
int launch_gemm_w4_persist(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_items, int n_tail, int grid,
                           const EpiParams& ep, const SplitFix& fix, hipStream_t s, int sched = 1);
int launch_gemm_w4(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover, int out_dtype, int epilogue,
                   const EpiParams& ep, hipStream_t s);
template <typename Epi>
int launch_gemm_w4_pairs(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_tail, const EpiParams& ep,
                         const SplitFix& fix, hipStream_t s);
template <typename Epi>
constexpr bool kHasW4Pairs = std::is_same<Epi, EpiResid>::value || std::is_same<Epi, EpiStore<bf16_t>>::value || std::is_same<Epi, EpiStore<float>>::value;
template <typename Epi>
constexpr bool kHasW4 = std::is_same<Epi, EpiStore<bf16_t>>::value || std::is_same<Epi, EpiStore<float>>::value || std::is_same<Epi, EpiResid>::value ||
                        std::is_same<Epi, EpiQkvRope<bf16_t>>::value || std::is_same<Epi, EpiGelu<bf16_t>>::value || std::is_same<Epi, EpiSwiglu<bf16_t>>::value;

template <int MT, typename Epi>
static int launch_cfg(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover,
                      const EpiParams& ep, hipStream_t s) {
    constexpr int BM = 2 * MT * 16;
    const int tiles_m = (int)ceil_div(M, BM), tiles_n = (int)ceil_div(n_cover, 256);
    gemm_nt_mfma_kernel<MT, Epi><<<dim3((unsigned)(tiles_m * tiles_n)), 512, 0, s>>>(
        (const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, tiles_m, tiles_n, n_cover, ep);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

// measured (profiles/r01_microbench_v2.log): a 128-row tile takes ~0.62 of a 256-row tile.  A mixed grid (whole
// rounds of 256-row tiles + the leftover rows as 128-row tiles) was tried against the 2.5-round N = 2560 GEMMs and
// measured 5-13 % SLOWER than plain 256-row tiles, so it is not used.
constexpr double kSmallTileCost = 0.625;

// Compute units of the current device (256 on MI355X; fewer on a partitioned one): the persistent grid is one block per
// CU and the "round" arithmetic of the launch policy counts in CUs.
#ifdef P2T_LAB
static std::atomic<int> g_cu_override{0};       // lab build: p2t_set_gemm_policy(1000 + n) makes the launch policy count n compute units
void set_cu_override(int n) { g_cu_override.store(n, std::memory_order_relaxed); }
#endif
static int cu_count() {
#ifdef P2T_LAB
    if (const int o = g_cu_override.load(std::memory_order_relaxed)) return o;
#endif
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// Launch-form override (p2t_set_gemm_policy): 0 = the measured default policy below; 9 = the same policy without the four-wave
// kernels of gemm_w4.hip / gemm_fp8_w4.hip (the eight-wave forms that the bit-identity tests compare against).  An explicit API
// call, deliberately not an environment variable.  Every other launch form of rounds 1-2 lives in the lab build only
// (tools/lab/gemm_forms_lab.h, -DP2T_LAB).
static std::atomic<int> g_gemm_policy{0};
void set_gemm_policy(int policy) { g_gemm_policy.store(policy, std::memory_order_relaxed); }
int get_gemm_policy() { return g_gemm_policy.load(std::memory_order_relaxed); }

// Layout of the split-K fix-up workspace: [0, 1024) flag words, [2048, ...) slabs.  The time-out word is NOT in here: it is the
// GPU's sticky fault word (misc.hip fault_word_ptr), which no forward pass re-zeroes.
constexpr size_t kFixHeader = 2048, kFixSlab = 256 * 256 * sizeof(float);
size_t gemm_fix_workspace_bytes() { return kFixHeader + 128 * kFixSlab; }
size_t gemm_fix_header_bytes() { return kFixHeader; }

static bool split_fix(SplitFix& f, void* fix_ws, size_t fix_bytes, unsigned fix_epoch, int64_t n_tiles) {
    if (!fix_ws || fix_bytes < kFixHeader + (size_t)n_tiles * kFixSlab) return false;
    f.flag = (unsigned*)fix_ws;
    f.timeout = fault_word_ptr();
    f.slab = (float*)((char*)fix_ws + kFixHeader);
    f.epoch = fix_epoch;
    return f.timeout != nullptr;
}

#ifdef P2T_LAB
#include "../../tools/lab/gemm_forms_lab.h"
#endif

// The default launch policy.  Every threshold is a measurement (profiles/r01_microbench_v*.log, r02_microbench_w4.log):
//   (1) shapes without edge tiles and at least one whole round of 256 x 256 tiles on the CUs: PERSISTENT kernels (one block per
//       CU walks the tile list; the K loops of consecutive tiles form one stream of stages) -- the four-wave kernel for the
//       epilogues it is built for, else the eight-wave one.  A partial last round runs as split-K pairs when K is long enough
//       for half a K loop to outweigh the slab hand-off (four-wave: K >= 6144 inside the same kernel; eight-wave: the same
//       threshold, else as 128-row halves), and as whole tiles otherwise;
//   (2) smaller grids: three quarters of a round or more -> four-wave kernel, one tile per block; at most half a round with
//       K >= 8192 -> every tile as a split-K pair on two CUs; otherwise the eight-wave per-tile kernel at the tile height
//       (256 / 128 rows) that fills the chip better, with split-K tails when they pay.
template <typename Epi>
static int launch_shape(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover,
                        const EpiParams& ep, int tile, void* fix_ws, size_t fix_bytes, unsigned fix_epoch, hipStream_t s) {
    const int policy = tile ? tile : g_gemm_policy.load(std::memory_order_relaxed);
#ifdef P2T_LAB
    if (policy != 0 && policy != 9) return launch_shape_lab<Epi>(A, lda, W, ldw, M, N, K, n_cover, ep, policy, fix_ws, fix_bytes, fix_epoch, s);
#endif
    const bool no_w4 = policy == 9;
    const int kCUs = cu_count();
    const int ns = K >> 5;
    const int64_t tn = ceil_div(n_cover, 256), tm256 = ceil_div(M, 256), tm128 = ceil_div(M, 128);
    const int64_t items = tm256 * tn, rem = items % kCUs;
    const bool whole_tiles = M % 256 == 0 && N % 256 == 0 && n_cover == N;
    const bool stride32 = (int64_t)256 * (lda > ldw ? lda : ldw) * 2 < ((int64_t)1 << 32);     // four-wave kernels: 32-bit lane offsets
    // ---- (1) persistent kernels
    // whole rounds, or long K, or many rounds (QKV: +13 %), or a read-modify-write of the residual stream (o-proj, 2.5 rounds:
    // with the stream cold in HBM, as inside a step, the undrained stores win)
    if (whole_tiles && items >= kCUs && (ns & 3) == 0 && ns >= 12 && (rem == 0 || ns >= 128 || items >= 4 * kCUs || Epi::kRmw)) {
        const bool pairs_fit = rem > 0 && 2 * rem <= kCUs;
        if constexpr (kHasW4<Epi>) {
            if (!no_w4 && stride32) {
                // FFN-down K = 10240: 752 us as pairs vs 821 as whole tiles; QKV K = 2560: 521 vs 496
                SplitFix f4{};
                const int64_t t4 = pairs_fit && ns >= 192 && split_fix(f4, fix_ws, fix_bytes, fix_epoch, rem) ? rem : 0;
                return launch_gemm_w4_persist<Epi>(A, lda, W, ldw, M, N, K, (int)(items - t4), (int)t4, kCUs, ep, f4, s);
            }
        }
        SplitFix fix{};
        int64_t n_full = items, n_tail = 0;
        int half_tail = 0;
        if (pairs_fit && rem <= 128 && (ns & 7) == 0 && ns >= 192 && split_fix(fix, fix_ws, fix_bytes, fix_epoch, rem)) {
            n_full = items - rem;
            n_tail = rem;
        } else if (pairs_fit && rem <= 128) {           // K too short for split-K to pay: the leftover tiles as 128-row halves
            n_full = items - rem;
            n_tail = rem;
            half_tail = 1;
        }
        gemm_nt_mfma_persist_kernel<Epi><<<dim3(kCUs), 512, 0, s>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, (int)tm256, (int)tn,
                                                                  (int)n_full, (int)n_tail, half_tail, n_cover, ep, fix);
        P2T_LAUNCH_CHECK();
        return P2T_OK;
    }
    // ---- (2) less than a round, or edge tiles
    if constexpr (std::is_same<Epi, EpiQkvRope<bf16_t>>::value || std::is_same<Epi, EpiStore<bf16_t>>::value) {
        // QKV of the text tower at 2 048 tokens (192 tiles): 1 319 TFLOP/s vs 1 163 for the eight-wave per-tile kernel; below that
        // fill, or with the fp32 read-modify-write epilogue, the eight-wave forms stay ahead
        if (!no_w4 && whole_tiles && items < kCUs && items * 4 >= kCUs * 3 && K % 128 == 0 && K >= 256) {
            const int rc = launch_gemm_w4(A, lda, W, ldw, M, N, K, n_cover, P2T_BF16, std::is_same<Epi, EpiQkvRope<bf16_t>>::value ? P2T_EPI_QKV_ROPE : P2T_EPI_STORE, ep, s);
            if (rc != P2T_ERR_UNSUPPORTED) return rc;
        }
    }
    const bool half_round_long_k = items * 2 <= kCUs && items * 8 >= kCUs * 3 && K >= 8192;
    if constexpr (kHasW4Pairs<Epi>) {
        // FFN-down of the text tower at 2 048 tokens (128 tiles, K = 14 336): 185 vs 204 us for the eight-wave pair kernel; at
        // K = 4 096 (o-proj) the slab hand-off costs more than it saves (81 vs 73 us)
        SplitFix f4{};
        if (!no_w4 && whole_tiles && half_round_long_k && K % 128 == 0 && stride32 && split_fix(f4, fix_ws, fix_bytes, fix_epoch, items))
            return launch_gemm_w4_pairs<Epi>(A, lda, W, ldw, M, N, K, (int)items, ep, f4, s);
    }
    SplitFix fix{};
    if (half_round_long_k && split_fix(fix, fix_ws, fix_bytes, fix_epoch, items)) {
        // every tile as two K halves on two CUs (half a tile time + the slab hand-off) instead of a full round of 128-row tiles
        gemm_nt_mfma_tail_kernel<Epi><<<dim3((unsigned)(2 * items)), 512, 0, s>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, (int)tm256,
                                                                                 (int)tn, 0, (int)items, n_cover, ep, fix);
        P2T_LAUNCH_CHECK();
        return P2T_OK;
    }
    const double cost256 = (double)ceil_div(items, kCUs);
    const double cost128 = (double)ceil_div(tm128 * tn, kCUs) * kSmallTileCost * 1.08;
    if (cost256 > cost128) return launch_cfg<4, Epi>(A, lda, W, ldw, M, N, K, n_cover, ep, s);
    // split-K tail: the tiles of a partial last round (at most half a round) as two K halves each; pays for long K (FFN-down
    // +8 %) or many full rounds (QKV +5 %), not at K = 2560 with two full rounds (o-proj)
    const int64_t n_full = (items / kCUs) * kCUs, n_tail = items - n_full;
    if (n_full > 0 && n_tail > 0 && n_tail <= 128 && (ns >= 128 || (ns >= 64 && n_full >= 4 * kCUs)) && split_fix(fix, fix_ws, fix_bytes, fix_epoch, n_tail)) {
        gemm_nt_mfma_tail_kernel<Epi><<<dim3((unsigned)(n_full + 2 * n_tail)), 512, 0, s>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K,
                                                                                            (int)tm256, (int)tn, (int)n_full, (int)n_tail, n_cover, ep, fix);
        P2T_LAUNCH_CHECK();
        return P2T_OK;
    }
    return launch_cfg<8, Epi>(A, lda, W, ldw, M, N, K, n_cover, ep, s);
}

int launch_gemm_mfma(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover,
                     int out_dtype, int epilogue, const EpiParams& ep, int tile, void* fix_ws, size_t fix_bytes,
                     unsigned fix_epoch, hipStream_t s) {
    const bool ob = out_dtype == P2T_BF16;
    switch (epilogue) {
        case P2T_EPI_STORE:
            return ob ? launch_shape<EpiStore<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s)
                      : launch_shape<EpiStore<float>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s);
        case P2T_EPI_GELU:
            if (ep.drop_p > 0.f || ep.z)
                return ob ? launch_shape<EpiGelu<bf16_t, true>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s)
                          : launch_shape<EpiGelu<float, true>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s);
            return ob ? launch_shape<EpiGelu<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s)
                      : launch_shape<EpiGelu<float>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s);
        case P2T_EPI_RESID:
            return launch_shape<EpiResid>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s);
        case P2T_EPI_SWIGLU:
            return ob ? launch_shape<EpiSwiglu<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s)
                      : launch_shape<EpiSwiglu<float>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s);
        case P2T_EPI_STORE_F32:
            return launch_shape<EpiF32>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s);
        case P2T_EPI_GELU_BWD:
            return ob ? launch_shape<EpiGeluBwd<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s)
                      : launch_shape<EpiGeluBwd<float>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s);
        case P2T_EPI_QKV_ROPE:
            return ob ? launch_shape<EpiQkvRope<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s)
                      : launch_shape<EpiQkvRope<float>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, fix_ws, fix_bytes, fix_epoch, s);
    }
    set_error("gemm: unknown epilogue %d", epilogue);
    return P2T_ERR_ARG;
}

}  // namespace p2t
