// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T, fp32 accumulate, fused epilogue.
//
// Both operands are K-contiguous (activations [M][K], PyTorch Linear weights [N][K]), so A and W
// tiles stage and read identically.  Structure (cdna_hip_programming.md section 5):
//   * block tile BM x BN x 64, WM x WN wavefronts, each owning (MT*16) x (NT*16) outputs as
//     v_mfma_f32_16x16x32_bf16 accumulators (the 16x16x32 shape holds the higher clock, microarch
//     DVFS note 7);
//   * global -> LDS by `global_load_lds_dwordx4` (no VGPR round trip), double-buffered;
//     the LDS image is lane-linear, so the bank-conflict swizzle (16-B chunk c of row r stored at
//     chunk c ^ (r & 7)) is applied to the per-lane SOURCE address and again on the ds_read_b128
//     (rule 21) -- conflict-free for the 16x16x32 fragment read pattern;
//   * operands are passed swapped (W fragment as MFMA "A", activation fragment as "B") so a lane
//     ends up with 4 consecutive output COLUMNS of one row: vector bias loads, 8/16-byte stores,
//     and the SwiGLU gate/up partner (+16 columns) in the same lane;
//   * 1-D grid with an XCD-aware, grouped tile order: the 8 XCDs get contiguous chunks of the
//     tile list and consecutive tiles share activation row-panels (L2 reuse, technique T1).
// Requirements: K % 64 == 0 (callers pad K with zeros), lda/ldw % 8 == 0, 16-byte aligned bases.
#include "common.h"
#include "epilogue.h"
#include "kernels.h"

namespace p2t {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

template <int WM, int WN, int MT, int NT, typename Epi>
__global__ void __launch_bounds__(WM* WN * 64)
    gemm_nt_mfma_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W, int64_t ldw, int64_t M,
                        int N, int K, int tiles_m, int tiles_n, int n_cover, EpiParams ep) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, NTHREADS = WM * WN * 64;
    constexpr int RPP = NTHREADS / 8;                       // rows staged per pass of all waves
    constexpr int A_PASSES = BM / RPP, W_PASSES = BN / RPP;
    constexpr int STAGE = (BM + BN) * 128;                  // bytes per K-tile (64 bf16 = 128 B per row)
    static_assert(BM % RPP == 0 && BN % RPP == 0 && NT % 2 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    // ---- tile coordinates: XCD chunking (bijective) + grouped (GM row-tiles) order ----
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    constexpr int GM = 4;
    const int band = swz / (GM * tiles_n), first_m = band * GM;
    const int gm = min(GM, tiles_m - first_m);
    const int in_band = swz - band * GM * tiles_n;
    const int tm = first_m + in_band % gm, tn = in_band / gm;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w / WN, wn = w % WN;

    // ---- staging addresses: lane -> (row in pass, 16-B slot); source chunk = slot ^ (row & 7) ----
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const bf16_t* a_src[A_PASSES];
    const bf16_t* w_src[W_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        int64_t row = m0 + i * RPP + w * 8 + srow;
        row = row < M ? row : M - 1;
        a_src[i] = A + row * lda + schunk * 8;
    }
#pragma unroll
    for (int i = 0; i < W_PASSES; ++i) {
        int row = n0 + i * RPP + w * 8 + srow;
        row = row < N ? row : N - 1;
        w_src[i] = W + (int64_t)row * ldw + schunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE + w * 1024;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + (int64_t)kt * 64), (lptr_t)(base + i * RPP * 128), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < W_PASSES; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + (int64_t)kt * 64),
                                             (lptr_t)(base + BM * 128 + i * RPP * 128), 16, 0, 0);
    };

    // ---- fragment read offsets (bytes within a stage) ----
    const int fr = lane & 15, kg = lane >> 4;
    const int sw0 = ((0 * 4 + kg) ^ (fr & 7)) << 4;         // k-step 0 chunk, swizzled
    const int sw1 = ((1 * 4 + kg) ^ (fr & 7)) << 4;         // k-step 1
    const int x_off = (wm * MT * 16 + fr) * 128;
    const int w_off = BM * 128 + (wn * NT * 16 + fr) * 128;

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K >> 6;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();          // tile kt is in LDS for every wave; everyone is done reading the other buffer
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char* sb = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = ks ? sw1 : sw0;
            bf16x8 xf[MT], wf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + w_off + j * 2048 + sw);
#pragma unroll
            for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(sb + x_off + j * 2048 + sw);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns row m, columns n .. n+3 of each 16x16 tile ----
    const int ncol = n0 + wn * NT * 16 + kg * 4;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int64_t m = m0 + wm * MT * 16 + j * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < NT; i += 2) {
            const int n = ncol + i * 16;
            if (n >= n_cover) continue;
            const float v0[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            const float v1[4] = {acc[i + 1][j][0], acc[i + 1][j][1], acc[i + 1][j][2], acc[i + 1][j][3]};
            Epi::apply2(ep, m, n, v0, v1);
        }
    }
}

template <int WM, int WN, int MT, int NT, typename Epi>
static int launch_cfg(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover,
                      const EpiParams& ep, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    const int tiles_m = (int)ceil_div(M, BM), tiles_n = (int)ceil_div(n_cover, BN);
    gemm_nt_mfma_kernel<WM, WN, MT, NT, Epi><<<dim3((unsigned)(tiles_m * tiles_n)), WM * WN * 64, 0, s>>>(
        (const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, tiles_m, tiles_n, n_cover, ep);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

template <typename Epi>
static int launch_shape(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover,
                        const EpiParams& ep, int tile, hipStream_t s) {
    // 256x256 fills the chip once there are >= ~256 tiles; otherwise 128x128 (2 blocks per CU).
    const int64_t big_tiles = ceil_div(M, 256) * ceil_div(n_cover, 256);
    if (tile == 256 || (tile == 0 && big_tiles >= 192))
        return launch_cfg<2, 4, 8, 4, Epi>(A, lda, W, ldw, M, N, K, n_cover, ep, s);
    return launch_cfg<2, 2, 4, 4, Epi>(A, lda, W, ldw, M, N, K, n_cover, ep, s);
}

int launch_gemm_mfma(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover,
                     int out_dtype, int epilogue, const EpiParams& ep, int tile, hipStream_t s) {
    const bool ob = out_dtype == P2T_BF16;
    switch (epilogue) {
        case P2T_EPI_STORE:
            return ob ? launch_shape<EpiStore<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, s)
                      : launch_shape<EpiStore<float>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, s);
        case P2T_EPI_GELU:
            return ob ? launch_shape<EpiGelu<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, s)
                      : launch_shape<EpiGelu<float>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, s);
        case P2T_EPI_RESID:
            return launch_shape<EpiResid>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, s);
        case P2T_EPI_SWIGLU:
            return ob ? launch_shape<EpiSwiglu<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, s)
                      : launch_shape<EpiSwiglu<float>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, s);
        case P2T_EPI_STORE_F32:
            return launch_shape<EpiF32>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, s);
        case P2T_EPI_GELU_BWD:
            return ob ? launch_shape<EpiGeluBwd<bf16_t>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, s)
                      : launch_shape<EpiGeluBwd<float>>(A, lda, W, ldw, M, N, K, n_cover, ep, tile, s);
    }
    set_error("gemm: unknown epilogue %d", epilogue);
    return P2T_ERR_ARG;
}

}  // namespace p2t
