// out[M, N] = x[M, K] . W[N, K]^T for M <= 64 rows (one decode step: M = rows that generate) -- an HBM-bound weight stream, not a
// matrix-pipe problem: every byte of W is read exactly once per call and nothing else scales with N * K.
//
// Block = 8 waves over 16 * NT output features: the waves split K eight ways (contiguous slices: a wave streams 16 * NT rows x its
// slice, 64-byte pieces per row and instruction), `v_mfma_f32_16x16x32_bf16` with
// A = 16 features of W, B = 16 rows of x (read from global: x is a few hundred KB and lives in L2 / L1), so a lane ends up with 4
// consecutive features of one row.  The eight partial tiles meet in LDS and are added in wave order (no atomics: the result does not
// depend on scheduling).  Grid = N / (16 NT) blocks: 256 for the smallest decoder projection (N = 4096, one block of 8 waves per CU
// with up to 16 loads of 16 bytes in flight per lane), thousands for gate/up and the LM head.
// Epilogues: plain store (model dtype or f32), f32 residual += (o-proj, down-proj), SwiGLU over the 32-row gate/up interleave of
// p2t_llama_layer.gu_w (a block takes 16 gate rows and their 16 up rows).
#include "common.h"
#include "kernels.h"

#include <type_traits>

namespace p2t {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
enum { SK_STORE = 0, SK_STORE_F32 = 1, SK_RESID = 2, SK_SWIGLU = 3, SK_QKV_ROPE = 4 };
constexpr int kSkWaves = 8;

// Weight fragments: non-temporal for the pre-shuffled stream copy (whole lines, read once: +4-6 % measured), plain loads for the
// row-major layout (64-byte pieces: the second half of every line is wanted one step later -- non-temporal costs 5-20 % there).
template <bool NT_LOAD>
__device__ __forceinline__ bf16x8 ld_stream(const bf16_t* p) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (NT_LOAD) return __builtin_bit_cast(bf16x8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)));
    else return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ bf16x8 ld_cached(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

// The tail every form of the kernel shares: the eight partial tiles of a block meet in LDS and are added in wave order (no
// atomics), then the epilogue on 4 consecutive features of one row per thread.
template <int MT, int NT, int EPI>
__device__ __forceinline__ void skinny_finish(float (&red)[kSkWaves][NT][MT][64][4], const f32x4 (&acc)[NT][MT], const int (&row0)[NT], int w, int lane,
                                              void* __restrict__ out, int64_t ldc, int M, int N, const SkinnyRope& ra) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) *reinterpret_cast<f32x4*>(&red[w][i][j][lane][0]) = acc[i][j];
    __syncthreads();
    // the eight partial tiles in wave order; thread -> (n-tile i, m-tile j, lane slot): D[feature 4 g + r][row r16]
    constexpr int kOutTiles = (EPI == SK_SWIGLU || EPI == SK_QKV_ROPE) ? 1 : NT;
    for (int e = threadIdx.x; e < kOutTiles * MT * 64; e += kSkWaves * 64) {
        const int i = e / (MT * 64), j = (e / 64) % MT, l = e & 63;
        const int m = j * 16 + (l & 15), gg = l >> 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][i][j][l][0]);
#pragma unroll
        for (int ww = 1; ww < kSkWaves; ++ww) v += *reinterpret_cast<const f32x4*>(&red[ww][i][j][l][0]);
        if (m >= M) continue;
        if constexpr (EPI == SK_QKV_ROPE) {
            // the QKV + RoPE epilogue of the towers for ONE new token per row (epilogue.h EpiQkvRope; head_dim 64: a 64-row block is a
            // head, partner = channel + 32; head_dim 128: packed rows, block 2 h holds channels (j, j + 64) for j < 32, block 2 h + 1
            // those for 32 <= j < 64): query * q_scale, rotation at position prompt_len + step, then q -> qbuf, k -> the generated
            // key segment at index step, v -> the transposed value segment
            f32x4 u = *reinterpret_cast<const f32x4*>(&red[0][1][j][l][0]);
#pragma unroll
            for (int ww = 1; ww < kSkWaves; ++ww) u += *reinterpret_cast<const f32x4*>(&red[ww][1][j][l][0]);
            const int n = row0[0] + 4 * gg;
            if (n < N) {
                const int hd = ra.d, half = hd >> 1, blk = n >> 6;
                const int head = hd == 64 ? blk : blk >> 1;
                const int ch = (hd == 64 ? 0 : 32 * (blk & 1)) + (n & 31);
                const int step = min(max(ra.step[0], 0), ra.G - 1);
                if (head < ra.nh + ra.nkv) {
                    const bool is_q = head < ra.nh;
                    const float qs = is_q ? ra.q_scale : 1.0f, pos = (float)(ra.prompt_len[m / ra.group] + step);
                    float o1[4], o2[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float a = __fmul_rn(pos, ra.inv_freq[ch + t]);
                        const float c = cosf(a), sn = sinf(a), a1 = v[t] * qs, a2 = u[t] * qs;
                        o1[t] = a1 * c - a2 * sn;
                        o2[t] = a2 * c + a1 * sn;
                    }
                    bf16_t* dst = is_q ? (bf16_t*)ra.q + ((int64_t)m * ra.nh + head) * hd
                                       : (bf16_t*)ra.k + (((int64_t)m * ra.nkv + (head - ra.nh)) * ra.G + step) * hd;
                    store4(dst + ch, o1);
                    store4(dst + ch + half, o2);
                } else {
                    bf16_t* dst = (bf16_t*)ra.vt + ((int64_t)m * ra.nkv + (head - ra.nh - ra.nkv)) * hd * ra.G + step;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        dst[(int64_t)(ch + t) * ra.G] = from_f32<bf16_t>(v[t]);
                        dst[(int64_t)(ch + t + half) * ra.G] = from_f32<bf16_t>(u[t]);
                    }
                }
            }
        } else if constexpr (EPI == SK_SWIGLU) {
            f32x4 u = *reinterpret_cast<const f32x4*>(&red[0][1][j][l][0]);
#pragma unroll
            for (int ww = 1; ww < kSkWaves; ++ww) u += *reinterpret_cast<const f32x4*>(&red[ww][1][j][l][0]);
            const int n = row0[0] + 4 * gg;                          // gate row; feature f = (n / 64) * 32 + n % 32
            if (n < N) {
                const int f = (n >> 6) * 32 + (n & 31);
                float r[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) r[t] = silu_for<bf16_t>(v[t]) * u[t];
                store4((bf16_t*)out + (int64_t)m * ldc + f, r);
            }
        } else {
            const int n = row0[i] + 4 * gg;
            if (n + 3 < N) {
                if constexpr (EPI == SK_STORE) {
                    const float r[4] = {v[0], v[1], v[2], v[3]};
                    store4((bf16_t*)out + (int64_t)m * ldc + n, r);
                } else if constexpr (EPI == SK_STORE_F32) {
                    *reinterpret_cast<f32x4*>((float*)out + (int64_t)m * ldc + n) = v;
                } else {
                    f32x4* o = reinterpret_cast<f32x4*>((float*)out + (int64_t)m * ldc + n);
                    *o = *o + v;
                }
            } else {
                for (int t = 0; t < 4 && n + t < N; ++t) {
                    if constexpr (EPI == SK_STORE) ((bf16_t*)out)[(int64_t)m * ldc + n + t] = from_f32<bf16_t>(v[t]);
                    else if constexpr (EPI == SK_STORE_F32) ((float*)out)[(int64_t)m * ldc + n + t] = v[t];
                    else ((float*)out)[(int64_t)m * ldc + n + t] += v[t];
                }
            }
        }
    }
}

// MT: 16-row tiles of x (M <= 16 MT); NT: 16-feature tiles of W per block (SK_SWIGLU: NT = 2 = gate tile + up tile)
// PRE: W in the pre-shuffled stream order of p2t_preshuffle_w -- [16-row tile][32-deep K step][lane][8]: the 1 KB one MFMA consumes is
// contiguous, a wave's loads are whole cache lines back to back (measured +8..30 % over 64-byte pieces of 16 separate rows).
template <int MT, int NT, int EPI, bool PRE>
__global__ void __launch_bounds__(kSkWaves * 64) gemm_skinny_kernel(const bf16_t* __restrict__ x, int64_t lda, const bf16_t* __restrict__ W,
                                                                   int64_t ldw, void* __restrict__ out, int64_t ldc, int M, int N, int K, SkinnyRope ra) {
    __shared__ float red[kSkWaves][NT][MT][64][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    // rows of W this block owns
    int row0[NT];
    if constexpr (EPI == SK_SWIGLU || EPI == SK_QKV_ROPE) {     // (first group, partner group) of a 64-row block
        const int j = blockIdx.x >> 1, q = blockIdx.x & 1;          // 64-row block j of gu_w: gate rows 64 j + 16 q, up rows + 32
        row0[0] = 64 * j + 16 * q;
        row0[1] = row0[0] + 32;
    } else {
#pragma unroll
        for (int i = 0; i < NT; ++i) row0[i] = (blockIdx.x * NT + i) * 16;
    }
    const int steps = K >> 5;                                       // 32 of K per MFMA
    constexpr int kStepStride = PRE ? 512 : 32;                     // elements between the fragments of consecutive K steps
    const bf16_t* wp[NT];
    bool wok[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if constexpr (PRE) {
            wok[i] = row0[i] < N;                                   // rows are padded to whole tiles with zeros
            wp[i] = W + (int64_t)(wok[i] ? row0[i] >> 4 : 0) * steps * 512 + lane * 8;
        } else {
            wok[i] = row0[i] + r16 < N;
            wp[i] = W + (int64_t)(wok[i] ? row0[i] + r16 : 0) * ldw + g * 8;
        }
    }
    const bf16_t* xp[MT];
    bool xok[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        xok[j] = j * 16 + r16 < M;
        xp[j] = x + (int64_t)(xok[j] ? j * 16 + r16 : 0) * lda + g * 8;
    }
    const int s0 = (int)((int64_t)steps * w / kSkWaves), s1 = (int)((int64_t)steps * (w + 1) / kSkWaves);
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 zero8 = {};
    // UU steps at once: all their loads first (that many 16-byte requests in flight per lane), then the MFMAs
    int s = s0;
    auto batch = [&](auto uu) {
        constexpr int UU = decltype(uu)::value;
        bf16x8 a[UU][NT], b[UU][MT];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
#pragma unroll
            for (int i = 0; i < NT; ++i) a[u][i] = wok[i] ? ld_stream<PRE>(wp[i] + (int64_t)(s + u) * kStepStride) : zero8;
#pragma unroll
            for (int j = 0; j < MT; ++j) b[u][j] = xok[j] ? ld_cached(xp[j] + (int64_t)(s + u) * 32) : zero8;
        }
#pragma unroll
        for (int u = 0; u < UU; ++u)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
        s += UU;
    };
    constexpr int U = NT * MT >= 4 ? 4 : (NT * MT == 1 ? 16 : 8);    // a 4096-deep K is 16 steps per wave: one round trip
    while (s + U <= s1) batch(std::integral_constant<int, U>{});
    if constexpr (U >= 16) { if (s + 8 <= s1) batch(std::integral_constant<int, 8>{}); }
    if constexpr (U >= 8) { if (s + 4 <= s1) batch(std::integral_constant<int, 4>{}); }
    if (s + 2 <= s1) batch(std::integral_constant<int, 2>{});
    if (s < s1) batch(std::integral_constant<int, 1>{});

    skinny_finish<MT, NT, EPI>(red, acc, row0, w, lane, out, ldc, M, N, ra);
}

// ---------------------------------------------------------------------------------------------
// The same stream on e4m3 operands with one E8M0 scale byte per row (quant.hip, DESIGN section 9: what `gemm_fp8` models hold):
// v_mfma_scale_f32_16x16x128_f8f6f4, a K step is 128 BYTES.  Operand registers 0..3 of lane (r16, g) hold the K bytes 16 g .. 16 g + 15
// of row r16, registers 4..7 the bytes 64 + 16 g .. (gemm_fp8.hip, lane maps); the scale byte of a lane's row rides in byte i / j of
// the two scale registers (op_sel).  Row-major W: two 16-byte loads per lane and step = a WHOLE 128-byte line per row; the stream copy
// (launch_preshuffle_fp8): [tile][step][half][lane][16 bytes].
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int MT, int NT, int EPI, bool PRE>
__global__ void __launch_bounds__(kSkWaves * 64) gemm_skinny_fp8_kernel(const uint8_t* __restrict__ x, int64_t lda, const uint8_t* __restrict__ xs,
                                                                       const uint8_t* __restrict__ W, int64_t ldw, const uint8_t* __restrict__ ws,
                                                                       void* __restrict__ out, int64_t ldc, int M, int N, int K, SkinnyRope ra) {
    __shared__ float red[kSkWaves][NT][MT][64][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    int row0[NT];
    if constexpr (EPI == SK_SWIGLU || EPI == SK_QKV_ROPE) {
        const int j = blockIdx.x >> 1, q = blockIdx.x & 1;
        row0[0] = 64 * j + 16 * q;
        row0[1] = row0[0] + 32;
    } else {
#pragma unroll
        for (int i = 0; i < NT; ++i) row0[i] = (blockIdx.x * NT + i) * 16;
    }
    const int steps = K >> 7;                                       // 128 bytes of K per MFMA
    constexpr int kStep = PRE ? 2048 : 128, kHalf = PRE ? 1024 : 64;
    const uint8_t* wp[NT];
    bool wok[NT];
    int sw = 0, sx = 0;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if constexpr (PRE) {
            wok[i] = row0[i] < N;
            wp[i] = W + (int64_t)(wok[i] ? row0[i] >> 4 : 0) * steps * 2048 + lane * 16;
        } else {
            wok[i] = row0[i] + r16 < N;
            wp[i] = W + (int64_t)(wok[i] ? row0[i] + r16 : 0) * ldw + g * 16;
        }
        sw |= (int)ws[row0[i] + r16 < N ? row0[i] + r16 : N - 1] << (8 * i);
    }
    const uint8_t* xp[MT];
    bool xok[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        xok[j] = j * 16 + r16 < M;
        xp[j] = x + (int64_t)(xok[j] ? j * 16 + r16 : 0) * lda + g * 16;
        sx |= (int)xs[xok[j] ? j * 16 + r16 : 0] << (8 * j);
    }
    const int s0 = (int)((int64_t)steps * w / kSkWaves), s1 = (int)((int64_t)steps * (w + 1) / kSkWaves);
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const v4i zero4 = {0, 0, 0, 0};
    auto ld16 = [](const uint8_t* p) { return *reinterpret_cast<const v4i*>(p); };
    auto ld16s = [](const uint8_t* p) {
        if constexpr (PRE) return __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p));
        else return *reinterpret_cast<const v4i*>(p);
    };
    int s = s0;
    auto batch = [&](auto uu) {
        constexpr int UU = decltype(uu)::value;
        v8i a[UU][NT], b[UU][MT];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const uint8_t* p = wp[i] + (int64_t)(s + u) * kStep;
                const v4i lo = wok[i] ? ld16s(p) : zero4, hi = wok[i] ? ld16s(p + kHalf) : zero4;
                a[u][i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const uint8_t* p = xp[j] + (int64_t)(s + u) * 128;
                const v4i lo = xok[j] ? ld16(p) : zero4, hi = xok[j] ? ld16(p + 64) : zero4;
                b[u][j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
#pragma unroll
        for (int u = 0; u < UU; ++u)
            static_for<0, NT>([&](auto ii) {
                static_for<0, MT>([&](auto jj) {
                    constexpr int I = decltype(ii)::value, J = decltype(jj)::value;
                    acc[I][J] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[u][I], b[u][J], acc[I][J], 0, 0, I, sw, J, sx);
                });
            });
        s += UU;
    };
    constexpr int U = NT * MT >= 4 ? 2 : (NT * MT == 1 ? 8 : 4);
    while (s + U <= s1) batch(std::integral_constant<int, U>{});
    if constexpr (U >= 8) { if (s + 4 <= s1) batch(std::integral_constant<int, 4>{}); }
    if constexpr (U >= 4) { if (s + 2 <= s1) batch(std::integral_constant<int, 2>{}); }
    if (s < s1) batch(std::integral_constant<int, 1>{});
    if (s < s1) batch(std::integral_constant<int, 1>{});
    skinny_finish<MT, NT, EPI>(red, acc, row0, w, lane, out, ldc, M, N, ra);
}

template <int MT, int EPI, bool PRE>
int launch_mt_fp8(const uint8_t* x, int64_t lda, const uint8_t* xs, const uint8_t* W, int64_t ldw, const uint8_t* ws, void* out, int64_t ldc, int M, int N,
                  int K, const SkinnyRope& ra, hipStream_t s) {
    if constexpr (EPI == SK_SWIGLU || EPI == SK_QKV_ROPE) {
        gemm_skinny_fp8_kernel<MT, 2, EPI, PRE><<<(unsigned)(N / 32), kSkWaves * 64, 0, s>>>(x, lda, xs, W, ldw, ws, out, ldc, M, N, K, ra);
    } else {
        const int tiles = (N + 15) / 16;
        if (tiles >= 1024 || (MT >= 2 && tiles >= 384))
            gemm_skinny_fp8_kernel<MT, 2, EPI, PRE><<<(unsigned)((tiles + 1) / 2), kSkWaves * 64, 0, s>>>(x, lda, xs, W, ldw, ws, out, ldc, M, N, K, ra);
        else
            gemm_skinny_fp8_kernel<MT, 1, EPI, PRE><<<(unsigned)tiles, kSkWaves * 64, 0, s>>>(x, lda, xs, W, ldw, ws, out, ldc, M, N, K, ra);
    }
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}
template <int EPI>
int launch_epi_fp8(const uint8_t* x, int64_t lda, const uint8_t* xs, const uint8_t* W, int64_t ldw, const uint8_t* ws, void* out, int64_t ldc, int M, int N,
                   int K, int pre, hipStream_t s, const SkinnyRope& ra = SkinnyRope{}) {
#define P2T_SKF(MTV)                                                                                                              \
    (pre ? launch_mt_fp8<MTV, EPI, true>(x, lda, xs, W, ldw, ws, out, ldc, M, N, K, ra, s)                                         \
         : launch_mt_fp8<MTV, EPI, false>(x, lda, xs, W, ldw, ws, out, ldc, M, N, K, ra, s))
    if (M <= 16) return P2T_SKF(1);
    if (M <= 32) return P2T_SKF(2);
    return P2T_SKF(4);
#undef P2T_SKF
}

// out[(((tile * steps + step) * 2 + half) * 64 + lane) * 16 + b] = W[16 tile + lane % 16][128 step + 64 half + 16 (lane / 16) + b]
__global__ void __launch_bounds__(256) preshuffle_fp8_kernel(const uint8_t* __restrict__ W, int64_t ldw, int N, int K, uint8_t* __restrict__ out,
                                                             int64_t chunks) {
    const int steps = K >> 7;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * 256) {
        const int lane = (int)(c & 63), half = (int)((c >> 6) & 1);
        const int64_t ts = c >> 7;
        const int step = (int)(ts % steps);
        const int64_t row = (ts / steps) * 16 + (lane & 15);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < N) v = *reinterpret_cast<const uint4*>(W + row * ldw + 128 * step + 64 * half + 16 * (lane >> 4));
        *reinterpret_cast<uint4*>(out + c * 16) = v;
    }
}

template <int MT, int EPI, bool PRE>
int launch_mt(const bf16_t* x, int64_t lda, const bf16_t* W, int64_t ldw, void* out, int64_t ldc, int M, int N, int K, const SkinnyRope& ra, hipStream_t s) {
    if constexpr (EPI == SK_SWIGLU || EPI == SK_QKV_ROPE) {
        gemm_skinny_kernel<MT, 2, EPI, PRE><<<(unsigned)(N / 32), kSkWaves * 64, 0, s>>>(x, lda, W, ldw, out, ldc, M, N, K, ra);
    } else {
        // feature tiles per block: more of them re-use the x fragments of a step, fewer give the small projections enough blocks
        const int tiles = (N + 15) / 16;
        if (MT <= 2 && tiles >= 4096) gemm_skinny_kernel<MT, 4, EPI, PRE><<<(unsigned)((tiles + 3) / 4), kSkWaves * 64, 0, s>>>(x, lda, W, ldw, out, ldc, M, N, K, ra);
        else if (tiles >= 1024 || (MT >= 2 && tiles >= 384)) gemm_skinny_kernel<MT, 2, EPI, PRE><<<(unsigned)((tiles + 1) / 2), kSkWaves * 64, 0, s>>>(x, lda, W, ldw, out, ldc, M, N, K, ra);
        else gemm_skinny_kernel<MT, 1, EPI, PRE><<<(unsigned)tiles, kSkWaves * 64, 0, s>>>(x, lda, W, ldw, out, ldc, M, N, K, ra);
    }
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

template <int EPI, bool PRE>
int launch_pre(const bf16_t* x, int64_t lda, const bf16_t* W, int64_t ldw, void* out, int64_t ldc, int M, int N, int K, const SkinnyRope& ra, hipStream_t s) {
    if (M <= 16) return launch_mt<1, EPI, PRE>(x, lda, W, ldw, out, ldc, M, N, K, ra, s);
    if (M <= 32) return launch_mt<2, EPI, PRE>(x, lda, W, ldw, out, ldc, M, N, K, ra, s);
    return launch_mt<4, EPI, PRE>(x, lda, W, ldw, out, ldc, M, N, K, ra, s);
}
template <int EPI>
int launch_epi(const bf16_t* x, int64_t lda, const bf16_t* W, int64_t ldw, void* out, int64_t ldc, int M, int N, int K, int pre, hipStream_t s,
               const SkinnyRope& ra = SkinnyRope{}) {
    return pre ? launch_pre<EPI, true>(x, lda, W, ldw, out, ldc, M, N, K, ra, s) : launch_pre<EPI, false>(x, lda, W, ldw, out, ldc, M, N, K, ra, s);
}

// out[((tile * steps + step) * 64 + lane) * 8 + e] = W[16 tile + lane % 16][32 step + 8 (lane / 16) + e], rows >= N as zeros
__global__ void __launch_bounds__(256) preshuffle_kernel(const bf16_t* __restrict__ W, int64_t ldw, int N, int K, bf16_t* __restrict__ out, int64_t chunks) {
    const int steps = K >> 5;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * 256) {
        const int lane = (int)(c & 63);
        const int64_t ts = c >> 6;
        const int step = (int)(ts % steps);
        const int64_t row = (ts / steps) * 16 + (lane & 15);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < N) v = *reinterpret_cast<const uint4*>(W + row * ldw + 32 * step + 8 * (lane >> 4));
        *reinterpret_cast<uint4*>(out + c * 8) = v;
    }
}

}  // namespace

// bf16 only, M <= 64, K % 32 == 0, rows 16-byte aligned; epilogue = P2T_EPI_STORE (out_dtype bf16 or f32) / P2T_EPI_STORE_F32 /
// P2T_EPI_RESID / P2T_EPI_SWIGLU (N = interleaved gate + up rows, a multiple of 64; out bf16 [M, N / 2]).
// -> P2T_ERR_UNSUPPORTED when the shape is not one of these (the caller then takes gemm_nt).
int launch_gemm_skinny(const void* x, int64_t lda, const void* W, int64_t ldw, void* out, int64_t ldc, int64_t M, int64_t N, int64_t K, int dtype,
                       int out_dtype, int epilogue, hipStream_t s, int pre) {
    if (dtype != P2T_BF16 || M < 1 || M > 64 || N < 1 || N >= (1 << 30) || K < 32 || K % 32 || lda % 8 || lda < K || (!pre && (ldw % 8 || ldw < K)))
        return P2T_ERR_UNSUPPORTED;
    const bf16_t* xb = (const bf16_t*)x;
    const bf16_t* wb = (const bf16_t*)W;
    const int m = (int)M, n = (int)N, k = (int)K;
    switch (epilogue) {
        case P2T_EPI_STORE:
            if (out_dtype == P2T_BF16 && ldc % 4 == 0) return launch_epi<SK_STORE>(xb, lda, wb, ldw, out, ldc, m, n, k, pre, s);
            if (out_dtype == P2T_F32 && ldc % 4 == 0) return launch_epi<SK_STORE_F32>(xb, lda, wb, ldw, out, ldc, m, n, k, pre, s);
            return P2T_ERR_UNSUPPORTED;
        case P2T_EPI_STORE_F32:
            return ldc % 4 == 0 ? launch_epi<SK_STORE_F32>(xb, lda, wb, ldw, out, ldc, m, n, k, pre, s) : P2T_ERR_UNSUPPORTED;
        case P2T_EPI_RESID:
            return ldc % 4 == 0 ? launch_epi<SK_RESID>(xb, lda, wb, ldw, out, ldc, m, n, k, pre, s) : P2T_ERR_UNSUPPORTED;
        case P2T_EPI_SWIGLU:
            return (out_dtype == P2T_BF16 && n % 64 == 0 && ldc % 4 == 0) ? launch_epi<SK_SWIGLU>(xb, lda, wb, ldw, out, ldc, m, n, k, pre, s)
                                                                          : P2T_ERR_UNSUPPORTED;
    }
    return P2T_ERR_UNSUPPORTED;
}


// The QKV projection of ONE new token per row with the rotation and the cache append as its epilogue (head_dim 64 / 128, no q/k
// norm: the shapes whose prefill runs the fused QKV + RoPE epilogue).  N = (nh + 2 nkv) * head_dim rows of qkv_w.
int launch_gemm_skinny_qkv_rope(const void* x, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N, int64_t K, const SkinnyRope& ra,
                                hipStream_t s, int pre) {
    if (M < 1 || M > 64 || N % 64 || K < 32 || K % 32 || lda % 8 || lda < K || (!pre && (ldw % 8 || ldw < K)) || (ra.d != 64 && ra.d != 128) ||
        N != (int64_t)(ra.nh + 2 * ra.nkv) * ra.d)
        return P2T_ERR_UNSUPPORTED;
    return launch_epi<SK_QKV_ROPE>((const bf16_t*)x, lda, (const bf16_t*)W, ldw, nullptr, 0, (int)M, (int)N, (int)K, pre, s, ra);
}

// e4m3 form: x / W are e4m3 bytes (K % 128 == 0, zero padded), xs / ws one E8M0 byte per row.  Same epilogues; ra != nullptr = the QKV
// + rotation + cache-append form.
int launch_gemm_skinny_fp8(const void* x, int64_t lda, const uint8_t* xs, const void* W, int64_t ldw, const uint8_t* ws, void* out, int64_t ldc, int64_t M,
                           int64_t N, int64_t K, int out_dtype, int epilogue, const SkinnyRope* ra, hipStream_t s, int pre) {
    if (!xs || !ws || M < 1 || M > 64 || N < 1 || N >= (1 << 30) || K < 128 || K % 128 || lda % 16 || lda < K || (!pre && (ldw % 16 || ldw < K)))
        return P2T_ERR_UNSUPPORTED;
    const uint8_t* xb = (const uint8_t*)x;
    const uint8_t* wb = (const uint8_t*)W;
    const int m = (int)M, n = (int)N, k = (int)K;
    if (ra) {
        if (N % 64 || (ra->d != 64 && ra->d != 128) || N != (int64_t)(ra->nh + 2 * ra->nkv) * ra->d) return P2T_ERR_UNSUPPORTED;
        return launch_epi_fp8<SK_QKV_ROPE>(xb, lda, xs, wb, ldw, ws, nullptr, 0, m, n, k, pre, s, *ra);
    }
    switch (epilogue) {
        case P2T_EPI_STORE:
            if (out_dtype == P2T_BF16 && ldc % 4 == 0) return launch_epi_fp8<SK_STORE>(xb, lda, xs, wb, ldw, ws, out, ldc, m, n, k, pre, s);
            if (out_dtype == P2T_F32 && ldc % 4 == 0) return launch_epi_fp8<SK_STORE_F32>(xb, lda, xs, wb, ldw, ws, out, ldc, m, n, k, pre, s);
            return P2T_ERR_UNSUPPORTED;
        case P2T_EPI_STORE_F32:
            return ldc % 4 == 0 ? launch_epi_fp8<SK_STORE_F32>(xb, lda, xs, wb, ldw, ws, out, ldc, m, n, k, pre, s) : P2T_ERR_UNSUPPORTED;
        case P2T_EPI_RESID:
            return ldc % 4 == 0 ? launch_epi_fp8<SK_RESID>(xb, lda, xs, wb, ldw, ws, out, ldc, m, n, k, pre, s) : P2T_ERR_UNSUPPORTED;
        case P2T_EPI_SWIGLU:
            return (out_dtype == P2T_BF16 && n % 64 == 0 && ldc % 4 == 0) ? launch_epi_fp8<SK_SWIGLU>(xb, lda, xs, wb, ldw, ws, out, ldc, m, n, k, pre, s)
                                                                          : P2T_ERR_UNSUPPORTED;
    }
    return P2T_ERR_UNSUPPORTED;
}

int launch_preshuffle_fp8(const void* W, int64_t ldw, int64_t N, int64_t K, void* out, hipStream_t s) {
    P2T_REQUIRE(W && out && N > 0 && N < (1 << 30) && K >= 128 && K % 128 == 0 && ldw >= K && ldw % 16 == 0, "p2t_preshuffle_w: bad e4m3 arguments (K %% 128, ldw %% 16)");
    const int64_t chunks = round_up(N, 16) * (K / 16);
    const int64_t blocks = ceil_div(chunks, 256);
    preshuffle_fp8_kernel<<<(unsigned)(blocks > 65535 ? 65535 : blocks), 256, 0, s>>>((const uint8_t*)W, ldw, (int)N, (int)K, (uint8_t*)out, chunks);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

int launch_preshuffle(const void* W, int64_t ldw, int64_t N, int64_t K, void* out, hipStream_t s) {
    P2T_REQUIRE(W && out && N > 0 && N < (1 << 30) && K >= 32 && K % 32 == 0 && ldw >= K && ldw % 8 == 0, "p2t_preshuffle_w: bad arguments (K %% 32, ldw %% 8)");
    const int64_t chunks = round_up(N, 16) * (K / 8);
    const int64_t blocks = ceil_div(chunks, 256);
    preshuffle_kernel<<<(unsigned)(blocks > 65535 ? 65535 : blocks), 256, 0, s>>>((const bf16_t*)W, ldw, (int)N, (int)K, (bf16_t*)out, chunks);
    P2T_LAUNCH_CHECK();
    return P2T_OK;
}

}  // namespace p2t
