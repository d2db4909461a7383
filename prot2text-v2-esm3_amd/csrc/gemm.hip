// GEMM and attention dispatch: argument checks, kernel choice (MFMA vs fp32-FMA), C entry points.
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"
#include "epilogue.h"
#include "kernels.h"

namespace p2t {

// ---- optional in-library timing of the MFMA kernels (bench.py's live roofline measurement) ----------
// HIP events are recorded on the launch stream right before / after each launch, so the elapsed time is
// that kernel's own duration on the GPU; nothing synchronises until p2t_prof_collect.
namespace {
struct ProfRec { hipEvent_t a, b; int cls; double flops; };
// Process-global measurement state (the library's only mutable global besides the launch-policy word and the
// thread-local error string): guarded by a mutex, off by default -- the unlocked fast path is one relaxed load.
struct Prof {
    std::atomic<bool> on{false};
    std::mutex mu;
    std::vector<ProfRec> pool;
    size_t used = 0;
} g_prof;
constexpr size_t kProfMax = 1 << 15;

int prof_begin(hipStream_t s, int cls, double flops) {
    if (!g_prof.on.load(std::memory_order_relaxed)) return -1;
    std::lock_guard<std::mutex> lock(g_prof.mu);
    if (g_prof.used >= kProfMax) return -1;
    if (g_prof.used == g_prof.pool.size()) {
        ProfRec r{};
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
        g_prof.pool.push_back(r);
    }
    ProfRec& r = g_prof.pool[g_prof.used];
    r.cls = cls; r.flops = flops;
    if (hipEventRecord(r.a, s) != hipSuccess) return -1;
    return (int)g_prof.used++;
}
void prof_end(hipStream_t s, int idx) {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lock(g_prof.mu);
    (void)hipEventRecord(g_prof.pool[idx].b, s);
}
}  // namespace

int gemm_nt(const GemmArgs& a, hipStream_t s) {
    P2T_REQUIRE(a.A && a.W && (a.out || a.epilogue == P2T_EPI_QKV_ROPE), "gemm_nt: null operand");
    P2T_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0 && a.N < (1 << 30) && a.K < (1 << 30), "gemm_nt: bad sizes M=%lld N=%lld K=%lld",
                (long long)a.M, (long long)a.N, (long long)a.K);
    if (a.M == 0) return P2T_OK;
    const bool swiglu = a.epilogue == P2T_EPI_SWIGLU;
    const bool rope = a.epilogue == P2T_EPI_QKV_ROPE;
    P2T_REQUIRE(a.N % ((swiglu || rope) ? 64 : 16) == 0, "gemm_nt: N=%lld must be a multiple of %d", (long long)a.N, (swiglu || rope) ? 64 : 16);
    P2T_REQUIRE(!rope || (a.cs && a.q && a.k && a.v && a.seq > 0 && (a.head_dim == 64 || a.head_dim == 128) &&
                          a.N == (int64_t)(a.nh + 2 * a.nkv) * a.head_dim && a.M % a.seq == 0),
                "gemm_nt: EPI_QKV_ROPE needs head_dim 64 or 128 outputs, the rotary table and M = B * seq");
    P2T_REQUIRE(!rope || (a.M < ((int64_t)1 << 31) && a.M * (int64_t)(a.nh > a.nkv ? a.nh : a.nkv) < ((int64_t)1 << 31)),
                "gemm_nt: EPI_QKV_ROPE indexes rows of the head-split outputs in 32 bits (M * heads = %lld)", (long long)(a.M * (a.nh > a.nkv ? a.nh : a.nkv)));
    if (a.dtype == P2T_FP8)
        P2T_REQUIRE(a.a_scale && a.w_scale && a.K % 128 == 0 && a.lda % 16 == 0 && a.ldw % 16 == 0 && (uintptr_t)a.A % 16 == 0 &&
                        (uintptr_t)a.W % 16 == 0 && a.epilogue != P2T_EPI_GELU_BWD && (a.epilogue != P2T_EPI_GELU_FP8 || a.out_row_scale),
                    "gemm_nt (fp8): needs both row-scale arrays, K %% 128 == 0 (zero padded), 16-byte aligned rows (K=%lld lda=%lld ldw=%lld)",
                    (long long)a.K, (long long)a.lda, (long long)a.ldw);
    P2T_REQUIRE(a.K % 4 == 0 && a.lda % 4 == 0 && a.ldw % 4 == 0 && a.ldc % 4 == 0 && a.lda >= a.K && a.ldw >= a.K,
                "gemm_nt: K and the row strides must be multiples of 4 (K=%lld lda=%lld ldw=%lld ldc=%lld)", (long long)a.K,
                (long long)a.lda, (long long)a.ldw, (long long)a.ldc);
    const int n_out = swiglu ? (int)a.N / 2 : (int)a.N;
    P2T_REQUIRE(rope || a.ldc >= n_out, "gemm_nt: ldc=%lld < %d output columns", (long long)a.ldc, n_out);
    int out_dtype = a.out_dtype;
    if (a.epilogue == P2T_EPI_RESID || a.epilogue == P2T_EPI_STORE_F32) out_dtype = P2T_F32;
    P2T_REQUIRE(a.epilogue != P2T_EPI_GELU_BWD || a.z, "gemm_nt: EPI_GELU_BWD needs z");

    int n_zero = n_out;
    if (a.epilogue == P2T_EPI_GELU_FP8) {
        P2T_REQUIRE(a.dtype == P2T_FP8 && a.out_row_scale && a.ldc % 8 == 0, "gemm_nt: EPI_GELU_FP8 needs the fp8 kernel, the output row scales and ldc %% 8 == 0");
        n_zero = (int)(round_up(n_out, 128) < a.ldc ? round_up(n_out, 128) : a.ldc);
    }
    if (a.epilogue == P2T_EPI_STORE || a.epilogue == P2T_EPI_GELU || swiglu || a.epilogue == P2T_EPI_GELU_BWD) {
        n_zero = a.n_zero >= 0 ? a.n_zero : (int)(round_up(n_out, 64) < a.ldc ? round_up(n_out, 64) : a.ldc);
        if (n_zero < n_out) n_zero = n_out;
    }
    const int n_cover = swiglu ? (int)(a.N > 2 * n_zero ? a.N : 2 * n_zero) : (int)(a.N > n_zero ? a.N : n_zero);

    EpiParams ep;
    ep.bias = a.bias; ep.out = a.out; ep.z = a.z; ep.ldc = a.ldc; ep.M = a.M; ep.N = (int)a.N; ep.n_zero = n_zero;
    ep.accumulate = a.accumulate; ep.drop_p = a.drop_p; ep.drop_scale = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    ep.drop_seed = a.drop_seed;
    ep.row_scale = a.out_row_scale;
    ep.cs = a.cs; ep.q = a.q; ep.k = a.k; ep.v = a.v; ep.seq = a.seq; ep.nh = a.nh; ep.nkv = a.nkv; ep.q_scale = a.q_scale; ep.head_dim = a.head_dim;

    if (a.dtype == P2T_FP8) {
        const int pi = prof_begin(s, 2, 2.0 * (double)a.M * (double)a.N * (double)a.K);
        const int rc = launch_gemm_fp8(a.A, a.lda, a.a_scale, a.W, a.ldw, a.w_scale, a.M, (int)a.N, (int)a.K, n_cover, out_dtype, a.epilogue, ep,
                                       a.tile, s);
        prof_end(s, pi);
        return rc;
    }
    const bool aligned = ((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.W % 16 == 0);
    const bool can_mfma = a.dtype == P2T_BF16 && a.K % 64 == 0 && a.lda % 8 == 0 && a.ldw % 8 == 0 && aligned;
    if (a.use_mfma == 1 && !can_mfma) {
        set_error("gemm_nt: MFMA kernel required but shape/dtype not eligible (dtype=%d K=%lld lda=%lld ldw=%lld)", a.dtype,
                  (long long)a.K, (long long)a.lda, (long long)a.ldw);
        return P2T_ERR_UNSUPPORTED;
    }
    if (can_mfma && a.use_mfma != 0) {
        const int pi = prof_begin(s, 0, 2.0 * (double)a.M * (double)a.N * (double)a.K);
        const int rc = launch_gemm_mfma(a.A, a.lda, a.W, a.ldw, a.M, (int)a.N, (int)a.K, n_cover, out_dtype, a.epilogue, ep, a.tile, a.fix_ws,
                                        a.fix_bytes, a.fix_epoch, s);
        prof_end(s, pi);
        return rc;
    }
    return launch_gemm_simple(a.A, a.lda, a.W, a.ldw, a.M, (int)a.N, (int)a.K, n_cover, a.dtype, out_dtype, a.epilogue, ep, s);
}

int attention(const void* q, const void* k, const void* v, const uint8_t* key_mask, const int32_t* kv_info, void* out,
              int64_t ld_out, int B, int T, int nh, int nkv, int d, int dp, float scale, int causal, int dtype, int use_mfma,
              int log2_scores, hipStream_t s, float* lse) {
    P2T_REQUIRE(q && k && v && key_mask && kv_info && out && B > 0 && T > 0 && nh > 0 && nkv > 0, "attention: bad arguments");
    P2T_REQUIRE(ld_out >= (int64_t)nh * d, "attention: ld_out too small");
    if (dtype == P2T_BF16 && use_mfma != 0) {
        const int pi = prof_begin(s, 1, 4.0 * B * nh * (double)T * T * d * (causal ? 0.5 : 1.0));
        // use_mfma: 2 = the general kernel (attn_mfma.hip) even where the hand-placed one applies; 3 = require the hand-placed one
        // (a forward that also returns the log-sum-exps -- the stage-2 training forward -- stays on the general kernel unless the
        // hand-placed one is asked for: its row sums are taken over the bf16-rounded probabilities, 3e-4 off the fp32 sums the exact
        // backward rebuilds P from; the frozen towers of the contrastive step never ask for them)
        const bool hand = use_mfma != 2 && (lse == nullptr || use_mfma == 3) && attn_fwd64_eligible(ld_out, T, nh, nkv, d, dp, log2_scores);
        if (use_mfma == 3 && !hand) {
            set_error("attention: the hand-placed kernel needs head_dim padded to 64, d %% 8 == 0 and log2_scores (d=%d dp=%d)", d, dp);
            return P2T_ERR_UNSUPPORTED;
        }
        const int rc = hand ? launch_attn_fwd64(q, k, v, key_mask, kv_info, out, ld_out, B, T, nh, nkv, d, causal, lse, s)
                            : launch_attn_mfma(q, k, v, key_mask, kv_info, out, ld_out, B, T, nh, nkv, d, dp, scale, causal, log2_scores, lse, s);
        prof_end(s, pi);
        return rc;
    }
    P2T_REQUIRE(use_mfma <= 0, "attention: MFMA kernel needs bf16");
    // exp(ln 2 * (s - m)) = 2^(s - m)
    return launch_attn_simple(q, k, v, key_mask, kv_info, out, ld_out, B, T, nh, nkv, d, dp, log2_scores ? kLn2 : scale, causal, dtype, lse, s);
}

}  // namespace p2t

using namespace p2t;

extern "C" int p2t_prof_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    g_prof.on.store(on != 0, std::memory_order_relaxed);
    g_prof.used = 0;
    return P2T_OK;
}

#ifdef P2T_LAB
namespace p2t { void set_cu_override(int n); }
#endif
extern "C" int p2t_set_gemm_policy(int policy) {
#ifdef P2T_LAB
    if (policy >= 1000 && policy <= 1256) {          // lab: the launch policy counts (policy - 1000) compute units (0 = the device's own count)
        p2t::set_cu_override(policy - 1000);
        return P2T_OK;
    }
    P2T_REQUIRE(policy == 0 || policy == 1 || policy == 2 || policy == 3 || policy == 4 || policy == 5 || policy == 6 || policy == 7 || policy == 8 || policy == 9 || policy == 10 || policy == 12 || policy == 128 || policy == 256,
                "p2t_set_gemm_policy (lab build): unknown policy %d", policy);
#else
    P2T_REQUIRE(policy == 0 || policy == 9, "p2t_set_gemm_policy: policy %d is not in the product library (0 = default, 9 = without the four-wave kernels; "
                "the other launch forms are in the lab build, tools/lab/)", policy);
#endif
    set_gemm_policy(policy);
    return P2T_OK;
}

/* 1 in the lab build (-DP2T_LAB: every launch form of rounds 1-2 selectable), 0 in the product library. */
extern "C" int p2t_is_lab_build(void) {
#ifdef P2T_LAB
    return 1;
#else
    return 0;
#endif
}

extern "C" int p2t_prof_collect(double* ms, int64_t* launches, double* flops, int n_classes) {
    P2T_REQUIRE(ms && launches && flops && n_classes >= 2 && n_classes <= 8, "p2t_prof_collect: bad arguments");
    std::lock_guard<std::mutex> lock(g_prof.mu);
    for (int c = 0; c < n_classes; ++c) { ms[c] = 0.0; launches[c] = 0; flops[c] = 0.0; }
    for (size_t i = 0; i < g_prof.used; ++i) {
        ProfRec& r = g_prof.pool[i];
        P2T_CHECK_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        P2T_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
        if (r.cls < n_classes) { ms[r.cls] += t; launches[r.cls] += 1; flops[r.cls] += r.flops; }
    }
    g_prof.used = 0;
    return P2T_OK;
}

extern "C" int p2t_gemm_nt(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldc,
                           void* z, int64_t M, int64_t N, int64_t K, int dtype, int out_dtype, int epilogue, int accumulate,
                           int use_mfma, void* fix_ws, size_t fix_ws_bytes, unsigned fix_epoch, p2t_stream stream) {
    GemmArgs a{A, lda, W, ldw, bias, out, ldc, z, M, N, K, dtype, out_dtype, epilogue, accumulate, use_mfma, -1, 0.f, 0, 0};
    a.fix_ws = fix_ws; a.fix_bytes = fix_ws_bytes; a.fix_epoch = fix_epoch;
    return gemm_nt(a, (hipStream_t)stream);
}

extern "C" size_t p2t_gemm_fix_workspace_bytes(void) { return gemm_fix_workspace_bytes(); }

extern "C" int p2t_gemm_nt_fp8(const void* A, int64_t lda, const uint8_t* a_scale, const void* W, int64_t ldw, const uint8_t* w_scale,
                               const float* bias, void* out, int64_t ldc, void* z, int64_t M, int64_t N, int64_t K, int out_dtype,
                               int epilogue, int accumulate, int tile, const uint8_t* out_row_scale, p2t_stream stream) {
    GemmArgs a{A, lda, W, ldw, bias, out, ldc, z, M, N, K, P2T_FP8, out_dtype, epilogue, accumulate, 1, -1, 0.f, 0, tile};
    a.a_scale = a_scale; a.w_scale = w_scale; a.out_row_scale = out_row_scale;
    return gemm_nt(a, (hipStream_t)stream);
}

extern "C" int p2t_gemm_qkv_rope(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int64_t M, int64_t K,
                                 int dtype, const float* inv_freq, float* cos_sin_scratch, void* q, void* k, void* v, int seq,
                                 int nh, int nkv, int head_dim, float q_scale, int use_mfma, void* fix_ws, size_t fix_ws_bytes,
                                 unsigned fix_epoch, p2t_stream stream) {
    P2T_REQUIRE(inv_freq && cos_sin_scratch && seq > 0 && nh > 0 && nkv > 0, "p2t_gemm_qkv_rope: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    P2T_TRY(launch_rope_table(inv_freq, seq, head_dim / 2, cos_sin_scratch, s));
    GemmArgs a{A, lda, W, ldw, bias, nullptr, 0, nullptr, M, (int64_t)(nh + 2 * nkv) * head_dim, K, dtype, dtype, P2T_EPI_QKV_ROPE,
               0, use_mfma, -1, 0.f, 0, 0};
    a.cs = cos_sin_scratch; a.q = q; a.k = k; a.v = v; a.seq = seq; a.nh = nh; a.nkv = nkv; a.q_scale = q_scale; a.head_dim = head_dim;
    a.fix_ws = fix_ws; a.fix_bytes = fix_ws_bytes; a.fix_epoch = fix_epoch;
    return gemm_nt(a, s);
}

extern "C" int p2t_mask_prepare(const int64_t* ids, const int64_t* mask, int B, int T, int mask_id, int token_dropout,
                                uint8_t* key_mask, int32_t* kv_info, float* emb_scale, p2t_stream stream) {
    P2T_REQUIRE(mask && key_mask && kv_info && B > 0 && T > 0, "p2t_mask_prepare: bad arguments");
    return launch_mask_prepare(ids, mask, B, T, mask_id, token_dropout, key_mask, kv_info, emb_scale, (hipStream_t)stream);
}

extern "C" int p2t_qkv_post(const void* qkv, int64_t ldq, const float* inv_freq, float* cos_sin_scratch, void* q, void* k,
                            void* v, int B, int T, int nh, int nkv, int d, int dp, float q_scale, int dtype, p2t_stream stream) {
    P2T_REQUIRE(qkv && inv_freq && cos_sin_scratch && q && k && v, "p2t_qkv_post: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    P2T_TRY(launch_rope_table(inv_freq, T, d / 2, cos_sin_scratch, s));
    return launch_qkv_post(qkv, ldq, cos_sin_scratch, q, k, v, B, T, nh, nkv, d, dp, q_scale, dtype, s);
}

extern "C" int p2t_attention(const void* q, const void* k, const void* v, const uint8_t* key_mask, const int32_t* kv_info,
                             void* out, int64_t ld_out, int B, int T, int nh, int nkv, int d, int dp, float scale, int causal,
                             int dtype, int use_mfma, int log2_scores, float* lse, p2t_stream stream) {
    return attention(q, k, v, key_mask, kv_info, out, ld_out, B, T, nh, nkv, d, dp, scale, causal, dtype, use_mfma, log2_scores,
                     (hipStream_t)stream, lse);
}

// The decode step's GEMM on its own (gemm_skinny.hip): see include/p2t_hip.h
extern "C" int p2t_preshuffle_w(const void* W, int64_t ldw, int64_t N, int64_t K, void* out, p2t_stream stream) {
    return p2t::launch_preshuffle(W, ldw, N, K, out, (hipStream_t)stream);
}

extern "C" int p2t_gemm_nt_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, int w_preshuffled, void* out, int64_t ldc, int64_t M, int64_t N,
                                  int64_t K, int out_dtype, int epilogue, p2t_stream stream) {
    P2T_REQUIRE(A && W && out, "p2t_gemm_nt_skinny: null argument");
    const int r = p2t::launch_gemm_skinny(A, lda, W, ldw, out, ldc, M, N, K, P2T_BF16, out_dtype, epilogue, (hipStream_t)stream, w_preshuffled != 0);
    if (r == P2T_ERR_UNSUPPORTED) p2t::set_error("p2t_gemm_nt_skinny: unsupported shape / epilogue (M %lld, N %lld, K %lld, epilogue %d)", (long long)M,
                                                 (long long)N, (long long)K, epilogue);
    return r;
}

extern "C" int p2t_preshuffle_w_fp8(const void* W, int64_t ldw, int64_t N, int64_t K, void* out, p2t_stream stream) {
    return p2t::launch_preshuffle_fp8(W, ldw, N, K, out, (hipStream_t)stream);
}

extern "C" int p2t_gemm_nt_skinny_fp8(const void* A, int64_t lda, const uint8_t* a_scale, const void* W, int64_t ldw, const uint8_t* w_scale,
                                      int w_preshuffled, void* out, int64_t ldc, int64_t M, int64_t N, int64_t K, int out_dtype, int epilogue,
                                      p2t_stream stream) {
    P2T_REQUIRE(A && a_scale && W && w_scale && out, "p2t_gemm_nt_skinny_fp8: null argument");
    const int r = p2t::launch_gemm_skinny_fp8(A, lda, a_scale, W, ldw, w_scale, out, ldc, M, N, K, out_dtype, epilogue, nullptr, (hipStream_t)stream,
                                              w_preshuffled != 0);
    if (r == P2T_ERR_UNSUPPORTED) p2t::set_error("p2t_gemm_nt_skinny_fp8: unsupported shape / epilogue (M %lld, N %lld, K %lld, epilogue %d)", (long long)M,
                                                 (long long)N, (long long)K, epilogue);
    return r;
}
