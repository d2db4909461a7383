// Internal launchers shared between translation units (not part of the C ABI).
#pragma once
#include "common.h"

namespace p2t {

struct EpiParams;

size_t colsum_scratch_bytes(int64_t cols);
int launch_colsum(const void* x, int dtype, int64_t rows, int64_t cols, int64_t ld, float* out, int accumulate, float* scratch,
                  hipStream_t s);

int launch_layernorm(const float* x, int64_t ld_x, const float* w, const float* b, float eps, void* y, int64_t ld_y,
                     int64_t rows, int64_t cols, int out_dtype, hipStream_t s);
int launch_rmsnorm(const float* x, int64_t ld_x, const float* w, float eps, void* y, int64_t ld_y, int64_t rows,
                   int64_t cols, int out_dtype, hipStream_t s);
int launch_rmsnorm_few_rows(const float* x, int64_t ld_x, const float* w, float eps, void* y, int64_t ld_y, int64_t rows, int64_t cols,
                            int out_dtype, hipStream_t s);     // decode step: block per row (norm.hip)
int launch_l2norm(const void* x, int in_dtype, int64_t ld_x, void* y, int out_dtype, int64_t ld_y, float* inv_norm,
                  int64_t rows, int64_t cols, float eps, hipStream_t s);

int launch_mask_prepare(const int64_t* ids, const int64_t* mask, int B, int T, int mask_id, int token_dropout,
                        uint8_t* key_mask, int32_t* kv_info, float* emb_scale, hipStream_t s);
int launch_esm_embed(const int64_t* ids, const int64_t* mask, const void* table, int dtype, const float* emb_scale, int T,
                     int H, int vocab, int mask_id, int token_dropout, float* x, int64_t M, hipStream_t s);
int launch_llama_embed(const int64_t* ids, const void* table, int dtype, int H, int vocab, float* x, int64_t M, hipStream_t s);
int launch_inv_freq(float* inv_freq, int half, float theta, int llama3, float factor, float low_ff, float high_ff,
                    float orig_max_pos, hipStream_t s);
int launch_rope_table(const float* inv_freq, int T, int half, float* cs, hipStream_t s);
int launch_qkv_post(const void* qkv, int64_t ldq, const float* cs, void* q, void* k, void* v, int B, int T, int nh, int nkv,
                    int d, int dp, float q_scale, int dtype, hipStream_t s);

int launch_qk_norm_rope(const void* qkv, int64_t ldq, const float* cs, const float* q_norm_w, const float* k_norm_w, float eps, void* q,
                        void* k, void* v, int B, int T, int nh, int nkv, int d, int dp, float q_scale, int dtype, hipStream_t s);

int launch_gemm_simple(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover, int dtype,
                       int out_dtype, int epilogue, const EpiParams& ep, hipStream_t s);
int launch_gemm_mfma(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int N, int K, int n_cover,
                     int out_dtype, int epilogue, const EpiParams& ep, int tile, void* fix_ws, size_t fix_bytes,
                     unsigned fix_epoch, hipStream_t s);
// fp8 path (quant.hip, gemm_fp8.hip)
int launch_quant_rows(const void* x, int dtype, int64_t ld_x, int64_t rows, int64_t cols, void* q, int64_t ld_q, uint8_t* scale,
                      hipStream_t s);
int launch_layernorm_fp8(const float* x, int64_t ld_x, const float* w, const float* b, float eps, void* q, int64_t ld_q,
                         uint8_t* scale, int64_t rows, int64_t cols, float bound_w, float bound_b, uint8_t* bound_scale, hipStream_t s);
int launch_rmsnorm_fp8(const float* x, int64_t ld_x, const float* w, float eps, void* q, int64_t ld_q, uint8_t* scale, int64_t rows,
                       int64_t cols, hipStream_t s);
// the decode step's forms for a few rows (block per row, one round trip; quant.hip)
int launch_quant_rows_few(const void* x, int dtype, int64_t ld_x, int64_t rows, int64_t cols, void* q, int64_t ld_q, uint8_t* scale, hipStream_t s);
int launch_rmsnorm_fp8_few(const float* x, int64_t ld_x, const float* w, float eps, void* q, int64_t ld_q, uint8_t* scale, int64_t rows, int64_t cols,
                           hipStream_t s);
int launch_gemm_fp8(const void* A, int64_t lda, const uint8_t* a_scale, const void* W, int64_t ldw, const uint8_t* w_scale, int64_t M, int N,
                    int K, int n_cover, int out_dtype, int epilogue, const EpiParams& ep, int tile, hipStream_t s);
unsigned* fault_word_ptr();            // the GPU's sticky fault word (misc.hip); nullptr if the symbol cannot be resolved
int get_gemm_policy();
void set_gemm_policy(int policy);       // launch-form override of the MFMA GEMM (tests / experiments), 0 = default
size_t gemm_fix_workspace_bytes();       // split-K tail fix-up workspace (flags + slabs); header must be zeroed once per
size_t gemm_fix_header_bytes();          // sequence of launches that use distinct epochs
// full dispatcher behind p2t_gemm_nt (gemm.hip)
struct GemmArgs {
    const void* A; int64_t lda; const void* W; int64_t ldw; const float* bias; void* out; int64_t ldc; void* z;
    int64_t M; int64_t N; int64_t K; int dtype; int out_dtype; int epilogue; int accumulate; int use_mfma;
    int n_zero;                 // -1: default (next multiple of 64, clipped to ldc)
    float drop_p; uint64_t drop_seed;
    int tile;                   // 0 auto, 128, 256 (rows of the MFMA block tile)
    // P2T_EPI_QKV_ROPE only (head_dim 64 or 128): rotary table [T, head_dim], outputs [B, heads, T, head_dim]
    const float* cs = nullptr; void* q = nullptr; void* k = nullptr; void* v = nullptr;
    int seq = 0, nh = 0, nkv = 0; float q_scale = 1.f; int head_dim = 64;
    // optional split-K tail fix-up (MFMA kernel, 256-row tiles): workspace + an epoch unique since its header was zeroed
    void* fix_ws = nullptr; size_t fix_bytes = 0; unsigned fix_epoch = 0;
    // dtype == P2T_FP8: A and W are e4m3 bytes (row strides in bytes), one E8M0 scale byte per row of each
    const uint8_t* a_scale = nullptr; const uint8_t* w_scale = nullptr;
    const uint8_t* out_row_scale = nullptr;        // P2T_EPI_GELU_FP8: E8M0 byte of every output row
};
int gemm_nt(const GemmArgs& a, hipStream_t s);
// weight-streaming GEMM for M <= 64 rows (gemm_skinny.hip): bf16, epilogues STORE / STORE_F32 / RESID / SWIGLU, no bias;
// P2T_ERR_UNSUPPORTED for anything else
// epilogue arguments of launch_gemm_skinny_qkv_rope: rotation at position prompt_len[row / group] + step[0] and the cache append
struct SkinnyRope {
    const float* inv_freq = nullptr; const int32_t* prompt_len = nullptr; const int32_t* step = nullptr;
    int group = 1, nh = 0, nkv = 0, d = 0, G = 0; float q_scale = 1.f;
    void* q = nullptr;       // bf16 [M, nh, d]
    void* k = nullptr;       // bf16 [M, nkv, G, d]   (one layer of p2t_kv_cache.k_gen)
    void* vt = nullptr;      // bf16 [M, nkv, d, G]   (one layer of p2t_kv_cache.vt_gen)
};
int launch_gemm_skinny_qkv_rope(const void* x, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N, int64_t K, const SkinnyRope& ra,
                                hipStream_t s, int pre);
// pre: W is the pre-shuffled stream copy written by launch_preshuffle (ldw unused)
int launch_gemm_skinny(const void* x, int64_t lda, const void* W, int64_t ldw, void* out, int64_t ldc, int64_t M, int64_t N, int64_t K, int dtype,
                       int out_dtype, int epilogue, hipStream_t s, int pre = 0);
int launch_preshuffle(const void* W, int64_t ldw, int64_t N, int64_t K, void* out, hipStream_t s);
// e4m3 operands + one E8M0 scale byte per row (gemm_fp8 models): K % 128 == 0; ra: the QKV + rotation + cache-append form
int launch_gemm_skinny_fp8(const void* x, int64_t lda, const uint8_t* xs, const void* W, int64_t ldw, const uint8_t* ws, void* out, int64_t ldc, int64_t M,
                           int64_t N, int64_t K, int out_dtype, int epilogue, const SkinnyRope* ra, hipStream_t s, int pre);
int launch_preshuffle_fp8(const void* W, int64_t ldw, int64_t N, int64_t K, void* out, hipStream_t s);

int launch_attn_simple(const void* q, const void* k, const void* v, const uint8_t* key_mask, const int32_t* kv_info, void* out,
                       int64_t ld_out, int B, int T, int nh, int nkv, int d, int dp, float scale, int causal, int dtype,
                       float* lse, hipStream_t s);
int launch_attn_mfma(const void* q, const void* k, const void* v, const uint8_t* key_mask, const int32_t* kv_info, void* out,
                     int64_t ld_out, int B, int T, int nh, int nkv, int d, int dp, float scale, int causal, int log2_scores, float* lse, hipStream_t s);
// hand-placed form for head_dim padded to 64 with log2-scores q (attn_fwd64.hip; tools/gen_attn_fwd64.py writes its loop)
bool attn_fwd64_eligible(int64_t ld_out, int T, int nh, int nkv, int d, int dp, int log2_scores);
int launch_attn_fwd64(const void* q, const void* k, const void* v, const uint8_t* key_mask, const int32_t* kv_info, void* out, int64_t ld_out,
                      int B, int T, int nh, int nkv, int d, int causal, float* lse, hipStream_t s);
// log2_scores: q was stored pre-multiplied by scale * log2(e) (kLog2e below), so q k^T is already the base-2 exponent: `scale` is
// ignored and p = exp2(s - m).  The towers do this for bf16 models (one multiply + add less per score in the MFMA kernel).
int attention(const void* q, const void* k, const void* v, const uint8_t* key_mask, const int32_t* kv_info, void* out,
              int64_t ld_out, int B, int T, int nh, int nkv, int d, int dp, float scale, int causal, int dtype, int use_mfma,
              int log2_scores, hipStream_t s, float* lse = nullptr);
// lse (optional, f32 [B, nh, T]): natural-log sum-exp of the effective logits (scale * q.k, or ln 2 * q.k with log2_scores) of
// every query row, +inf for a row without a visible key -- what the attention backward (llama_train.hip) rebuilds P from.
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

// stage-2 training through the frozen decoder (llama_train.hip): the per-layer activations the backward reads again
struct LlamaTapeLayer {
    float* x_in; float* x_mid;          // residual stream before the layer / after its attention branch, f32 [M, H]
    void* q; void* k; void* v;          // rotated (and, for bf16 models, pre-scaled) heads, `dtype` [B, heads, T, dp]
    float* lse;                         // f32 [B, nh, T], see attention()
    void* ao;                           // attention output, `dtype` [M, QO]
    void* gu;                           // gate / up pre-activations, `dtype` [M, 2F] in the interleaved order of gu_w
};
struct LlamaTape {
    static constexpr int kMaxLayers = 128;
    int n_layers; bool overflow;
    float* x_last;                      // input of the final RMSNorm, f32 [M, H]
    LlamaTapeLayer layer[kMaxLayers];
};
size_t llama_tape_plan(const p2t_llama_config* c, int B, int T, void* base, size_t bytes, LlamaTape* tape);
int llama_forward_impl(const p2t_llama_config* c, const p2t_llama_weights* w, const int64_t* ids, const float* inputs_embeds, const int64_t* mask,
                       int B, int T, int k, float* out, void* workspace, size_t workspace_bytes, p2t_stream stream, const LlamaTape* tape,
                       const p2t_kv_cache* kv = nullptr);
// generation prefill (llama_decode.hip): layer l's rotated keys / values ([B, kv_heads, T, dp]) -> the prompt segment of the cache
int llama_kv_store(const p2t_llama_config* c, const p2t_kv_cache* kc, int layer, const void* k, const void* v, int B, int T, hipStream_t s);
int launch_swiglu_from_gu(const void* gu, int64_t ld_gu, void* act, int64_t ld_act, int64_t M, int64_t F, int dtype, hipStream_t s);
// dst[m, c] = (Tdst)src[m, c] for c < cols, 0 for cols <= c < ld_dst  (a GEMM operand with its K padding)
int launch_cast_rows(const void* src, int src_dtype, int64_t ld_src, void* dst, int dst_dtype, int64_t ld_dst, int64_t rows, int64_t cols, hipStream_t s);

// adapter tail helpers (adapter.hip)
int launch_adapter_dz2(const void* g2, const void* z2, const float* inv_norm, const float* dy, void* dz2, int64_t ld, int64_t M,
                       int D, int dtype, float drop_p, uint64_t drop_seed, hipStream_t s);

// bump allocator over a caller-provided workspace
struct Arena {
    char* base; size_t size; size_t off = 0; bool overflow = false;
    Arena(void* b, size_t n) : base((char*)b), size(n) {}
    void* take(size_t bytes) {
        const size_t a = (off + 255) & ~(size_t)255;
        if (a + bytes > size) { overflow = true; off = a + bytes; return base; }
        off = a + bytes;
        return base + a;
    }
};

static inline int head_dim_padded(int d) { return d <= 32 ? 32 : (d <= 64 ? 64 : 128); }

}  // namespace p2t
