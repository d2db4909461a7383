// Tower forwards: host-side launch sequences (no Python between kernels, no allocation, no sync;
// capturable in a hipGraph).  Residual stream is fp32; GEMM operands are `dtype` (bf16 or fp32).
//
//   p2t_esm2_forward         HF EsmModel.forward (transformers/models/esm/modeling_esm.py:685-755)
//                            as called by reference models/modeling_esm2llama_instruct.py:175-185
//   p2t_llama_hidden_forward HF LlamaModel.forward (transformers/models/llama/modeling_llama.py:367-417)
//                            as called by reference scripts/train_contrast.py:292-304
#include "common.h"
#include "kernels.h"

using namespace p2t;

namespace {

struct EsmBuffers {
    uint8_t* key_mask; int32_t* kv_info; float* emb_scale; float* inv_freq; float* cs;
    float* x; void* h; void* qkv; void* q; void* k; void* v; void* ao; void* ffn; void* fix;
    // gemm_fp8: e4m3 copies of the GEMM operands (row stride = K rounded up to 128 bytes) + one E8M0 scale per row
    uint8_t* hq; uint8_t* hs; uint8_t* aoq; uint8_t* aos; uint8_t* ffnq; uint8_t* ffns;
};

size_t esm_plan(const p2t_esm2_config* c, int B, int T, Arena* ar, EsmBuffers* b) {
    const size_t e = dtype_size(c->dtype);
    const int64_t M = (int64_t)B * T, H = c->hidden, F = c->ffn, Hp = round_up(H, 64), Fp = round_up(F, 64);
    const int d = c->head_dim, dp = head_dim_padded(d), nh = c->heads;
    Arena local(nullptr, ~(size_t)0 >> 1);
    Arena& a = ar ? *ar : local;
    EsmBuffers t;
    t.key_mask = (uint8_t*)a.take((size_t)M);
    t.kv_info = (int32_t*)a.take(sizeof(int32_t) * 2 * B);
    t.emb_scale = (float*)a.take(sizeof(float) * 2 * B);
    t.inv_freq = (float*)a.take(sizeof(float) * (d / 2 + 1));
    t.cs = (float*)a.take(sizeof(float) * (size_t)T * d);
    t.x = (float*)a.take(sizeof(float) * (size_t)M * H);
    t.h = a.take(e * (size_t)M * Hp);
    t.qkv = a.take(e * (size_t)M * 3 * H);
    t.q = a.take(e * (size_t)B * nh * T * dp);
    t.k = a.take(e * (size_t)B * nh * T * dp);
    t.v = a.take(e * (size_t)B * nh * T * dp);
    t.ao = a.take(e * (size_t)M * Hp);
    t.ffn = a.take(e * (size_t)M * Fp);
    t.fix = a.take(gemm_fix_workspace_bytes());
    t.hq = t.hs = t.aoq = t.aos = t.ffnq = t.ffns = nullptr;
    if (c->gemm_fp8) {
        const int64_t Hq = round_up(H, 128), Fq = round_up(F, 128);
        t.hq = (uint8_t*)a.take((size_t)M * Hq);   t.hs = (uint8_t*)a.take((size_t)M);
        t.aoq = (uint8_t*)a.take((size_t)M * Hq);  t.aos = (uint8_t*)a.take((size_t)M);
        t.ffnq = (uint8_t*)a.take((size_t)M * Fq); t.ffns = (uint8_t*)a.take((size_t)M);
    }
    if (b) *b = t;
    return a.off + 256;
}

struct LlamaBuffers {
    uint8_t* key_mask; int32_t* kv_info; float* inv_freq; float* cs;
    float* x; void* h; void* qkv; void* q; void* k; void* v; void* ao; void* act; void* fix;
    uint8_t* hq; uint8_t* hs; uint8_t* aoq; uint8_t* aos; uint8_t* actq; uint8_t* acts;      // gemm_fp8 (see EsmBuffers)
};

size_t llama_plan(const p2t_llama_config* c, int B, int T, Arena* ar, LlamaBuffers* b) {
    const size_t e = dtype_size(c->dtype);
    const int64_t M = (int64_t)B * T, H = c->hidden, F = c->ffn, Hp = round_up(H, 64), Fp = round_up(F, 64);
    const int d = c->head_dim, dp = head_dim_padded(d), nh = c->heads, nkv = c->kv_heads;
    const int64_t QO = round_up((int64_t)nh * d, 64);
    Arena local(nullptr, ~(size_t)0 >> 1);
    Arena& a = ar ? *ar : local;
    LlamaBuffers t;
    t.key_mask = (uint8_t*)a.take((size_t)M);
    t.kv_info = (int32_t*)a.take(sizeof(int32_t) * 2 * B);
    t.inv_freq = (float*)a.take(sizeof(float) * (d / 2 + 1));
    t.cs = (float*)a.take(sizeof(float) * (size_t)T * d);
    t.x = (float*)a.take(sizeof(float) * (size_t)M * H);
    t.h = a.take(e * (size_t)M * Hp);
    t.qkv = a.take(e * (size_t)M * (nh + 2 * nkv) * d);
    t.q = a.take(e * (size_t)B * nh * T * dp);
    t.k = a.take(e * (size_t)B * nkv * T * dp);
    t.v = a.take(e * (size_t)B * nkv * T * dp);
    t.ao = a.take(e * (size_t)M * QO);
    t.act = a.take(e * (size_t)M * Fp);
    t.fix = a.take(gemm_fix_workspace_bytes());
    t.hq = t.hs = t.aoq = t.aos = t.actq = t.acts = nullptr;
    if (c->gemm_fp8) {
        const int64_t Hq = round_up(H, 128), Fq = round_up(F, 128), QOq = round_up((int64_t)nh * d, 128);
        t.hq = (uint8_t*)a.take((size_t)M * Hq);    t.hs = (uint8_t*)a.take((size_t)M);
        t.aoq = (uint8_t*)a.take((size_t)M * QOq);  t.aos = (uint8_t*)a.take((size_t)M);
        t.actq = (uint8_t*)a.take((size_t)M * Fq);  t.acts = (uint8_t*)a.take((size_t)M);
    }
    if (b) *b = t;
    return a.off + 256;
}

}  // namespace

extern "C" size_t p2t_esm2_workspace_bytes(const p2t_esm2_config* cfg, int B, int T) {
    if (!cfg || B <= 0 || T <= 0) return 0;
    return esm_plan(cfg, B, T, nullptr, nullptr);
}

extern "C" int p2t_esm2_forward(const p2t_esm2_config* c, const p2t_esm2_weights* w, const int64_t* ids, const int64_t* mask,
                                int B, int T, void* out, int64_t ld_out, void* workspace, size_t workspace_bytes,
                                p2t_stream stream) {
    P2T_REQUIRE(c && w && ids && mask && out && workspace && B > 0 && T > 0, "p2t_esm2_forward: null/empty argument");
    P2T_REQUIRE(c->hidden == c->heads * c->head_dim && c->head_dim % 4 == 0 && c->head_dim <= 128 && c->hidden % 16 == 0 && c->ffn % 16 == 0,
                "p2t_esm2_forward: unsupported shape hidden=%d heads=%d head_dim=%d ffn=%d", c->hidden, c->heads, c->head_dim, c->ffn);
    P2T_REQUIRE(ld_out >= c->hidden && ld_out % 4 == 0, "p2t_esm2_forward: ld_out");
    P2T_REQUIRE(!c->gemm_fp8 || c->dtype == P2T_BF16, "p2t_esm2_forward: gemm_fp8 needs bf16 activations (dtype = P2T_BF16)");
    P2T_REQUIRE(w->layers && w->word_emb && w->final_ln_w && w->final_ln_b, "p2t_esm2_forward: missing weights");
    P2T_REQUIRE(workspace_bytes >= p2t_esm2_workspace_bytes(c, B, T), "p2t_esm2_forward: workspace too small (%zu < %zu)",
                workspace_bytes, p2t_esm2_workspace_bytes(c, B, T));
    hipStream_t s = (hipStream_t)stream;
    Arena ar(workspace, workspace_bytes);
    EsmBuffers b;
    esm_plan(c, B, T, &ar, &b);
    P2T_REQUIRE(!ar.overflow, "p2t_esm2_forward: workspace overflow");
    const int dt = c->dtype;
    const int64_t M = (int64_t)B * T, H = c->hidden, F = c->ffn, Hp = round_up(H, 64), Fp = round_up(F, 64);
    const int d = c->head_dim, dp = head_dim_padded(d), nh = c->heads;

    P2T_TRY(launch_mask_prepare(ids, mask, B, T, c->mask_id, c->token_dropout, b.key_mask, b.kv_info, b.emb_scale, s));
    P2T_TRY(launch_esm_embed(ids, mask, w->word_emb, dt, b.emb_scale, T, (int)H, c->vocab, c->mask_id, c->token_dropout, b.x, M, s));
    if (c->emb_layer_norm_before) {
        // LayerNorm of the embeddings, then re-apply the attention mask (modeling_esm.py:264-268)
        set_error("p2t_esm2_forward: emb_layer_norm_before=True is not an ESM2 configuration");
        return P2T_ERR_UNSUPPORTED;
    }
    const float* inv_freq = w->inv_freq;
    if (!inv_freq) {
        P2T_TRY(launch_inv_freq(b.inv_freq, d / 2, c->rope_theta, 0, 1.f, 1.f, 1.f, 1.f, s));
        inv_freq = b.inv_freq;
    }
    P2T_TRY(launch_rope_table(inv_freq, T, d / 2, b.cs, s));
    // ESM scales q before rotary; SDPA scale is 1.0.  bf16 models: log2(e) rides along, so the attention kernel's exponent is
    // q k^T itself (kernels.h attention(): log2_scores) -- q is rounded to bf16 once either way
    const int l2s = dt == P2T_BF16;
    const float q_scale = (l2s ? kLog2e : 1.0f) / sqrtf((float)d);
    P2T_CHECK_HIP(hipMemsetAsync(b.fix, 0, gemm_fix_header_bytes(), s));      // split-K tail flags; epochs below are unique
    unsigned epoch = 0;
    auto with_fix = [&](GemmArgs& g) { g.fix_ws = b.fix; g.fix_bytes = gemm_fix_workspace_bytes(); g.fix_epoch = ++epoch; };
    const int64_t Hq = round_up(H, 128), Fq = round_up(F, 128);
    for (int l = 0; c->gemm_fp8 && l < c->n_layers; ++l) {
        // fp8 GEMMs (BASELINE.json configs[4]): the same layer, operands e4m3 with a power-of-two scale per row.  The
        // LayerNorms write their output directly in that format; the attention output and the GELU output are produced
        // in bf16 (attention reads bf16 q/k/v; GELU rows span many tiles) and quantised by one streaming pass each.
        const p2t_esm2_layer& L = w->layers[l];
        P2T_REQUIRE(L.qkv_ws && L.o_ws && L.fc1_ws && L.fc2_ws, "p2t_esm2_forward: gemm_fp8 needs the row scales of layer %d", l);
        auto fp8 = [&](const void* A, int64_t lda, const uint8_t* as, const void* W, const uint8_t* ws, const float* bias, void* out, int64_t ldc,
                       int64_t N, int64_t K, int out_dtype, int epi) {
            GemmArgs g{A, lda, W, K, bias, out, ldc, nullptr, M, N, K, P2T_FP8, out_dtype, epi, 0, 1, -1, 0.f, 0, 0};
            g.a_scale = as; g.w_scale = ws;
            return g;
        };
        P2T_TRY(launch_layernorm_fp8(b.x, H, L.ln1_w, L.ln1_b, c->layer_norm_eps, b.hq, Hq, b.hs, M, H, 0.f, 0.f, nullptr, s));
        if (d == 64) {
            GemmArgs g1 = fp8(b.hq, Hq, b.hs, L.qkv_w, L.qkv_ws, L.qkv_b, nullptr, 0, 3 * H, Hq, dt, P2T_EPI_QKV_ROPE);
            g1.cs = b.cs; g1.q = b.q; g1.k = b.k; g1.v = b.v; g1.seq = T; g1.nh = nh; g1.nkv = nh; g1.q_scale = q_scale;
            P2T_TRY(gemm_nt(g1, s));
        } else {
            GemmArgs g1 = fp8(b.hq, Hq, b.hs, L.qkv_w, L.qkv_ws, L.qkv_b, b.qkv, 3 * H, 3 * H, Hq, dt, P2T_EPI_STORE);
            g1.n_zero = (int)(3 * H);
            P2T_TRY(gemm_nt(g1, s));
            P2T_TRY(launch_qkv_post(b.qkv, 3 * H, b.cs, b.q, b.k, b.v, B, T, nh, nh, d, dp, q_scale, dt, s));
        }
        P2T_TRY(attention(b.q, b.k, b.v, b.key_mask, b.kv_info, b.ao, Hp, B, T, nh, nh, d, dp, 1.0f, 0, dt, -1, l2s, s));
        P2T_TRY(launch_quant_rows(b.ao, dt, Hp, M, H, b.aoq, Hq, b.aos, s));
        GemmArgs g2 = fp8(b.aoq, Hq, b.aos, L.o_w, L.o_ws, L.o_b, b.x, H, H, Hq, P2T_F32, P2T_EPI_RESID);
        P2T_TRY(gemm_nt(g2, s));
        // FFN-up writes its GELU output straight as e4m3 when the layer carries the weight-norm bound: the per-token scale is
        // then known before the GEMM runs (||LN(x)||_2 * max_n ||W_n||_2 + max |bias| bounds every element of the row),
        // derived by the LayerNorm kernel that already holds the row; otherwise: bf16 output + one quantise pass
        const bool fused_q = L.fc1_wnorm_bound > 0.f;
        P2T_TRY(launch_layernorm_fp8(b.x, H, L.ln2_w, L.ln2_b, c->layer_norm_eps, b.hq, Hq, b.hs, M, H, L.fc1_wnorm_bound, L.fc1_babs_bound,
                                     fused_q ? b.ffns : nullptr, s));
        if (fused_q) {
            GemmArgs g3 = fp8(b.hq, Hq, b.hs, L.fc1_w, L.fc1_ws, L.fc1_b, b.ffnq, Fq, F, Hq, dt, P2T_EPI_GELU_FP8);
            g3.out_row_scale = b.ffns;
            P2T_TRY(gemm_nt(g3, s));
        } else {
            GemmArgs g3 = fp8(b.hq, Hq, b.hs, L.fc1_w, L.fc1_ws, L.fc1_b, b.ffn, Fp, F, Hq, dt, P2T_EPI_GELU);
            P2T_TRY(gemm_nt(g3, s));
            P2T_TRY(launch_quant_rows(b.ffn, dt, Fp, M, F, b.ffnq, Fq, b.ffns, s));
        }
        GemmArgs g4 = fp8(b.ffnq, Fq, b.ffns, L.fc2_w, L.fc2_ws, L.fc2_b, b.x, H, H, Fq, P2T_F32, P2T_EPI_RESID);
        P2T_TRY(gemm_nt(g4, s));
    }
    for (int l = 0; !c->gemm_fp8 && l < c->n_layers; ++l) {
        const p2t_esm2_layer& L = w->layers[l];
        P2T_TRY(launch_layernorm(b.x, H, L.ln1_w, L.ln1_b, c->layer_norm_eps, b.h, Hp, M, H, dt, s));
        if (d == 64) {
            // QKV projection with bias + q-scale + rotary + head split fused into the GEMM epilogue
            GemmArgs g1{b.h, Hp, L.qkv_w, Hp, L.qkv_b, nullptr, 0, nullptr, M, 3 * H, Hp, dt, dt, P2T_EPI_QKV_ROPE, 0, -1, -1, 0.f, 0, 0};
            g1.cs = b.cs; g1.q = b.q; g1.k = b.k; g1.v = b.v; g1.seq = T; g1.nh = nh; g1.nkv = nh; g1.q_scale = q_scale;
            with_fix(g1);
            P2T_TRY(gemm_nt(g1, s));
        } else {
            GemmArgs g1{b.h, Hp, L.qkv_w, Hp, L.qkv_b, b.qkv, 3 * H, nullptr, M, 3 * H, Hp, dt, dt, P2T_EPI_STORE, 0, -1, (int)(3 * H), 0.f, 0, 0};
            P2T_TRY(gemm_nt(g1, s));
            P2T_TRY(launch_qkv_post(b.qkv, 3 * H, b.cs, b.q, b.k, b.v, B, T, nh, nh, d, dp, q_scale, dt, s));
        }
        P2T_TRY(attention(b.q, b.k, b.v, b.key_mask, b.kv_info, b.ao, Hp, B, T, nh, nh, d, dp, 1.0f, 0, dt, -1, l2s, s));
        GemmArgs g2{b.ao, Hp, L.o_w, Hp, L.o_b, b.x, H, nullptr, M, H, Hp, dt, P2T_F32, P2T_EPI_RESID, 0, -1, -1, 0.f, 0, 0};
        with_fix(g2);
        P2T_TRY(gemm_nt(g2, s));
        P2T_TRY(launch_layernorm(b.x, H, L.ln2_w, L.ln2_b, c->layer_norm_eps, b.h, Hp, M, H, dt, s));
        GemmArgs g3{b.h, Hp, L.fc1_w, Hp, L.fc1_b, b.ffn, Fp, nullptr, M, F, Hp, dt, dt, P2T_EPI_GELU, 0, -1, -1, 0.f, 0, 0};
        with_fix(g3);
        P2T_TRY(gemm_nt(g3, s));
        GemmArgs g4{b.ffn, Fp, L.fc2_w, Fp, L.fc2_b, b.x, H, nullptr, M, H, Fp, dt, P2T_F32, P2T_EPI_RESID, 0, -1, -1, 0.f, 0, 0};
        with_fix(g4);
        P2T_TRY(gemm_nt(g4, s));
    }
    return launch_layernorm(b.x, H, w->final_ln_w, w->final_ln_b, c->layer_norm_eps, out, ld_out, M, H, dt, s);
}

extern "C" size_t p2t_llama_workspace_bytes(const p2t_llama_config* cfg, int B, int T) {
    if (!cfg || B <= 0 || T <= 0) return 0;
    return llama_plan(cfg, B, T, nullptr, nullptr);
}

// ids != nullptr: token embedding lookup; else inputs_embeds (f32 [B*T, hidden]) is the layer-0 input (SFT path:
// placeholder positions already replaced by adapter outputs, reference models/modeling_esm2llama_instruct.py:195-215)
// tape != nullptr (p2t_llama_train_forward, llama_train.hip): every layer keeps what the backward reads again -- its input and
// mid-layer residual streams, the rotated heads, the attention output and log-sum-exps, the gate / up pre-activations (the
// gate/up GEMM then stores them plainly and SwiGLU runs as its own pass).
int p2t::llama_forward_impl(const p2t_llama_config* c, const p2t_llama_weights* w, const int64_t* ids, const float* inputs_embeds,
                            const int64_t* mask, int B, int T, int k, float* out, void* workspace,
                            size_t workspace_bytes, p2t_stream stream, const LlamaTape* tape, const p2t_kv_cache* kv) {
    P2T_REQUIRE(c && w && (ids || inputs_embeds) && mask && out && workspace && B > 0 && T > 0, "p2t_llama_hidden_forward: null/empty argument");
    P2T_REQUIRE(k >= 0 && k <= c->n_layers, "p2t_llama_hidden_forward: hidden_states[%d] out of range for %d layers", k, c->n_layers);
    P2T_REQUIRE(c->heads % c->kv_heads == 0 && c->head_dim % 4 == 0 && c->head_dim <= 128 && c->hidden % 16 == 0 && c->ffn % 32 == 0 &&
                    ((int64_t)c->heads * c->head_dim) % 16 == 0,
                "p2t_llama_hidden_forward: unsupported shape");
    P2T_REQUIRE((w->embed || !ids) && (k == 0 || w->layers) && (k < c->n_layers || w->final_norm_w), "p2t_llama_hidden_forward: missing weights");
    P2T_REQUIRE(workspace_bytes >= p2t_llama_workspace_bytes(c, B, T), "p2t_llama_hidden_forward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Arena ar(workspace, workspace_bytes);
    LlamaBuffers b;
    llama_plan(c, B, T, &ar, &b);
    P2T_REQUIRE(!ar.overflow, "p2t_llama_hidden_forward: workspace overflow");
    const int dt = c->dtype;
    const int64_t M = (int64_t)B * T, H = c->hidden, F = c->ffn, Hp = round_up(H, 64), Fp = round_up(F, 64);
    const int d = c->head_dim, dp = head_dim_padded(d), nh = c->heads, nkv = c->kv_heads;
    const int64_t NQKV = (int64_t)(nh + 2 * nkv) * d, QO = round_up((int64_t)nh * d, 64);

    P2T_TRY(launch_mask_prepare(nullptr, mask, B, T, -1, 0, b.key_mask, b.kv_info, nullptr, s));
    if (ids) {
        P2T_TRY(launch_llama_embed(ids, w->embed, dt, (int)H, c->vocab, b.x, M, s));
    } else {
        P2T_CHECK_HIP(hipMemcpyAsync(b.x, inputs_embeds, sizeof(float) * (size_t)M * H, hipMemcpyDeviceToDevice, s));
    }
    const float* inv_freq = w->inv_freq;
    if (!inv_freq) {
        P2T_TRY(launch_inv_freq(b.inv_freq, d / 2, c->rope_theta, c->rope_llama3, c->rope_factor, c->rope_low_freq_factor,
                                c->rope_high_freq_factor, (float)c->rope_original_max_pos, s));
        inv_freq = b.inv_freq;
    }
    P2T_TRY(launch_rope_table(inv_freq, T, d / 2, b.cs, s));
    const float scale = 1.0f / sqrtf((float)d);
    // bf16 models: the softmax scale and log2(e) are folded into q where it is written (rotation is linear), and the attention
    // kernel's exponent is q k^T itself (kernels.h attention(): log2_scores)
    const int l2s = dt == P2T_BF16;
    const float q_fold = l2s ? scale * kLog2e : 1.0f;
    P2T_CHECK_HIP(hipMemsetAsync(b.fix, 0, gemm_fix_header_bytes(), s));      // split-K flags; epochs below are unique
    unsigned epoch = 0;
    auto with_fix = [&](GemmArgs& g) { g.fix_ws = b.fix; g.fix_bytes = gemm_fix_workspace_bytes(); g.fix_epoch = ++epoch; };
    P2T_REQUIRE(!c->gemm_fp8 || dt == P2T_BF16, "p2t_llama_hidden_forward: gemm_fp8 needs bf16 activations (dtype = P2T_BF16)");
    const int64_t Hq = round_up(H, 128), Fq = round_up(F, 128), QOq = round_up((int64_t)nh * d, 128);
    for (int l = 0; c->gemm_fp8 && l < k; ++l) {
        const p2t_llama_layer& L = w->layers[l];
        P2T_REQUIRE(L.qkv_ws && L.o_ws && L.gu_ws && L.down_ws, "p2t_llama_hidden_forward: gemm_fp8 needs the row scales of layer %d", l);
        auto fp8 = [&](const void* A, int64_t lda, const uint8_t* as, const void* W, const uint8_t* ws, void* out, int64_t ldc, int64_t N,
                       int64_t K, int out_dtype, int epi) {
            GemmArgs g{A, lda, W, K, nullptr, out, ldc, nullptr, M, N, K, P2T_FP8, out_dtype, epi, 0, 1, -1, 0.f, 0, 0};
            g.a_scale = as; g.w_scale = ws;
            return g;
        };
        P2T_REQUIRE(!L.q_norm_w == !L.k_norm_w, "p2t_llama_hidden_forward: q_norm_w and k_norm_w go together (layer %d)", l);
        P2T_TRY(launch_rmsnorm_fp8(b.x, H, L.ln1_w, c->rms_norm_eps, b.hq, Hq, b.hs, M, H, s));
        if (L.q_norm_w) {          // Qwen3: projection -> per-head RMSNorm -> rotation (not fusable: the norm spans the head)
            GemmArgs g1 = fp8(b.hq, Hq, b.hs, L.qkv_w, L.qkv_ws, b.qkv, NQKV, NQKV, Hq, dt, P2T_EPI_STORE);
            g1.n_zero = (int)NQKV;
            P2T_TRY(gemm_nt(g1, s));
            P2T_TRY(launch_qk_norm_rope(b.qkv, NQKV, b.cs, L.q_norm_w, L.k_norm_w, c->rms_norm_eps, b.q, b.k, b.v, B, T, nh, nkv, d, dp, q_fold, dt, s));
        } else if (d == 64 || d == 128) {
            GemmArgs g1 = fp8(b.hq, Hq, b.hs, L.qkv_w, L.qkv_ws, nullptr, 0, NQKV, Hq, dt, P2T_EPI_QKV_ROPE);
            g1.cs = b.cs; g1.q = b.q; g1.k = b.k; g1.v = b.v; g1.seq = T; g1.nh = nh; g1.nkv = nkv; g1.q_scale = q_fold; g1.head_dim = d;
            P2T_TRY(gemm_nt(g1, s));
        } else {
            GemmArgs g1 = fp8(b.hq, Hq, b.hs, L.qkv_w, L.qkv_ws, b.qkv, NQKV, NQKV, Hq, dt, P2T_EPI_STORE);
            g1.n_zero = (int)NQKV;
            P2T_TRY(gemm_nt(g1, s));
            P2T_TRY(launch_qkv_post(b.qkv, NQKV, b.cs, b.q, b.k, b.v, B, T, nh, nkv, d, dp, q_fold, dt, s));
        }
        P2T_TRY(attention(b.q, b.k, b.v, b.key_mask, b.kv_info, b.ao, QO, B, T, nh, nkv, d, dp, scale, 1, dt, -1, l2s, s));
        if (kv) P2T_TRY(llama_kv_store(c, kv, l, b.k, b.v, B, T, s));
        P2T_TRY(launch_quant_rows(b.ao, dt, QO, M, (int64_t)nh * d, b.aoq, QOq, b.aos, s));
        GemmArgs g2 = fp8(b.aoq, QOq, b.aos, L.o_w, L.o_ws, b.x, H, H, QOq, P2T_F32, P2T_EPI_RESID);
        P2T_TRY(gemm_nt(g2, s));
        P2T_TRY(launch_rmsnorm_fp8(b.x, H, L.ln2_w, c->rms_norm_eps, b.hq, Hq, b.hs, M, H, s));
        GemmArgs g3 = fp8(b.hq, Hq, b.hs, L.gu_w, L.gu_ws, b.act, Fp, 2 * F, Hq, dt, P2T_EPI_SWIGLU);
        P2T_TRY(gemm_nt(g3, s));
        P2T_TRY(launch_quant_rows(b.act, dt, Fp, M, F, b.actq, Fq, b.acts, s));
        GemmArgs g4 = fp8(b.actq, Fq, b.acts, L.down_w, L.down_ws, b.x, H, H, Fq, P2T_F32, P2T_EPI_RESID);
        P2T_TRY(gemm_nt(g4, s));
    }
    for (int l = 0; !c->gemm_fp8 && l < k; ++l) {
        const p2t_llama_layer& L = w->layers[l];
        P2T_REQUIRE(!L.q_norm_w == !L.k_norm_w, "p2t_llama_hidden_forward: q_norm_w and k_norm_w go together (layer %d)", l);
        float* lse = nullptr;
        if (tape) {                       // this layer's heads / attention output live on the tape instead of the shared workspace
            const LlamaTapeLayer& S = tape->layer[l];
            b.q = S.q; b.k = S.k; b.v = S.v; b.ao = S.ao;
            lse = S.lse;
            P2T_CHECK_HIP(hipMemcpyAsync(S.x_in, b.x, sizeof(float) * (size_t)M * H, hipMemcpyDeviceToDevice, s));
        }
        P2T_TRY(launch_rmsnorm(b.x, H, L.ln1_w, c->rms_norm_eps, b.h, Hp, M, H, dt, s));
        if (L.q_norm_w) {          // Qwen3: projection -> per-head RMSNorm -> rotation (not fusable: the norm spans the head)
            GemmArgs g1{b.h, Hp, L.qkv_w, Hp, nullptr, b.qkv, NQKV, nullptr, M, NQKV, Hp, dt, dt, P2T_EPI_STORE, 0, -1, (int)NQKV, 0.f, 0, 0};
            P2T_TRY(gemm_nt(g1, s));
            P2T_TRY(launch_qk_norm_rope(b.qkv, NQKV, b.cs, L.q_norm_w, L.k_norm_w, c->rms_norm_eps, b.q, b.k, b.v, B, T, nh, nkv, d, dp, q_fold, dt, s));
        } else if (d == 64 || d == 128) {
            // bias-free QKV projection + rotary + head split in the GEMM epilogue (d = 128: rows packed per head, see the header)
            GemmArgs g1{b.h, Hp, L.qkv_w, Hp, nullptr, nullptr, 0, nullptr, M, NQKV, Hp, dt, dt, P2T_EPI_QKV_ROPE, 0, -1, -1, 0.f, 0, 0};
            g1.cs = b.cs; g1.q = b.q; g1.k = b.k; g1.v = b.v; g1.seq = T; g1.nh = nh; g1.nkv = nkv; g1.q_scale = q_fold; g1.head_dim = d;
            P2T_TRY(gemm_nt(g1, s));
        } else {
            GemmArgs g1{b.h, Hp, L.qkv_w, Hp, nullptr, b.qkv, NQKV, nullptr, M, NQKV, Hp, dt, dt, P2T_EPI_STORE, 0, -1, (int)NQKV, 0.f, 0, 0};
            P2T_TRY(gemm_nt(g1, s));
            P2T_TRY(launch_qkv_post(b.qkv, NQKV, b.cs, b.q, b.k, b.v, B, T, nh, nkv, d, dp, q_fold, dt, s));
        }
        P2T_TRY(attention(b.q, b.k, b.v, b.key_mask, b.kv_info, b.ao, QO, B, T, nh, nkv, d, dp, scale, 1, dt, -1, l2s, s, lse));
        if (kv) P2T_TRY(llama_kv_store(c, kv, l, b.k, b.v, B, T, s));      // generation prefill: this layer's keys / values -> prompt segment
        GemmArgs g2{b.ao, QO, L.o_w, QO, nullptr, b.x, H, nullptr, M, H, QO, dt, P2T_F32, P2T_EPI_RESID, 0, -1, -1, 0.f, 0, 0};
        P2T_TRY(gemm_nt(g2, s));
        if (tape) P2T_CHECK_HIP(hipMemcpyAsync(tape->layer[l].x_mid, b.x, sizeof(float) * (size_t)M * H, hipMemcpyDeviceToDevice, s));
        P2T_TRY(launch_rmsnorm(b.x, H, L.ln2_w, c->rms_norm_eps, b.h, Hp, M, H, dt, s));
        if (tape) {
            GemmArgs g3{b.h, Hp, L.gu_w, Hp, nullptr, tape->layer[l].gu, 2 * F, nullptr, M, 2 * F, Hp, dt, dt, P2T_EPI_STORE, 0, -1, (int)(2 * F), 0.f, 0, 0};
            P2T_TRY(gemm_nt(g3, s));
            P2T_TRY(launch_swiglu_from_gu(tape->layer[l].gu, 2 * F, b.act, Fp, M, F, dt, s));
        } else {
            GemmArgs g3{b.h, Hp, L.gu_w, Hp, nullptr, b.act, Fp, nullptr, M, 2 * F, Hp, dt, dt, P2T_EPI_SWIGLU, 0, -1, -1, 0.f, 0, 0};
            P2T_TRY(gemm_nt(g3, s));
        }
        GemmArgs g4{b.act, Fp, L.down_w, Fp, nullptr, b.x, H, nullptr, M, H, Fp, dt, P2T_F32, P2T_EPI_RESID, 0, -1, -1, 0.f, 0, 0};
        with_fix(g4);
        P2T_TRY(gemm_nt(g4, s));
    }
    if (tape) P2T_CHECK_HIP(hipMemcpyAsync(tape->x_last, b.x, sizeof(float) * (size_t)M * H, hipMemcpyDeviceToDevice, s));
    if (k == c->n_layers) return launch_rmsnorm(b.x, H, w->final_norm_w, c->rms_norm_eps, out, H, M, H, P2T_F32, s);
    P2T_CHECK_HIP(hipMemcpyAsync(out, b.x, sizeof(float) * (size_t)M * H, hipMemcpyDeviceToDevice, s));
    return P2T_OK;
}

extern "C" int p2t_llama_hidden_forward(const p2t_llama_config* c, const p2t_llama_weights* w, const int64_t* ids,
                                        const int64_t* mask, int B, int T, int k, float* out, void* workspace,
                                        size_t workspace_bytes, p2t_stream stream) {
    P2T_REQUIRE(ids, "p2t_llama_hidden_forward: null ids");
    return llama_forward_impl(c, w, ids, nullptr, mask, B, T, k, out, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int p2t_llama_hidden_forward_embeds(const p2t_llama_config* c, const p2t_llama_weights* w, const float* inputs_embeds,
                                               const int64_t* mask, int B, int T, int k, float* out, void* workspace,
                                               size_t workspace_bytes, p2t_stream stream) {
    P2T_REQUIRE(inputs_embeds, "p2t_llama_hidden_forward_embeds: null inputs_embeds");
    return llama_forward_impl(c, w, nullptr, inputs_embeds, mask, B, T, k, out, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int p2t_llama_embed_tokens(const p2t_llama_config* c, const p2t_llama_weights* w, const int64_t* ids, int64_t n_tokens,
                                      float* out, p2t_stream stream) {
    P2T_REQUIRE(c && w && w->embed && ids && out && n_tokens > 0, "p2t_llama_embed_tokens: null/empty argument");
    return launch_llama_embed(ids, w->embed, c->dtype, c->hidden, c->vocab, out, n_tokens, (hipStream_t)stream);
}
