/* libp2t_hip -- C ABI of the MI355X-native contrastive-alignment hot path of Prot2Text-V2.
 *
 * The reference (RockingMat/Prot2Text-V2-esm3) is pure Python: it has no FFI, the hot path is
 * reached through nn.Module calls.  This header is the drop-in boundary a maintainer binds with
 * ctypes (INTEGRATION.md shows the stub); every entry point names the reference call it replaces:
 *
 *   p2t_esm2_forward          Esm2LlamaInstructForCausalLM.forward(return_encoder_outputs=True)
 *                             models/modeling_esm2llama_instruct.py:175-189 -> HF EsmModel.forward
 *   p2t_adapter_forward       ModalityAdapter.forward, models/modeling_esm2llama_instruct.py:60-68
 *   p2t_adapter_backward      autograd of the same (loss.backward(), scripts/train_contrast.py:448)
 *   p2t_llama_hidden_forward  llm_decoder.model(..., output_hidden_states=True).hidden_states[k]
 *                             scripts/train_contrast.py:292-304 -> HF LlamaModel.forward
 *   p2t_readout / _backward   readout_embeddings, scripts/train_contrast.py:198-248
 *   p2t_l2norm_rows           torch.nn.functional.normalize(p=2, dim=-1), train_contrast.py:354,365
 *   p2t_infonce_forward/_backward  SegmentedBatchInfoNCELoss / BatchInfoNCELoss, train_contrast.py:72-114
 *   p2t_clip_adamw_step       clip_grad_norm_ + AdamW.step, train_contrast.py:453-465,621-626
 *
 * Conventions: plain pointers and sizes, no framework types.  All pointers are DEVICE pointers
 * unless named host_*.  Every call only ENQUEUES work on `stream` (a hipStream_t passed as void*),
 * never allocates, never synchronises; buffers are owned by the caller (PyTorch's allocator in the
 * Python host).  Return value: P2T_OK or a negative code, text via p2t_last_error().
 * The library is re-entrant per stream.  Global state: the thread-local error string; the measurement hooks
 * (p2t_prof_*, mutex-guarded, off by default); the GEMM launch-form override (p2t_set_gemm_policy, one atomic word,
 * default 0); one sticky fault word per GPU in device memory (p2t_fault_status).  No environment variable is read.
 */
#ifndef P2T_HIP_H
#define P2T_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2T_VERSION 103

enum { P2T_OK = 0, P2T_ERR_ARG = -1, P2T_ERR_HIP = -2, P2T_ERR_UNSUPPORTED = -3 };
enum { P2T_F32 = 0, P2T_BF16 = 1,                         /* storage dtype of weights / activations */
       P2T_FP8 = 2 };                                      /* GEMM operands only: OCP e4m3fn bytes + one E8M0 scale byte per row */
enum { P2T_READOUT_LAST = 0, P2T_READOUT_MEAN = 1, P2T_READOUT_STD = 2, P2T_READOUT_MIX = 3 };
enum {                                                     /* GEMM epilogues (p2t_gemm_nt) */
    P2T_EPI_STORE = 0,      /* C = acc + bias                                   */
    P2T_EPI_GELU = 1,       /* C = gelu_erf(acc + bias); optional Z = acc + bias */
    P2T_EPI_RESID = 2,      /* R(f32, in place) += acc + bias                   */
    P2T_EPI_SWIGLU = 3,     /* C[m, f] = silu(gate) * up, gate/up interleaved by 32 rows of W */
    P2T_EPI_STORE_F32 = 4,  /* C(f32) = acc (+ C if accumulate)                  */
    P2T_EPI_GELU_BWD = 5,   /* C = acc * gelu_erf'(Z), Z read from z (adapter backward) */
    P2T_EPI_QKV_ROPE = 6,   /* internal to the towers: bias + q-scale + rotary + head split, head_dim 64 */
    P2T_EPI_GELU_FP8 = 7    /* p2t_gemm_nt_fp8 only: C(e4m3 bytes) = gelu_erf(acc + bias) * 2^-(row_scale[m] - 127) */
};

typedef void* p2t_stream;

int p2t_version(void);
const char* p2t_last_error(void);
/* sizeof() of the ABI structs, for bindings to self-check: 0 esm2_config, 1 esm2_layer, 2 esm2_weights,
 * 3 llama_config, 4 llama_layer, 5 llama_weights, 6 adapter_config, 7 adapter_weights, 8 adapter_saved, 9 llama_layer_t. */
size_t p2t_struct_size(int which);

/* ---------------------------------------------------------------- measurement hooks (bench.py) */
/* When enabled, every bf16 MFMA GEMM (class 0), MFMA attention (class 1) and fp8 MFMA GEMM (class 2; only reported
 * when n_classes >= 3) launch is bracketed by HIP events on its
 * own launch stream.  p2t_prof_collect synchronises those events (the only host sync in the library) and returns,
 * per class, the summed kernel time in ms, the launch count and the algorithmic FLOPs (GEMM: 2 M N K; attention:
 * 4 B nh T^2 d, halved when causal); it then resets the record list. */
int p2t_prof_enable(int on);
int p2t_prof_collect(double* ms, int64_t* launches, double* flops, int n_classes);
/* Launch-form override of the MFMA GEMMs (process-wide, one atomic word): 0 = the measured default policy; 9 = the same
 * policy without the four-wave kernels (bf16 and fp8) -- the eight-wave forms the bit-identity tests compare against
 * (tests/test_gpu_fullsize.py).  Results are identical up to the fp32 summation order.  Any other value is refused by
 * the product library: the launch-form zoo of the development rounds (forced tile heights, forced split-K, the other
 * generated instruction order ...) is compiled into the LAB build only (csrc/Makefile `make lab`, -DP2T_LAB,
 * tools/lab/gemm_forms_lab.h -> tools/build/libp2t_lab.so), which tools/ and tests/test_gpu_lab_forms.py load
 * through P2T_HIP_LIB. */
int p2t_set_gemm_policy(int policy);
int p2t_is_lab_build(void);     /* 1: lab build, 0: product library */

/* ---------------------------------------------------------------- runtime guards of the epoch loop */
/* The reference's train_epoch / eval_epoch keep `ddp_loss = [sum of batch losses, batches seen]` and `ddp_gradnorm =
 * [sum of gradient norms, optimizer steps]` (scripts/train_contrast.py:410-413,443-444,461-462,493-512) and look at
 * every batch loss on the host (`loss.item()`, the "impossible batch loss" print of :431-434, the epoch NaN abort of
 * :476-480).  p2t_epoch_accumulate keeps the same four sums on the device -- sums f32[4] = {loss sum, batches,
 * gradient-norm sum, optimizer steps}; grad_norm may be NULL when no optimizer step ran after this batch -- and the
 * guards as flags i32[4] = {number of impossible losses so far (NaN, inf or <= 0: the condition of :433), batch index
 * of the first one (caller initialises to -1), OR of the GPU's sticky fault word, bit pattern of the first impossible
 * loss}: one launch per batch, no host synchronisation; the host reads `flags` every N batches and at the epoch end.
 *
 * Fault word: one word per GPU, bit 0 = a split-K consumer of an MFMA GEMM gave up waiting for its producer (bounded
 * spin, never expected: csrc/gemm_mfma.hip TILE_CONSUME, csrc/gemm_w4.hip).  While it is set every such consumer
 * writes NaN tiles, so the loss of the step is NaN as well.  Sticky: only p2t_fault_status(clear = 1) resets it.
 * p2t_fault_status copies it to the host (SYNCHRONOUS hipMemcpy: for epoch boundaries and tests);
 * p2t_fault_inject sets it from a stream (tests of the guard). */
int p2t_epoch_accumulate(const float* loss, const float* grad_norm, int batch_idx, float* sums, int32_t* flags,
                         p2t_stream stream);
int p2t_fault_status(unsigned* host_out, int clear);
int p2t_fault_inject(unsigned value, p2t_stream stream);

/* ---------------------------------------------------------------- synthetic data (bench / tests) */
/* dst[i] = (int(hash24(i)) - 2^23) * scale23 + offset ; see p2t_hip/synth.py (bit-identical). */
int p2t_fill_hash(void* dst, int64_t n, uint64_t add, uint64_t xorv, float scale23, float offset,
                  int dtype, p2t_stream stream);
int p2t_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, p2t_stream stream);
/* x[i] *= scalar[0], scalar on the device (chain-rule scaling of a gradient without a host sync). */
int p2t_scale_by_device_scalar(float* x, int64_t n, const float* scalar, p2t_stream stream);
/* dst[c, r] = src[r, c]  (row strides in elements) */
int p2t_transpose(const void* src, int64_t rows, int64_t cols, int64_t ld_src, void* dst, int64_t ld_dst,
                  int dtype, p2t_stream stream);

/* ---------------------------------------------------------------- building blocks */
/* C[M,N] = A[M,K] * W[N,K]^T with a fused epilogue.  A, W: `dtype`, K-contiguous, row strides
 * lda/ldw (elements, multiples of 8).  For dtype BF16 with K % 64 == 0 the MFMA kernel runs
 * (v_mfma_f32_16x16x32_bf16, 256x256x64 LDS tiles); otherwise the fp32-FMA kernel.  bias: f32 [N] or NULL.
 * out: `out_dtype` [M, ldc]; columns N..min(ldc, N rounded up to 64)-1 are written as zeros (they are the zero
 * K-padding of the next GEMM).  z (GELU only, may be NULL): pre-activation, out_dtype, same stride.  EPI_RESID: out is f32 and accumulated in place.  EPI_SWIGLU: N is the
 * interleaved gate/up row count (a multiple of 64), out has N/2 columns.  use_mfma: -1 auto, 0 force FMA kernel, 1 require MFMA.
 * fix_ws (optional, p2t_gemm_fix_workspace_bytes() bytes): lets the MFMA kernel run the tiles of a last partial
 * "round" of the 256 CUs as two concurrent K halves (split-K fix-up).  Its first 2048 bytes (the tiles' flag words)
 * must have been zeroed (once) before a sequence of calls that pass strictly increasing fix_epoch values >= 1.
 * A consumer that gives up waiting sets the GPU's fault word (see p2t_fault_status) and poisons its tile with NaN. */
int p2t_gemm_nt(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldc,
                void* z, int64_t M, int64_t N, int64_t K, int dtype, int out_dtype, int epilogue, int accumulate,
                int use_mfma, void* fix_ws, size_t fix_ws_bytes, unsigned fix_epoch, p2t_stream stream);
size_t p2t_gemm_fix_workspace_bytes(void);
/* ---- fp8 GEMM operands (BASELINE.json configs[4]: "fp8 weights, CDNA4 fp8 MFMA") ----
 * Row-wise quantisation: x `dtype` (F32 / BF16) [rows, ld_x] -> q: OCP e4m3fn bytes [rows, ld_q] (columns cols..ld_q-1 zero;
 * ld_q a multiple of 8, of 128 when q feeds p2t_gemm_nt_fp8) and scale: one E8M0 byte per row, E = 127 + e with 2^e the
 * smallest power of two such that amax(row) / 2^e <= 448; q = e4m3_rne(x * 2^-e).  An all-zero row gets E = 127. */
int p2t_quant_rows_fp8(const void* x, int dtype, int64_t ld_x, int64_t rows, int64_t cols, void* q, int64_t ld_q,
                       uint8_t* scale, p2t_stream stream);
/* torch.nn.LayerNorm / LlamaRMSNorm of the f32 stream written directly in that format (no bf16 intermediate).
 * bound_scale (LayerNorm only, may be NULL): E8M0 byte per row of the smallest power of two >= (||y_row||_2 * bound_w +
 * bound_b) / 448, y the normalised row -- with bound_w >= max_n ||W_n||_2 and bound_b >= max_n |bias_n| of the projection
 * that consumes y, a valid scale for every element of gelu(y W^T + bias) (Cauchy-Schwarz; |gelu(z)| <= |z|): the row scale
 * P2T_EPI_GELU_FP8 needs BEFORE its GEMM runs. */
int p2t_layernorm_fp8(const float* x, int64_t ld_x, const float* w, const float* b, float eps, void* q, int64_t ld_q,
                      uint8_t* scale, int64_t rows, int64_t cols, float bound_w, float bound_b, uint8_t* bound_scale,
                      p2t_stream stream);
int p2t_rmsnorm_fp8(const float* x, int64_t ld_x, const float* w, float eps, void* q, int64_t ld_q, uint8_t* scale,
                    int64_t rows, int64_t cols, p2t_stream stream);
/* C[M,N] = (A8 * 2^(a_scale-127))[M,K] * (W8 * 2^(w_scale-127))[N,K]^T on v_mfma_scale_f32_16x16x128_f8f6f4 (both scales
 * applied by the instruction), fp32 accumulate, the epilogues of p2t_gemm_nt (all but GELU_BWD).  A8 / W8: e4m3 bytes,
 * row strides lda / ldw in BYTES (multiples of 16), K % 128 == 0 with the padding zeroed; a_scale [M], w_scale [N] E8M0
 * bytes.  tile: 0 auto (the persistent four-wave kernel when the shape has no edge tiles, K % 256 == 0 and at least one tile per
 * CU; else the per-tile kernel), 4 = four-wave kernel required, 128 / 256 = per-tile kernel of that tile height.  P2T_EPI_GELU_FP8: out is e4m3 bytes [M, ldc bytes] (columns N up to
 * the next multiple of 128 zeroed), out_row_scale the E8M0 byte of every output row (an input: see p2t_layernorm_fp8). */
int p2t_gemm_nt_fp8(const void* A, int64_t lda, const uint8_t* a_scale, const void* W, int64_t ldw, const uint8_t* w_scale,
                    const float* bias, void* out, int64_t ldc, void* z, int64_t M, int64_t N, int64_t K, int out_dtype,
                    int epilogue, int accumulate, int tile, const uint8_t* out_row_scale, p2t_stream stream);

/* The QKV projection as the towers launch it for head_dim 64 / 128 (P2T_EPI_QKV_ROPE): acc + bias, query * q_scale
 * BEFORE the rotation (HF EsmSelfAttention.forward, modeling_esm.py:345,362-378: q_scale = head_dim^-1/2 and SDPA scale 1;
 * Llama: q_scale 1, no bias, modeling_llama.py:254-259), rotate-half rotary with the table built from inv_freq
 * (f32 [head_dim/2]; cos_sin_scratch f32 [seq * head_dim]), head split.  A `dtype` [M = B * seq, lda]; W `dtype`
 * [(nh + 2 nkv) * head_dim, ldw], rows q heads | k heads | v heads, for head_dim 128 in the packed per-head row order of
 * p2t_llama_layer; bias f32 or NULL.  Outputs `dtype`: q [B, nh, seq, head_dim], k and v [B, nkv, seq, head_dim].
 * Exposed so that the fused epilogue can be checked on its own against the reference arithmetic. */
int p2t_gemm_qkv_rope(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int64_t M, int64_t K,
                      int dtype, const float* inv_freq, float* cos_sin_scratch, void* q, void* k, void* v, int seq, int nh,
                      int nkv, int head_dim, float q_scale, int use_mfma, void* fix_ws, size_t fix_ws_bytes,
                      unsigned fix_epoch, p2t_stream stream);

/* torch.nn.LayerNorm over the last dim: x f32 [rows, ld_x] -> y `out_dtype` [rows, ld_y]; pad zeroed. */
int p2t_layernorm(const float* x, int64_t ld_x, const float* w, const float* b, float eps, void* y, int64_t ld_y,
                  int64_t rows, int64_t cols, int out_dtype, p2t_stream stream);
/* LlamaRMSNorm (fp32 variance, weight last). */
int p2t_rmsnorm(const float* x, int64_t ld_x, const float* w, float eps, void* y, int64_t ld_y, int64_t rows,
                int64_t cols, int out_dtype, p2t_stream stream);

/* mask int64 [B, T] -> key_mask u8 [B, T]; kv_info i32 [2B]: [b] = 1 + last valid key, [B + b] = 1 if the mask
 * is a plain prefix (the right-padded batch contract); emb_scale f32 [2B] (optional, ESM token-dropout rescale:
 * numerator, denominator; needs ids). */
int p2t_mask_prepare(const int64_t* ids, const int64_t* mask, int B, int T, int mask_id, int token_dropout,
                     uint8_t* key_mask, int32_t* kv_info, float* emb_scale, p2t_stream stream);
/* Head split + query scale + rotary.  qkv `dtype` [B*T, ldq] rows = [q heads | k heads | v heads];
 * inv_freq f32 [d/2]; cos_sin_scratch f32 [T * d].  Outputs q [B, nh, T, dp], k and v [B, nkv, T, dp], head dim
 * zero-padded to dp in {32, 64, 128}.  (For head_dim 64 the towers fuse this pass into the QKV GEMM epilogue.) */
int p2t_qkv_post(const void* qkv, int64_t ldq, const float* inv_freq, float* cos_sin_scratch, void* q, void* k,
                 void* v, int B, int T, int nh, int nkv, int d, int dp, float q_scale, int dtype, p2t_stream stream);

/* Attention.  q [B, nh, T, dp], k and v [B, nkv, T, dp], all `dtype`, row-major (the MFMA kernel transposes V with
 * ds_read_b64_tr_b16).  key_mask u8 [B, T] (1 = valid), kv_info i32 [2B] from p2t_mask_prepare.
 * softmax(scale * q k^T + mask) v -> out [B*T, ld_out] `dtype`, head h in columns h*d..h*d+d-1; columns nh*d up to the
 * next multiple of 64 (the o-proj K padding) zeroed.  causal: also require key <= query.
 * log2_scores != 0: q was stored pre-multiplied by scale * log2(e) (what the towers do for bf16 models, through the q_scale
 * of p2t_gemm_qkv_rope / p2t_qkv_post: q is rounded to bf16 once either way), so q k^T is the base-2 exponent itself:
 * `scale` is ignored, the result is softmax_2(q k^T + mask) v = the same attention, one multiply-add less per score.
 * use_mfma: -1 auto / 1 require = the MFMA kernels (bf16): the hand-placed kernel (csrc/attn_fwd64.hip) where dp == 64,
 * d % 8 == 0 and log2_scores, else the general one (csrc/attn_mfma.hip); 2 = the general MFMA kernel everywhere;
 * 3 = require the hand-placed one; 0 = the exact fp32-softmax kernel. */
int p2t_attention(const void* q, const void* k, const void* v, const uint8_t* key_mask, const int32_t* kv_info,
                  void* out, int64_t ld_out, int B, int T, int nh, int nkv, int d, int dp, float scale, int causal,
                  int dtype, int use_mfma, int log2_scores, float* lse, p2t_stream stream);
/* lse (may be NULL): f32 [B, nh, T], the natural-log sum-exp of every query row's effective logits (scale * q k^T, or
 * ln 2 * q k^T with log2_scores), +inf for a row without a visible key -- what p2t_attention_backward rebuilds P from.
 *
 * Backward of the attention above (exact-fp32 arithmetic on `dtype` operands; stage-2 training, see
 * p2t_llama_train_backward): with P = exp(logits - lse), D = rowsum(dO o O), dS = P o (dO v^T - D):
 * dq [B, nh, T, dp] = c dS k, dk [B, nkv, T, dp] = c dS^T q, dv = P^T dO (f32; c = scale, or ln 2 with log2_scores;
 * GQA: the query heads of a group accumulate into their shared key / value head; padded head dims written as 0).
 * o / d_o: the attention output and its gradient in the [B*T, ld] layout of p2t_attention's `out`.
 * D_scratch: f32 [B, nh, T].  use_mfma: -1 auto = the MFMA kernels (csrc/attn_bwd_mfma.hip: bf16, log2_scores, head_dim
 * 64 / 128; seven MFMA products per tile pair, no float atomics) where they apply, else the exact kernels; 0 = exact; 1 = require. */
int p2t_attention_backward(const void* q, const void* k, const void* v, const void* o, int64_t ld_o, const void* d_o,
                           int64_t ld_do, const float* lse, const uint8_t* key_mask, const int32_t* kv_info, float* dq,
                           float* dk, float* dv, float* D_scratch, int B, int T, int nh, int nkv, int d, int dp,
                           float scale, int causal, int dtype, int log2_scores, int use_mfma, p2t_stream stream);

/* ---------------------------------------------------------------- ESM2 encoder */
typedef struct {
    int32_t n_layers, hidden, ffn, heads, head_dim, vocab;
    int32_t pad_id, mask_id, token_dropout, emb_layer_norm_before;
    float layer_norm_eps, rope_theta;
    int32_t dtype;                       /* P2T_F32 | P2T_BF16: activations, attention, vectors */
    int32_t gemm_fp8;                    /* 1: the four projections run on the fp8 MFMA kernel (needs dtype BF16): the
                                            matrices are e4m3 bytes with one E8M0 scale per row, see p2t_esm2_layer */
} p2t_esm2_config;

/* Packed per-layer weights (device).  Matrices are `dtype`, row-major [N][ld], ld = K rounded up
 * to 64, zero padded.  qkv_w rows: query (H), key (H), value (H).  Vectors are f32. */
typedef struct {
    const void* qkv_w;  const float* qkv_b;
    const void* o_w;    const float* o_b;
    const float* ln1_w; const float* ln1_b;          /* attention.LayerNorm */
    const void* fc1_w;  const float* fc1_b;          /* intermediate.dense  */
    const void* fc2_w;  const float* fc2_b;          /* output.dense        */
    const float* ln2_w; const float* ln2_b;          /* layer LayerNorm     */
    /* gemm_fp8 only: the four matrices above are then e4m3 bytes, row-major [N][ld], ld = K rounded up to 128 (zero
     * padded), quantised with p2t_quant_rows_fp8, and these are their E8M0 row scales [N] */
    const uint8_t* qkv_ws; const uint8_t* o_ws; const uint8_t* fc1_ws; const uint8_t* fc2_ws;
    /* gemm_fp8, optional (0 = quantise the GELU output in a separate pass): upper bounds of max_n ||fc1_w[n]||_2 (of the
     * weights the GEMM multiplies with, i.e. after quantisation) and of max_n |fc1_b[n]|; with them the FFN-up GEMM writes
     * its GELU output directly as e4m3 under the per-token bound scale of p2t_layernorm_fp8 */
    float fc1_wnorm_bound, fc1_babs_bound;
} p2t_esm2_layer;

typedef struct {
    const void* word_emb;                            /* [vocab][hidden] dtype, unpadded */
    const float* emb_ln_w; const float* emb_ln_b;    /* only if emb_layer_norm_before */
    const p2t_esm2_layer* layers;                    /* HOST array of n_layers structs */
    const float* final_ln_w; const float* final_ln_b;
    const float* inv_freq;                           /* rotary_embeddings.inv_freq f32 [head_dim/2] (a checkpoint
                                                        buffer in HF); NULL = 1/theta^(2j/d) computed on device */
} p2t_esm2_weights;

size_t p2t_esm2_workspace_bytes(const p2t_esm2_config* cfg, int B, int T);
/* ids, mask: int64 [B, T] (right-padded batch contract, dataset/dataloader.py:113-123).
 * out: last_hidden_state, `dtype` [B*T, ld_out] (ld_out >= hidden; pad columns zeroed). */
int p2t_esm2_forward(const p2t_esm2_config* cfg, const p2t_esm2_weights* w, const int64_t* ids,
                     const int64_t* mask, int B, int T, void* out, int64_t ld_out, void* workspace,
                     size_t workspace_bytes, p2t_stream stream);

/* ---------------------------------------------------------------- Llama text tower */
typedef struct {
    int32_t n_layers, hidden, ffn, heads, kv_heads, head_dim, vocab;
    float rms_norm_eps, rope_theta;
    int32_t rope_llama3;                 /* 0 default, 1 llama3 scaling */
    float rope_factor, rope_low_freq_factor, rope_high_freq_factor;
    int32_t rope_original_max_pos;
    int32_t dtype;
    int32_t gemm_fp8;                    /* as p2t_esm2_config.gemm_fp8 */
} p2t_llama_config;

/* qkv_w rows: q (heads*d), k (kv*d), v (kv*d); for head_dim 128 the 128 rows of EVERY head are stored in the order
 * 0..31, 64..95, 32..63, 96..127 (the rotary partners j, j+64 then sit 32 rows apart inside a 64-row block, which is what
 * the fused QKV + RoPE epilogue pairs up); natural order for every other head_dim.  gu_w: gate/up interleaved in blocks of 32 rows
 * (rows 64j..64j+31 = gate features 32j.., rows 64j+32..64j+63 = up features 32j..).  No biases. */
typedef struct {
    const void* qkv_w; const void* o_w; const void* gu_w; const void* down_w;
    const float* ln1_w;                              /* input_layernorm */
    const float* ln2_w;                              /* post_attention_layernorm */
    const uint8_t* qkv_ws; const uint8_t* o_ws; const uint8_t* gu_ws; const uint8_t* down_ws;   /* gemm_fp8: E8M0 row scales */
    /* Qwen3 (both or neither): self_attn.q_norm / k_norm weights, f32 [head_dim] -- RMSNorm over head_dim of every query /
     * key head after the projection and before the rotation (HF Qwen3Attention.forward).  qkv_w rows are then in natural order. */
    const float* q_norm_w; const float* k_norm_w;
} p2t_llama_layer;

typedef struct {
    const void* embed;                               /* [vocab][hidden] dtype */
    const p2t_llama_layer* layers;                   /* HOST array, at least the first k layers */
    const float* final_norm_w;
    const float* inv_freq;                           /* f32 [head_dim/2] or NULL = computed from the config */
} p2t_llama_weights;

size_t p2t_llama_workspace_bytes(const p2t_llama_config* cfg, int B, int T);
/* hidden_states[k]: residual stream after k layers (k < n_layers) or the post-norm output (k == n_layers).
 * out: f32 [B*T, hidden]. */
int p2t_llama_hidden_forward(const p2t_llama_config* cfg, const p2t_llama_weights* w, const int64_t* ids,
                             const int64_t* mask, int B, int T, int k, float* out, void* workspace,
                             size_t workspace_bytes, p2t_stream stream);

/* Same tower from caller-supplied layer-0 inputs (f32 [B*T, hidden]) instead of token ids: the decoder half of
 * Esm2LlamaInstructForCausalLM.forward, models/modeling_esm2llama_instruct.py:204-215 (`inputs_embeds=...`). */
int p2t_llama_hidden_forward_embeds(const p2t_llama_config* cfg, const p2t_llama_weights* w, const float* inputs_embeds,
                                    const int64_t* mask, int B, int T, int k, float* out, void* workspace,
                                    size_t workspace_bytes, p2t_stream stream);
/* llama_decoder.get_input_embeddings()(input_ids), models/modeling_esm2llama_instruct.py:134: f32 [n_tokens, hidden]. */
int p2t_llama_embed_tokens(const p2t_llama_config* cfg, const p2t_llama_weights* w, const int64_t* ids, int64_t n_tokens,
                           float* out, p2t_stream stream);

/* ---------------------------------------------------------------- decoder inputs + LM loss (SFT forward) */
/* Flat positions, in array order, of the elements with value == match (mode 0) or value != 0 (mode 1): the order in
 * which torch's boolean-mask indexing enumerates them (`input_ids == placeholder_id`, `encoder_attention_mask.bool()`,
 * models/modeling_esm2llama_instruct.py:136-137).  pos: int32 [n] (capacity), count: int32 [1]; both on the device. */
int p2t_positions_where(const int64_t* values, int64_t n, int mode, int64_t match, int32_t* pos, int32_t* count,
                        p2t_stream stream);
/* dst[dst_pos[r], :H] = src[src_pos[r], :H] for r < min(*n_dst, *n_src)  (`inputs_embeds[placeholder_mask] =
 * encoder_hidden_states[encoder_mask]`, :138).  dst f32, src `src_dtype`; counts are read on the device (no host sync;
 * the caller compares them afterwards to raise torch's shape-mismatch error). */
int p2t_scatter_rows(float* dst, int64_t ld_dst, const int32_t* dst_pos, const void* src, int64_t ld_src, int src_dtype,
                     const int32_t* src_pos, const int32_t* n_dst, const int32_t* n_src, int64_t max_rows, int H,
                     p2t_stream stream);
/* HF causal-LM loss (transformers loss_utils.ForCausalLMLoss as called by LlamaForCausalLM.forward with labels): logits
 * `dtype` [B*T, ld >= V]; position (b, t) predicts labels[b, t+1]; targets equal to ignore_index (and out-of-range ones)
 * are skipped; loss = mean over the rest in f32 (NaN when there is none).  row_loss f32 [B*T], row_valid int32 [B*T]
 * are outputs as well (deterministic two-stage reduction); count may be NULL. */
int p2t_cross_entropy_shifted(const void* logits, int64_t ld, int dtype, const int64_t* labels, int B, int T, int V,
                              int64_t ignore_index, float* row_loss, int32_t* row_valid, float* loss, int32_t* count,
                              p2t_stream stream);

/* ---------------------------------------------------------------- stage-2 training through the frozen decoder */
/* The reference's stage-2 step is `loss = model(**batch).loss; loss.backward()` (scripts/train_instruct.py:192-213) on
 * Esm2LlamaInstructForCausalLM.forward (models/modeling_esm2llama_instruct.py:195-215).  With the decoder frozen the
 * backward is a chain of dX operations from the LM loss to `inputs_embeds`, whose placeholder rows are the adapter's
 * outputs (:138) -- the adapter is then differentiated by p2t_adapter_backward.  No LoRA matrices yet.
 *
 * d loss / d logits of p2t_cross_entropy_shifted (count: its device-side counter): rows with a counted target get
 * (softmax - onehot) / count, all others 0; d_logits `dtype` [B*T, ld_d >= V], columns V .. next multiple of 64 zeroed
 * (the K padding of the LM-head dX GEMM). */
int p2t_cross_entropy_shifted_backward(const void* logits, int64_t ld, int dtype, const int64_t* labels, int B, int T,
                                       int V, int64_t ignore_index, const int32_t* count, void* d_logits, int64_t ld_d,
                                       p2t_stream stream);
/* LlamaRMSNorm backward (modeling_llama.py:62-67): dx (+)= r (w dy) - x r^3 mean(w dy x), r = rsqrt(mean(x^2) + eps);
 * x, dx f32, dy `dy_dtype` (F32 / BF16); accumulate != 0 adds into dx (the residual-stream gradient). */
int p2t_rmsnorm_backward(const float* x, int64_t ld_x, const float* w, float eps, const void* dy, int64_t ld_dy,
                         int dy_dtype, float* dx, int64_t ld_dx, int64_t rows, int64_t cols, int accumulate,
                         p2t_stream stream);
/* Pieces of the stage-2 step exposed one by one, for the LoRA form of that step (scripts/train_instruct.py:146-183; p2t_hip/
 * decoder_train.py drives them per layer -- the low-rank branches sit BETWEEN the fused blocks of p2t_llama_train_forward):
 * SwiGLU on the interleaved gate / up pre-activations of p2t_llama_layer.gu_w (64-column block j = gate[32 j..] | up[32 j..]):
 * d_act == NULL: out[m, f] = silu(g) u (out `dtype` [M, ld_out >= F], zeros up to the next multiple of 64);
 * else the backward: out = d_gu in the same interleaved layout [M, 2F] from d_act [M, ld_da]. */
int p2t_swiglu_gu(const void* gu, int64_t ld_gu, const void* d_act, int64_t ld_da, void* out, int64_t ld_out, int64_t M, int64_t F,
                  int dtype, p2t_stream stream);
/* Backward of the head split + rotation (+ q_scale folded into q) of p2t_qkv_post: dq [B, nh, T, dp], dk, dv [B, nkv, T, dp] f32
 * -> d_qkv `dtype` [B*T, ld], rows = [q heads | k heads | v heads] in the natural order of q_proj / k_proj / v_proj.
 * cos_sin_scratch f32 [T * d]. */
int p2t_rope_backward_pack(const float* dq, const float* dk, const float* dv, const float* inv_freq, float* cos_sin_scratch,
                           void* d_qkv, int64_t ld, int B, int T, int nh, int nkv, int d, int dp, float q_scale, int dtype,
                           p2t_stream stream);
/* dst[m, c] (+)= keep(seed, m K + c) ? src[m, c] / (1 - p) : 0, c < K (F32 / BF16 either side): the input dropout of a LoRA
 * branch (`lora_dropout`) and, called again with the same seed on the gradient, its backward -- the mask is the counter-hash
 * of (seed, element), regenerated, never stored.  p == 0: a (converting, optionally accumulating) copy. */
int p2t_dropout_rows(const void* src, int src_dtype, int64_t ld_src, void* dst, int dst_dtype, int64_t ld_dst, int64_t M,
                     int64_t K, float p, uint64_t seed, int accumulate, p2t_stream stream);
/* dst[dst_pos[r], :H] = src[src_pos[r], :H], r < min(*n_dst, *n_src), f32: the backward of p2t_scatter_rows (the
 * gradient of `inputs_embeds[placeholder_mask] = encoder_hidden_states[encoder_mask]` with respect to the encoder
 * states: call it with the two position lists swapped; rows not listed keep their contents -- zero dst first). */
int p2t_gather_rows_f32(float* dst, int64_t ld_dst, const int32_t* dst_pos, const float* src, int64_t ld_src,
                        const int32_t* src_pos, const int32_t* n_dst, const int32_t* n_src, int64_t max_rows, int H,
                        p2t_stream stream);
/* Transposed decoder weights for the dX GEMMs, `dtype`, one struct per layer (HOST array): each matrix is the forward
 * weight transposed -- [forward K rows][row stride = forward N rounded up to 64, zero padded] -- built once (the decoder is
 * frozen).  qkv_wT from the NATURAL row order q_proj | k_proj | v_proj (not the packed order of p2t_llama_layer.qkv_w);
 * gu_wT from the interleaved gate / up order of p2t_llama_layer.gu_w. */
typedef struct p2t_llama_layer_t {
    const void* qkv_wT;     /* [hidden, >= (heads + 2 kv_heads) head_dim] */
    const void* o_wT;       /* [heads head_dim, >= hidden] */
    const void* gu_wT;      /* [hidden, >= 2 ffn] */
    const void* down_wT;    /* [ffn, >= hidden] */
} p2t_llama_layer_t;
/* Bytes of the activation "tape" one training forward writes and the backward reads (all n_layers: residual streams
 * before / inside every layer, rotated heads, attention outputs and log-sum-exps, gate / up pre-activations -- about
 * 170 KB per token and layer for Llama-3.1-8B: sized for 288 GB of HBM, nothing is recomputed), and of the transient
 * workspace both calls share. */
size_t p2t_llama_tape_bytes(const p2t_llama_config* cfg, int B, int T);
size_t p2t_llama_train_workspace_bytes(const p2t_llama_config* cfg, int B, int T);
/* p2t_llama_hidden_forward_embeds(k = n_layers) that also fills the tape.  out: f32 [B*T, hidden], post final RMSNorm. */
int p2t_llama_train_forward(const p2t_llama_config* cfg, const p2t_llama_weights* w, const float* inputs_embeds,
                            const int64_t* mask, int B, int T, float* out, void* tape, size_t tape_bytes, void* workspace,
                            size_t workspace_bytes, p2t_stream stream);
/* d_out: f32 [B*T, hidden] = d loss / d (post-norm hidden states), e.g. d_logits . lm_head.  d_inputs_embeds: f32
 * [B*T, hidden], overwritten.  The fp32 residual-stream gradient is accumulated in place in d_inputs_embeds; GEMM
 * operands are `dtype`; dtype BF16 uses the MFMA GEMMs, F32 the exact kernels (parity mode). */
int p2t_llama_train_backward(const p2t_llama_config* cfg, const p2t_llama_weights* w, const p2t_llama_layer_t* wT,
                             const int64_t* mask, int B, int T, const float* d_out, const void* tape, size_t tape_bytes,
                             float* d_inputs_embeds, void* workspace, size_t workspace_bytes, p2t_stream stream);

/* ---------------------------------------------------------------- ModalityAdapter */
typedef struct {
    int32_t input_dim, intermediate_dim, output_dim;
    float dropout_p;                     /* 0 in eval */
    uint64_t dropout_seed;
    int32_t dtype;
} p2t_adapter_config;

typedef struct {                          /* matrices `dtype` [N][ld = K up to 64]; biases f32 */
    const void* fc1_w; const float* fc1_b; const void* fc2_w; const float* fc2_b;
} p2t_adapter_weights;

/* Activation buffers (caller-allocated, M = rows): z1,h1 [M, ld(intermediate)], z2,g2 [M, ld(output)] `dtype`,
 * ld(n) = n rounded up to 64; inv_norm f32 [M].  h1 and g2 are always required (they are the forward's
 * intermediates); z1, z2, inv_norm may be NULL for inference and are required by p2t_adapter_backward. */
typedef struct { void* z1; void* h1; void* z2; void* g2; float* inv_norm; } p2t_adapter_saved;

/* x `dtype` [M, ld_x] -> y `dtype` [M, ld(output_dim)], rows L2-normalised. */
int p2t_adapter_forward(const p2t_adapter_config* cfg, const p2t_adapter_weights* w, const void* x, int64_t ld_x,
                        int64_t M, void* y, const p2t_adapter_saved* save, p2t_stream stream);
size_t p2t_adapter_backward_workspace_bytes(const p2t_adapter_config* cfg, int64_t M);
/* dy f32 [M, output_dim] -> f32 gradients (accumulated if accumulate != 0): d_fc1_w [I, input_dim],
 * d_fc1_b [I], d_fc2_w [O, I], d_fc2_b [O] (unpadded, contiguous). */
int p2t_adapter_backward(const p2t_adapter_config* cfg, const p2t_adapter_weights* w, const void* x, int64_t ld_x,
                         int64_t M, const p2t_adapter_saved* saved, const float* dy, float* d_fc1_w, float* d_fc1_b,
                         float* d_fc2_w, float* d_fc2_b, int accumulate, void* workspace, size_t workspace_bytes,
                         p2t_stream stream);

/* ---------------------------------------------------------------- readout / normalise / loss */
/* emb [B, T, ld] (`dtype`, or f32), mask int64 [B, T] (NULL = all ones) -> out f32 [B, D] (mean/std/last)
 * or [B, 2D] (mix).  D and ld must be multiples of 4 (vector loads). */
int p2t_readout(const void* emb, int dtype, int64_t ld, const int64_t* mask, int B, int T, int D, int mode,
                float* out, p2t_stream stream);
/* d_out f32 [B, D or 2D] -> d_emb f32 [B, T, D]. */
int p2t_readout_backward(const void* emb, int dtype, int64_t ld, const int64_t* mask, int B, int T, int D, int mode,
                         const float* pooled, const float* d_out, float* d_emb, p2t_stream stream);
/* y = x / max(||x||, eps) per row; inv_norm (optional) f32 [rows]. */
int p2t_l2norm_rows(const float* x, float* y, float* inv_norm, int64_t rows, int64_t cols, float eps,
                    p2t_stream stream);
int p2t_l2norm_rows_backward(const float* x, const float* dy, float* dx, int64_t rows, int64_t cols, float eps,
                             p2t_stream stream);
/* Row-wise InfoNCE of `seg` [S, D] against `batch` [N, D] (both f32, L2-normalised), labels i32 [S]:
 * loss_out[0] (+)= weight * mean_i( logsumexp_j(l_ij) - l_i,label_i ), l = seg batch^T / temperature.
 * logits f32 [S, N] and row_loss f32 [S] are outputs / scratch (required). */
int p2t_infonce_forward(const float* seg, const float* batch, const int32_t* labels, int S, int N, int D,
                        float temperature, float weight, int accumulate, float* loss_out, float* logits,
                        float* row_loss, p2t_stream stream);
/* d_seg f32 [S, D] = weight * d loss / d seg, from the logits the forward wrote. */
int p2t_infonce_backward(const float* batch, const int32_t* labels, const float* logits, int S, int N, int D,
                         float temperature, float weight, float* d_seg, p2t_stream stream);

/* Optional column (text -> protein) term -- not in the reference loop, which is row-only (train_contrast.py:94-114);
 * its arithmetic is the reference's own BatchInfoNCELoss with the arguments swapped (:72-91), i.e.
 * F.cross_entropy(logits^T, arange).  p_all, t_all: f32 [N, D], the L2-normalised protein / text embeddings of the GLOBAL
 * batch (all-gathered), positives on the diagonal.  col_lse[j] = logsumexp_i(l_ij) for every column j (output, f32 [N]);
 * loss_out[0] (+)= weight * mean of (col_lse_j - l_jj) over `count` columns: j = cols[i] (i32 [count], device) when cols is
 * given, else j = first .. first+count-1 (this rank's / this segment's own columns).
 * scratch_logits f32 [N, N], scratch_col_loss f32 [N]. */
int p2t_infonce_col_forward(const float* p_all, const float* t_all, int N, int D, float temperature, const int32_t* cols,
                            int first, int count, float weight, int accumulate, float* loss_out, float* col_lse,
                            float* scratch_logits, float* scratch_col_loss, p2t_stream stream);
/* Gradient of sum_j (col_lse_j - l_jj) with respect to the S protein rows whose row logits [S, N] the row forward wrote:
 * d_seg[i] (+)= scale / temperature * sum_j (exp(l_ij - col_lse_j) - [j == labels_i]) t_all[j]. */
int p2t_infonce_col_backward(const float* t_all, const int32_t* labels, const float* logits, const float* col_lse, int S,
                             int N, int D, float temperature, float scale, int accumulate, float* d_seg, p2t_stream stream);

/* ---------------------------------------------------------------- generation: KV cache, prefill, decode steps */
/* What `llama_decoder.generate(inputs_embeds=..., attention_mask=..., **kwargs)` runs under
 * Esm2LlamaInstructForCausalLM.generate (models/modeling_esm2llama_instruct.py:217-251): HF GenerationMixin over the cached
 * LlamaAttention path.  Here: the prompt rows are compacted (valid tokens first: positions 0..len-1 = HF's
 * cumsum(attention_mask) - 1 on the tokens under the mask), one prefill writes the PROMPT segment of the cache, and every
 * decode step appends one token per row to the GENERATED segment.  BB = B0 * group rows generate in lock-step (group = beams
 * per prompt; the beams of a prompt share its prompt segment).  All buffers are the caller's; capacities Tp and G are
 * multiples of 64; dp = head_dim rounded up to 32 / 64 / 128.  Keys are stored after the rotation, values transposed. */
typedef struct {
    void* k_prompt;  void* vt_prompt;   /* `dtype` [n_layers][B0][kv_heads][Tp][dp]  /  [n_layers][B0][kv_heads][dp][Tp] */
    void* k_gen;     void* vt_gen;      /* `dtype` [n_layers][BB][kv_heads][G][dp]   /  [n_layers][BB][kv_heads][dp][G]  */
    const int32_t* prompt_len;          /* device i32 [B0]: valid prompt tokens per row (p2t_compact_rows' lens) */
    int32_t* step;                      /* device i32 [1]: generated tokens already in the cache; a decode step appends at
                                           index step[0], attends to prompt_len + step[0] + 1 keys and increments it last */
    int32_t B0, group, Tp, G;
} p2t_kv_cache;

/* Stable partition of every row by its mask: out[b, r] = x[b, t_r] for the r-th token with mask != 0, rows >= lens[b] zero;
 * out_mask[b, r] = r < lens[b].  x, out: f32 [B, T, H] (H % 4 == 0, out != x); scratch: i32 [B * T]. */
int p2t_compact_rows(const float* x, const int64_t* mask, int B, int T, int H, float* out, int64_t* out_mask, int32_t* lens,
                     int32_t* scratch, p2t_stream stream);
size_t p2t_llama_prefill_workspace_bytes(const p2t_llama_config* cfg, int B, int T);
/* All layers over the compacted prompts (mask: prefix masks, as p2t_compact_rows writes them; cache->prompt_len = its lens),
 * every layer's rotated keys / values into the prompt segment; last_hidden f32 [B, hidden] = the post-final-RMSNorm state of
 * each row's LAST valid token (the LM head of the first generated token reads it). */
int p2t_llama_prefill(const p2t_llama_config* cfg, const p2t_llama_weights* w, const float* inputs_embeds, const int64_t* mask,
                      int B, int T, const p2t_kv_cache* cache, float* last_hidden, void* workspace, size_t workspace_bytes,
                      p2t_stream stream);
size_t p2t_llama_decode_workspace_bytes(const p2t_llama_config* cfg, int BB, int Tp, int G);
/* Optional second copies of the four projection weights of a layer in the STREAM order of the decode GEMM (p2t_preshuffle_w;
 * bf16 models; built once per generate-capable model: 288 GB of HBM hold both layouts of an 8-B decoder many times over).
 * w_stream: HOST array of n_layers entries or NULL (the step then streams p2t_llama_layer's own matrices, 64-byte pieces of 16
 * rows per load instead of whole lines: 8-30 % slower per GEMM).  gemm_fp8 models: the p2t_preshuffle_w_fp8 copies of the e4m3
 * matrices (their row scales stay p2t_llama_layer's *_ws). */
typedef struct { const void* qkv_w; const void* o_w; const void* gu_w; const void* down_w; } p2t_llama_layer_stream;
/* One token per row: x f32 [BB, hidden] (the embedding of the token chosen last) through all layers at position
 * prompt_len[row / group] + step[0], its keys / values appended at index step[0], final RMSNorm, LM head
 * (lm_head `dtype` [vocab, ld_head], ld_head >= hidden rounded up to 64, pad zero) -> logits `dtype` [BB, ld_logits]
 * (HF keeps the logits in the model dtype and up-casts the last position to f32: generation/utils.py `_sample`); step[0] += 1.
 * lm_head_preshuffled: lm_head is the p2t_preshuffle_w copy (ld_head ignored).
 * Every length is read on the device: the call can be captured into a HIP graph and replayed. */
enum { P2T_DECODE_NO_ROPE_FUSION = 1 };     /* flags: run the QKV projection and the rotation + cache append as two launches even where the
                                               fused epilogue applies (bf16, head_dim 64 / 128, no q/k norm): the same arithmetic, bit for bit */
int p2t_llama_decode_step(const p2t_llama_config* cfg, const p2t_llama_weights* w, const p2t_llama_layer_stream* w_stream,
                          const void* lm_head, int64_t ld_head, int lm_head_preshuffled, const p2t_kv_cache* cache,
                          const float* x, void* logits, int64_t ld_logits, int flags, void* workspace, size_t workspace_bytes,
                          p2t_stream stream);
/* Greedy choice with HF's finished-row rule: next[r] = finished[r] ? pad_id : argmax(logits[r, :V]) (lowest index among equal
 * maxima, torch.argmax), out_tokens[r, step[0]] = next[r], finished[r] |= next[r] in eos_ids.  eos_ids i64 [n_eos], finished
 * i32 [BB], next_tokens i64 [BB], out_tokens i64 [BB, ld_tokens >= G]: all on the device. */
int p2t_greedy_select(const void* logits, int dtype, int64_t ld, int V, int BB, const int64_t* eos_ids, int n_eos,
                      int64_t pad_id, int32_t* finished, int64_t* next_tokens, int64_t* out_tokens, int64_t ld_tokens,
                      const int32_t* step, int G, p2t_stream stream);
/* The decode step's projections on their own: out[M, N] = A[M, K] . W[N, K]^T for M <= 64 rows of bf16 -- a weight stream (every
 * byte of W read once, 8 waves split K per 16 output features, partial sums added in wave order; HBM-bound, not MFMA-bound).
 * K % 32 == 0, lda / ldw multiples of 8, no bias.  epilogue: P2T_EPI_STORE (out_dtype P2T_BF16 or P2T_F32), P2T_EPI_STORE_F32,
 * P2T_EPI_RESID (out f32, += ), P2T_EPI_SWIGLU (W = the gate/up interleave of p2t_llama_layer.gu_w with N = 2 F rows, out bf16
 * [M, F]).  w_preshuffled: W is the p2t_preshuffle_w copy (ldw ignored).  Anything else: P2T_ERR_UNSUPPORTED. */
int p2t_gemm_nt_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, int w_preshuffled, void* out, int64_t ldc, int64_t M,
                       int64_t N, int64_t K, int out_dtype, int epilogue, p2t_stream stream);
/* W bf16 [N, ldw] -> the stream order the decode GEMM reads with whole-line loads: out[((t * K/32 + s) * 64 + lane) * 8 + e] =
 * W[16 t + lane % 16][32 s + 8 (lane / 16) + e], rows N .. next multiple of 16 as zeros; out: bf16 [round_up(N, 16) * K].
 * K % 32 == 0.  Pass the result to p2t_gemm_nt_skinny with w_preshuffled = 1 (same N, K). */
int p2t_preshuffle_w(const void* W, int64_t ldw, int64_t N, int64_t K, void* out, p2t_stream stream);
/* The same stream on e4m3 operands (gemm_fp8 models, section "fp8 GEMMs" of DESIGN.md): A / W e4m3 bytes [rows, K rounded up to 128, zero
 * padded], one E8M0 scale byte per row each (p2t_quant_rows_fp8 / p2t_rmsnorm_fp8 write both for A); v_mfma_scale_f32_16x16x128_f8f6f4.
 * lda / ldw multiples of 16.  p2t_preshuffle_w_fp8: out[(((t * K/128 + s) * 2 + h) * 64 + lane) * 16 + b] = W[16 t + lane % 16][128 s + 64 h
 * + 16 (lane / 16) + b]; out: bytes [round_up(N, 16) * K]. */
int p2t_gemm_nt_skinny_fp8(const void* A, int64_t lda, const uint8_t* a_scale, const void* W, int64_t ldw, const uint8_t* w_scale,
                           int w_preshuffled, void* out, int64_t ldc, int64_t M, int64_t N, int64_t K, int out_dtype, int epilogue,
                           p2t_stream stream);
int p2t_preshuffle_w_fp8(const void* W, int64_t ldw, int64_t N, int64_t K, void* out, p2t_stream stream);
/* The decode step's attention on its own (exposed so that it can be checked against the plain arithmetic): ONE query token per
 * row, q `dtype` [BB, nh, dp], over prompt_len[row / group] keys of the prompt segment and step[0] + 1 keys of the generated
 * segment (the layouts of ONE layer of p2t_kv_cache), GQA; softmax((scale) q k^T) v with f32 statistics -- log2_scores: q holds
 * scale * log2(e) already (as the towers store it for bf16 models) and `scale` is ignored.  out `dtype` [BB, ld_out >= nh *
 * head_dim].  One block per (row, kv head) streams all its keys; its waves are merged in LDS in wave order (no atomics).  use_mfma: -1 / 1
 * = the matrix-pipe kernel for bf16 (what the decode step runs), 0 = the lane-per-key kernel (every dtype; f32 always). */
int p2t_attention_decode(const void* q, const void* k_prompt, const void* vt_prompt, const void* k_gen, const void* vt_gen,
                         const int32_t* prompt_len, const int32_t* step, int B0, int group, int nh, int nkv, int head_dim, int Tp,
                         int G, float scale, int log2_scores, int dtype, int use_mfma, void* out, int64_t ld_out, p2t_stream stream);
/* Beam re-ordering of the generated segment: row r of (k_dst, vt_dst) = row src_row[r] of the cache's generated segment, the
 * first step[0] tokens of every layer / head (k_dst, vt_dst: second buffers of the same shape; the caller swaps). */
int p2t_kv_reorder(const p2t_llama_config* cfg, const p2t_kv_cache* cache, const int64_t* src_row, void* k_dst, void* vt_dst,
                   p2t_stream stream);

/* ---------------------------------------------------------------- optimizer tail */
/* One clip_grad_norm_(max_norm) + AdamW step over n_tensors (<= 64) f32 parameter tensors (HOST arrays of
 * device pointers).  shadow[i] (optional, may be NULL per tensor) receives the updated parameter in
 * `shadow_dtype` as a matrix with cols[i] columns and row stride shadow_ld[i] (the GEMM-layout copy the
 * next forward reads).  max_norm <= 0 or >= 1e30 = no clipping.  grad_norm_out f32 [1] (device): total norm
 * before clipping.  scratch: f32 [256 * n_tensors]. */
int p2t_clip_adamw_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, const int64_t* numel, void* const* shadow, const int64_t* cols,
                        const int64_t* shadow_ld, int shadow_dtype, int step, double lr, double beta1, double beta2,
                        double eps, double weight_decay, double max_norm, float* grad_norm_out, float* scratch,
                        p2t_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* P2T_HIP_H */
