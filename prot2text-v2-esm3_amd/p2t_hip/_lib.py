"""ctypes binding of libp2t_hip.so (include/p2t_hip.h).  No torch types cross this boundary:
only integers, floats and raw device pointers (tensor.data_ptr()) plus the current HIP stream.

The product path has NO fallback: if the shared library is missing or a symbol is absent this
module raises at import time, and every non-zero return code raises with the library's message.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# P2T_HIP_LIB: load another build of the same ABI (kernel experiments); the default is the in-tree library
LIB_PATH = os.environ.get("P2T_HIP_LIB") or os.path.join(_HERE, "lib", "libp2t_hip.so")

F32, BF16, FP8 = 0, 1, 2
READOUT = {"last": 0, "mean": 1, "std": 2, "mix": 3}
EPI_STORE, EPI_GELU, EPI_RESID, EPI_SWIGLU, EPI_STORE_F32, EPI_GELU_BWD = range(6)
EPI_GELU_FP8 = 7


class P2TError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
        "or make -C prot2text-v2-esm3_amd/csrc).  There is no CPU fallback.")
lib = C.CDLL(LIB_PATH)

vp, i32, i64, u64, f32, f64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_size_t


class EsmConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_layers", "hidden", "ffn", "heads", "head_dim", "vocab", "pad_id", "mask_id",
                                        "token_dropout", "emb_layer_norm_before")] + \
               [("layer_norm_eps", f32), ("rope_theta", f32), ("dtype", C.c_int32), ("gemm_fp8", C.c_int32)]


class EsmLayerC(C.Structure):
    _fields_ = [(n, vp) for n in ("qkv_w", "qkv_b", "o_w", "o_b", "ln1_w", "ln1_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                                  "ln2_w", "ln2_b", "qkv_ws", "o_ws", "fc1_ws", "fc2_ws")] + \
               [("fc1_wnorm_bound", f32), ("fc1_babs_bound", f32)]


class EsmWeightsC(C.Structure):
    _fields_ = [("word_emb", vp), ("emb_ln_w", vp), ("emb_ln_b", vp), ("layers", C.POINTER(EsmLayerC)),
                ("final_ln_w", vp), ("final_ln_b", vp), ("inv_freq", vp)]


class LlamaConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_layers", "hidden", "ffn", "heads", "kv_heads", "head_dim", "vocab")] + \
               [("rms_norm_eps", f32), ("rope_theta", f32), ("rope_llama3", C.c_int32), ("rope_factor", f32),
                ("rope_low_freq_factor", f32), ("rope_high_freq_factor", f32), ("rope_original_max_pos", C.c_int32),
                ("dtype", C.c_int32), ("gemm_fp8", C.c_int32)]


class LlamaLayerC(C.Structure):
    _fields_ = [(n, vp) for n in ("qkv_w", "o_w", "gu_w", "down_w", "ln1_w", "ln2_w", "qkv_ws", "o_ws", "gu_ws", "down_ws",
                                  "q_norm_w", "k_norm_w")]


class LlamaWeightsC(C.Structure):
    _fields_ = [("embed", vp), ("layers", C.POINTER(LlamaLayerC)), ("final_norm_w", vp), ("inv_freq", vp)]


class LlamaLayerTC(C.Structure):
    _fields_ = [(n, vp) for n in ("qkv_wT", "o_wT", "gu_wT", "down_wT")]


class AdapterConfigC(C.Structure):
    _fields_ = [("input_dim", C.c_int32), ("intermediate_dim", C.c_int32), ("output_dim", C.c_int32),
                ("dropout_p", f32), ("dropout_seed", u64), ("dtype", C.c_int32)]


class AdapterWeightsC(C.Structure):
    _fields_ = [(n, vp) for n in ("fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class AdapterSavedC(C.Structure):
    _fields_ = [(n, vp) for n in ("z1", "h1", "z2", "g2", "inv_norm")]


class KvCacheC(C.Structure):
    _fields_ = [("k_prompt", vp), ("vt_prompt", vp), ("k_gen", vp), ("vt_gen", vp), ("prompt_len", vp), ("step", vp),
                ("B0", C.c_int32), ("group", C.c_int32), ("Tp", C.c_int32), ("G", C.c_int32)]


class LlamaLayerStreamC(C.Structure):
    _fields_ = [(n, vp) for n in ("qkv_w", "o_w", "gu_w", "down_w")]


_STRUCTS = [EsmConfigC, EsmLayerC, EsmWeightsC, LlamaConfigC, LlamaLayerC, LlamaWeightsC, AdapterConfigC,
            AdapterWeightsC, AdapterSavedC, LlamaLayerTC, KvCacheC, LlamaLayerStreamC]

# name -> (restype, argtypes); every symbol include/p2t_hip.h declares
SIGNATURES = {
    "p2t_version": (i32, []),
    "p2t_last_error": (C.c_char_p, []),
    "p2t_struct_size": (sz, [i32]),
    "p2t_prof_enable": (i32, [i32]),
    "p2t_set_gemm_policy": (i32, [i32]),
    "p2t_is_lab_build": (i32, []),
    "p2t_epoch_accumulate": (i32, [vp, vp, i32, vp, vp, vp]),
    "p2t_fault_status": (i32, [C.POINTER(C.c_uint), i32]),
    "p2t_fault_inject": (i32, [C.c_uint, vp]),
    "p2t_prof_collect": (i32, [C.POINTER(f64), C.POINTER(i64), C.POINTER(f64), i32]),
    "p2t_fill_hash": (i32, [vp, i64, u64, u64, f32, f32, i32, vp]),
    "p2t_cast": (i32, [vp, i32, vp, i32, i64, vp]),
    "p2t_scale_by_device_scalar": (i32, [vp, i64, vp, vp]),
    "p2t_transpose": (i32, [vp, i64, i64, i64, vp, i64, i32, vp]),
    "p2t_gemm_nt": (i32, [vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i64, i32, i32, i32, i32, i32, vp, sz, C.c_uint, vp]),
    "p2t_gemm_fix_workspace_bytes": (sz, []),
    "p2t_quant_rows_fp8": (i32, [vp, i32, i64, i64, i64, vp, i64, vp, vp]),
    "p2t_layernorm_fp8": (i32, [vp, i64, vp, vp, f32, vp, i64, vp, i64, i64, f32, f32, vp, vp]),
    "p2t_rmsnorm_fp8": (i32, [vp, i64, vp, f32, vp, i64, vp, i64, i64, vp]),
    "p2t_gemm_nt_fp8": (i32, [vp, i64, vp, vp, i64, vp, vp, vp, i64, vp, i64, i64, i64, i32, i32, i32, i32, vp, vp]),
    "p2t_gemm_qkv_rope": (i32, [vp, i64, vp, i64, vp, i64, i64, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, sz,
                                C.c_uint, vp]),
    "p2t_layernorm": (i32, [vp, i64, vp, vp, f32, vp, i64, i64, i64, i32, vp]),
    "p2t_rmsnorm": (i32, [vp, i64, vp, f32, vp, i64, i64, i64, i32, vp]),
    "p2t_mask_prepare": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "p2t_qkv_post": (i32, [vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, vp]),
    "p2t_attention": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, f32, i32, i32, i32, i32, vp, vp]),
    "p2t_attention_backward": (i32, [vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, i32, i32, i32, vp]),
    "p2t_cross_entropy_shifted_backward": (i32, [vp, i64, i32, vp, i32, i32, i32, i64, vp, vp, i64, vp]),
    "p2t_rmsnorm_backward": (i32, [vp, i64, vp, f32, vp, i64, i32, vp, i64, i64, i64, i32, vp]),
    "p2t_gather_rows_f32": (i32, [vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, vp]),
    "p2t_swiglu_gu": (i32, [vp, i64, vp, i64, vp, i64, i64, i64, i32, vp]),
    "p2t_rope_backward_pack": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, f32, i32, vp]),
    "p2t_dropout_rows": (i32, [vp, i32, i64, vp, i32, i64, i64, i64, f32, u64, i32, vp]),
    "p2t_compact_rows": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "p2t_llama_prefill_workspace_bytes": (sz, [C.POINTER(LlamaConfigC), i32, i32]),
    "p2t_llama_prefill": (i32, [C.POINTER(LlamaConfigC), C.POINTER(LlamaWeightsC), vp, vp, i32, i32, C.POINTER(KvCacheC), vp, vp, sz, vp]),
    "p2t_llama_decode_workspace_bytes": (sz, [C.POINTER(LlamaConfigC), i32, i32, i32]),
    "p2t_llama_decode_step": (i32, [C.POINTER(LlamaConfigC), C.POINTER(LlamaWeightsC), C.POINTER(LlamaLayerStreamC), vp, i64, i32, C.POINTER(KvCacheC), vp, vp, i64,
                              i32, vp, sz, vp]),
    "p2t_greedy_select": (i32, [vp, i32, i64, i32, i32, vp, i32, i64, vp, vp, vp, i64, vp, i32, vp]),
    "p2t_attention_decode": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, i32, i32, i32, vp, i64, vp]),
    "p2t_gemm_nt_skinny": (i32, [vp, i64, vp, i64, i32, vp, i64, i64, i64, i64, i32, i32, vp]),
    "p2t_preshuffle_w": (i32, [vp, i64, i64, i64, vp, vp]),
    "p2t_preshuffle_w_fp8": (i32, [vp, i64, i64, i64, vp, vp]),
    "p2t_gemm_nt_skinny_fp8": (i32, [vp, i64, vp, vp, i64, vp, i32, vp, i64, i64, i64, i64, i32, i32, vp]),
    "p2t_kv_reorder": (i32, [C.POINTER(LlamaConfigC), C.POINTER(KvCacheC), vp, vp, vp, vp]),
    "p2t_llama_tape_bytes": (sz, [C.POINTER(LlamaConfigC), i32, i32]),
    "p2t_llama_train_workspace_bytes": (sz, [C.POINTER(LlamaConfigC), i32, i32]),
    "p2t_llama_train_forward": (i32, [C.POINTER(LlamaConfigC), C.POINTER(LlamaWeightsC), vp, vp, i32, i32, vp, vp, sz, vp, sz, vp]),
    "p2t_llama_train_backward": (i32, [C.POINTER(LlamaConfigC), C.POINTER(LlamaWeightsC), C.POINTER(LlamaLayerTC), vp, i32, i32, vp, vp, sz,
                                       vp, vp, sz, vp]),
    "p2t_esm2_workspace_bytes": (sz, [C.POINTER(EsmConfigC), i32, i32]),
    "p2t_esm2_forward": (i32, [C.POINTER(EsmConfigC), C.POINTER(EsmWeightsC), vp, vp, i32, i32, vp, i64, vp, sz, vp]),
    "p2t_llama_workspace_bytes": (sz, [C.POINTER(LlamaConfigC), i32, i32]),
    "p2t_llama_hidden_forward": (i32, [C.POINTER(LlamaConfigC), C.POINTER(LlamaWeightsC), vp, vp, i32, i32, i32, vp, vp,
                                       sz, vp]),
    "p2t_llama_hidden_forward_embeds": (i32, [C.POINTER(LlamaConfigC), C.POINTER(LlamaWeightsC), vp, vp, i32, i32, i32, vp, vp,
                                              sz, vp]),
    "p2t_llama_embed_tokens": (i32, [C.POINTER(LlamaConfigC), C.POINTER(LlamaWeightsC), vp, i64, vp, vp]),
    "p2t_positions_where": (i32, [vp, i64, i32, i64, vp, vp, vp]),
    "p2t_scatter_rows": (i32, [vp, i64, vp, vp, i64, i32, vp, vp, vp, i64, i32, vp]),
    "p2t_cross_entropy_shifted": (i32, [vp, i64, i32, vp, i32, i32, i32, i64, vp, vp, vp, vp, vp]),
    "p2t_adapter_forward": (i32, [C.POINTER(AdapterConfigC), C.POINTER(AdapterWeightsC), vp, i64, i64, vp,
                                  C.POINTER(AdapterSavedC), vp]),
    "p2t_adapter_backward_workspace_bytes": (sz, [C.POINTER(AdapterConfigC), i64]),
    "p2t_adapter_backward": (i32, [C.POINTER(AdapterConfigC), C.POINTER(AdapterWeightsC), vp, i64, i64,
                                   C.POINTER(AdapterSavedC), vp, vp, vp, vp, vp, i32, vp, sz, vp]),
    "p2t_readout": (i32, [vp, i32, i64, vp, i32, i32, i32, i32, vp, vp]),
    "p2t_readout_backward": (i32, [vp, i32, i64, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "p2t_l2norm_rows": (i32, [vp, vp, vp, i64, i64, f32, vp]),
    "p2t_l2norm_rows_backward": (i32, [vp, vp, vp, i64, i64, f32, vp]),
    "p2t_infonce_forward": (i32, [vp, vp, vp, i32, i32, i32, f32, f32, i32, vp, vp, vp, vp]),
    "p2t_infonce_backward": (i32, [vp, vp, vp, i32, i32, i32, f32, f32, vp, vp]),
    "p2t_infonce_col_forward": (i32, [vp, vp, i32, i32, f32, vp, i32, i32, f32, i32, vp, vp, vp, vp, vp]),
    "p2t_infonce_col_backward": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, f32, i32, vp, vp]),
    "p2t_clip_adamw_step": (i32, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64),
                                  C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), i32, i32, f64, f64, f64, f64, f64, f64,
                                  vp, vp, vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError as e:          # pragma: no cover
        raise ImportError(f"libp2t_hip.so lacks symbol {_name}; rebuild it") from e
    _fn.restype, _fn.argtypes = _res, _args

for _i, _s in enumerate(_STRUCTS):
    if lib.p2t_struct_size(_i) != C.sizeof(_s):
        raise ImportError(f"ABI mismatch: {_s.__name__} is {C.sizeof(_s)} bytes here, {lib.p2t_struct_size(_i)} in the library")

_NO_RC = {"p2t_gemm_fix_workspace_bytes", "p2t_version", "p2t_is_lab_build", "p2t_last_error", "p2t_struct_size", "p2t_esm2_workspace_bytes", "p2t_llama_workspace_bytes", "p2t_llama_tape_bytes", "p2t_llama_train_workspace_bytes", "p2t_llama_prefill_workspace_bytes", "p2t_llama_decode_workspace_bytes",
          "p2t_adapter_backward_workspace_bytes"}


def call(name: str, *args):
    """Invoke an entry point; raise P2TError on a non-zero return code."""
    rc = getattr(lib, name)(*args)
    if name in _NO_RC:
        return rc
    if rc != 0:
        msg = lib.p2t_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{name}: {msg}")
        raise P2TError(f"{name} failed ({rc}): {msg}")
    return rc


def version() -> int:
    return lib.p2t_version()


def fault_status(clear: bool = False) -> int:
    """The GPU's sticky fault word (bit 0: a split-K consumer timed out).  Synchronises with the device."""
    out = C.c_uint(0)
    call("p2t_fault_status", C.byref(out), int(bool(clear)))
    return int(out.value)
