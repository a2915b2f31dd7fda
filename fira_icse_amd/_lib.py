"""ctypes binding of libfira_hip.so (the C ABI of include/fira_hip.h).

There is deliberately no fallback: if the HIP library is missing the import fails, and every call
raises ``FiraError`` with the library's own message when it returns non-zero.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# FIRA_HIP_LIB: load another build of the same ABI instead (A/B timing of two builds on one GPU box)
LIB_PATH = os.environ.get("FIRA_HIP_LIB") or os.path.join(HERE, "libfira_hip.so")


class FiraError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [("sou_len", C.c_int32), ("sub_len", C.c_int32), ("ast_len", C.c_int32), ("tar_len", C.c_int32),
                ("d_model", C.c_int32), ("n_head", C.c_int32), ("n_layer", C.c_int32), ("vocab", C.c_int32),
                ("ast_vocab", C.c_int32), ("d_ff", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("B", C.c_int32), ("nnz", C.c_int32), ("sou", C.c_void_p), ("tar", C.c_void_p), ("mark", C.c_void_p),
                ("ast_change", C.c_void_p), ("tar_label", C.c_void_p), ("sub_token", C.c_void_p),
                ("n_nodes", C.c_int32), ("node_rows", C.c_void_p), ("rowptr", C.c_void_p), ("col", C.c_void_p),
                ("val", C.c_void_p), ("n_code", C.c_int32), ("code_rows", C.c_void_p), ("code_mark", C.c_void_p),
                ("n_mem", C.c_int32), ("mem_rows", C.c_void_p), ("mem_dst", C.c_void_p), ("head_rows", C.c_void_p),
                ("n_head_rows", C.c_int32), ("n_emb_items", C.c_int32), ("emb_item_tok", C.c_void_p),
                ("emb_item_ptr", C.c_void_p), ("emb_rows", C.c_void_p), ("n_ast_items", C.c_int32),
                ("ast_rows", C.c_void_p), ("ast_ids", C.c_void_p), ("dec_off", C.c_void_p), ("n_dec_rows", C.c_int32),
                ("dec_off_host", C.c_void_p)]


class TrainOpts(C.Structure):
    _fields_ = [("dropout", C.c_float), ("gcn_dropout", C.c_float), ("seed", C.c_uint64), ("compact_head", C.c_int32),
                ("dtype", C.c_int32), ("compact_dec", C.c_int32), ("zero_grads", C.c_int32)]


class AdamOpts(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("step", C.c_int32),
                ("m", C.c_void_p), ("v", C.c_void_p)]


_P, _I, _F, _L, _Z = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t
_U64, _U32 = C.c_uint64, C.c_uint32
_DP, _BP, _OP = C.POINTER(Dims), C.POINTER(Batch), C.POINTER(TrainOpts)

# name -> (restype, argtypes); every symbol declared in include/fira_hip.h
SIGNATURES = {
    "fira_last_error": (C.c_char_p, []),
    "fira_abi_version": (_I, []),
    "fira_param_count": (_I, [_DP]),
    "fira_param_info": (_I, [_DP, _I, C.c_char_p, C.POINTER(_L), C.POINTER(_L), C.POINTER(C.c_int32), C.POINTER(_L)]),
    "fira_param_total": (_L, [_DP]),
    "fira_workspace_bytes": (_Z, [_DP, _I, _I]),
    "fira_decode_workspace_bytes": (_Z, [_DP, _I, _I]),
    "fira_decode_workspace_bytes_ex": (_Z, [_DP, _I, _I, _I]),
    "fira_gemm_f32": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I]),
    "fira_gemm_bf16": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I]),
    "fira_weight_shadow": (_I, [_P, _I, _I, _P, _P, _P]),
    "fira_gemm_bf16_wb": (_I, [_P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I]),
    "fira_csr_spmm_f32": (_I, [_P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I]),
    "fira_csr_spmm": (_I, [_P, _I, _L, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I]),
    "fira_gcn_layer_fwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _U64, _U32, _I]),
    "fira_gcn_layer_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I]),
    "fira_gcn_weight_planes": (_I, [_P, _I, _P, _P]),
    "fira_linear_x3": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _I, _I]),
    "fira_linear_dgrad_x3": (_I, [_P, _I, _I, _P, _I, _P, _P, _I, _I, _I]),
    "fira_dgrad_x3_splitk_planes_bytes": (_Z, [_I]),
    "fira_dgrad_x3_splitk": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _I, _I]),
    "fira_head_logits_x3_scratch_bytes": (_Z, [_I]),
    "fira_head_logits_x3": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _I, _P]),
    "fira_combination_block_fwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _U64, _U32, _U32, _I]),
    "fira_combination_block_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _F, _U64, _U32, _U32, _I]),
    "fira_combination_block_bwd_part_floats": (_I, []),
    "fira_embed_gather_fwd": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _I]),
    "fira_embed_gather_bwd": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _I]),
    "fira_combination_fwd": (_I, [_P, _I, _P, _P, _P, _P, _F, _U64, _U32]),
    "fira_combination_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _F, _U64, _U32]),
    "fira_add_layernorm_fwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _F, _U64, _U32]),
    "fira_linear_presum_f32": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _P, _F, _U64, _U32]),
    "fira_ln_linear_f32": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "fira_ln_bwd_linear_f32": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _U64, _U32]),
    "fira_linear_layernorm_bf16_fwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _F, _U64, _U32]),
    "fira_add_layernorm_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F, _U64, _U32]),
    "fira_dropout_mask": (_I, [_P, _U64, _U32, _L, _F, _P]),
    "fira_colsum_f32": (_I, [_P, _I, _I, _P, _I, _P]),
    "fira_attention_fwd": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I]),
    "fira_attention_bwd": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I,
                                _P, _I, _P, _I, _P, _I]),
    "fira_attention_fwd_ex": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _P]),
    "fira_attention_bwd_ex": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I,
                                   _P, _I, _P, _I, _P, _I, _P, _I, _I, _P]),
    "fira_decode_attention": (_I, [_P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "fira_copy_score_fwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "fira_copy_score_bwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fira_head_loss": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I]),
    "fira_adam_step": (_I, [_P, _L, _P, _P, _P, _P, _F, _F, _F, _F, _I, _P]),
    "fira_adam_step_mb": (_I, [_P, _L, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I, _P, _P]),
    "fira_inv_count": (_I, [_P, _P, _P]),
    "fira_adam_step_count": (_I, [_P, _L, _P, _P, _P, _P, _F, _F, _F, _F, _I, _P]),
    "fira_pack_stats": (_I, [_P, _P, _P, _P]),
    "fira_debug_chain": (_I, [_P, _I, _I, _P]),
    "fira_train_fwd_bwd": (_I, [_P, _DP, _BP, _P, _P, _P, _Z, _OP, _P, _P, _P]),
    "fira_train_step": (_I, [_P, _DP, _BP, _P, _P, _P, _Z, _OP, _P, _P, C.POINTER(AdamOpts)]),
    "fira_gemm_wgrad_panel": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _Z]),
    "fira_ffn_fwd": (_I, [_P, _I, _I] + [_P] * 11 + [_F, _U64, _U32, _I]),
    "fira_ffn_bwd": (_I, [_P, _I, _I] + [_P] * 17 + [_F, _U64, _U32, _I]),
    "fira_head_topk": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _I]),
    "fira_train_step_rows": (_I, [_P, _DP, _BP, _P, _P, _P, _Z, _OP, _P, _P, C.POINTER(AdamOpts), _P]),
    "fira_adam_rows_sync": (_I, [_P, _DP, _P, C.POINTER(AdamOpts), _P]),
    "fira_adam_rows_catchup": (_I, [_P, _DP, _P, C.POINTER(AdamOpts), _P, _I, _P, _I]),
    "fira_adam_rows_step": (_I, [_P, _DP, _P, _P, C.POINTER(AdamOpts), _P, _P, _P, _I]),
    "fira_train_step_begin_rows": (_I, [_P, _DP, _BP, _P, _P, _P, _Z, _OP, _P, _P, _P, C.POINTER(AdamOpts), _P]),
    "fira_train_step_end_rows": (_I, [_P, _P, C.POINTER(AdamOpts), _P, _P, _P]),
    "fira_train_step_begin": (_I, [_P, _DP, _BP, _P, _P, _P, _Z, _OP, _P, _P, _P]),
    "fira_train_step_end": (_I, [_P, _P, C.POINTER(AdamOpts), _P, _P]),
    "fira_f32_to_bf16": (_I, [_P, _L, _P, _P]),
    "fira_bf16_to_f32": (_I, [_P, _L, _P, _P]),
    "fira_prof_enable": (None, [_I]),
    "fira_prof_report": (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_L)]),
    "fira_param_groups": (_I, [_DP, C.POINTER(_L), C.POINTER(_L)]),
    "fira_host_node_lists": (_I, [_I, _I, _I, _I, _I] + [_P] * 7 + [_I] + [_P] * 14),
    "fira_host_collate_csr": (_I, [_I, _I, _P, _L, _P, _P, _P, _P, _P, _P, _P]),
    "fira_forward_dev": (_I, [_P, _DP, _BP, _P, _P, _Z, _P, _P, _P, _I]),
    "fira_decode_begin": (_I, [_P, _DP, _BP, _P, _P, _Z, _I]),
    "fira_decode_begin_ex": (_I, [_P, _DP, _BP, _P, _P, _Z, _I, _I]),
    "fira_decode_step_ex": (_I, [_P, _DP, _P, _P, _Z, _I, _I, _I, _P, _P, _P, _P, _P, _I]),
    "fira_beam_prepare": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "fira_beam_select": (_I, [_P, _DP, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fira_greedy_advance": (_I, [_P, _DP, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fira_decode_step": (_I, [_P, _DP, _P, _P, _Z, _I, _I, _I, _P, _P, _P, _P, _P]),
    "fira_decoder_forward": (_I, [_P, _DP, _P, _P, _Z, _I, _P, _P, _P, _P]),
    "fira_decode_memory": (_P, [_DP, _P, _I, _I]),
    "fira_decode_mem_valid": (_P, [_DP, _P, _I, _I]),
}


def _ensure_built():
    """Build the library on first use if it is missing (same command as `python -m fira_icse_amd.build`).
    This is a build step, not a compute fallback: without hipcc the import fails."""
    if os.path.exists(LIB_PATH):
        return
    import fcntl
    from . import build as _build
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:      # ranks of one node import concurrently
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.exists(LIB_PATH):
                _build.build()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def load():
    try:
        _ensure_built()
    except Exception as e:
        raise ImportError(
            "fira_icse_amd: %s is missing and could not be built (%s). Build it with `python -m fira_icse_amd.build` "
            "(needs hipcc, gfx950). There is no CPU fallback." % (LIB_PATH, e))
    lib = C.CDLL(LIB_PATH)
    foreign = bool(os.environ.get("FIRA_HIP_LIB"))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)      # AttributeError if the library does not export a declared symbol
        except AttributeError:
            if foreign:                  # an older build loaded for A/B timing: its missing op-level entries are never called
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    # FIRA_HIP_LIB (A/B timing of an older build through the model-level entry points, whose signatures did not change
    # between v5 and v8 -- v8 appended a field to fira_batch, which older builds never read; callers ask has_symbol() before
    # using an entry an older build lacks) may load an older library; the tree's own library must be v8
    ok = (10,) if not os.environ.get("FIRA_HIP_LIB") else (5, 6, 7, 8, 9, 10)
    if lib.fira_abi_version() not in ok:
        raise ImportError("libfira_hip.so ABI version mismatch")
    return lib


_lib = None


def has_symbol(name: str) -> bool:
    """True if the loaded library exports ``name`` (an older build loaded through FIRA_HIP_LIB may not)."""
    try:
        getattr(lib(), name)
        return True
    except AttributeError:
        return False


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


_host_warned = False
_host_failed = False


def host_lib():
    """The library for its two HOST-ONLY helpers (fira_host_collate_csr / fira_host_node_lists), or None when it cannot be
    loaded or built (a CPU-only box without hipcc): the callers then run the numpy statements those helpers are tested
    against (tests/test_host_lists.py) -- same arrays, slower.  Device code never goes through this accessor."""
    global _host_warned, _host_failed
    if _host_failed:                     # one failed load / build attempt is remembered: no hipcc run per batch afterwards
        return None
    try:
        return lib()
    except (ImportError, OSError, AttributeError) as e:      # (AttributeError: a stale library without a declared symbol)
        _host_failed = True
        if not _host_warned:
            _host_warned = True
            import warnings
            warnings.warn("fira_icse_amd: libfira_hip.so unavailable (%s); host collate falls back to numpy" % (e,))
        return None


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().fira_last_error()
        raise FiraError("%s failed: %s" % (what or "libfira_hip call", msg.decode() if msg else "unknown error"))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def cur_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_dims(cfg) -> Dims:
    return Dims(cfg.sou_len, cfg.sub_token_len, cfg.ast_change_len, cfg.tar_len, cfg.embedding_dim, cfg.num_head,
                cfg.num_layers, cfg.vocab_size, cfg.ast_change_vocab_size, 4 * cfg.embedding_dim)
