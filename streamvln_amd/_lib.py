"""ctypes binding of include/streamvln_hip.h (libstreamvln_hip.so, built in-tree by csrc/build.sh).

There is no CPU fallback: a missing library or a failing call raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVLN_LIB: another build of the same engine (tools/ab_lib.sh: A/B of two builds on one GPU box); default = the in-tree library
LIB_PATH = os.environ.get("SVLN_LIB") or os.path.join(_HERE, "libstreamvln_hip.so")

SVLN_BF16, SVLN_F32 = 0, 1
EPI_NONE, EPI_GELU_TANH, EPI_GELU_ERF, EPI_SWIGLU, EPI_ARGMAX = 0, 1, 2, 3, 4


class SvlnConfig(C.Structure):
    _fields_ = [
        ("v_hidden", C.c_int32), ("v_inter", C.c_int32), ("v_heads", C.c_int32), ("v_layers", C.c_int32),
        ("v_patch", C.c_int32), ("v_image", C.c_int32), ("v_eps", C.c_float),
        ("hidden", C.c_int32), ("layers", C.c_int32), ("q_heads", C.c_int32), ("kv_heads", C.c_int32),
        ("head_dim", C.c_int32), ("inter", C.c_int32), ("vocab", C.c_int32),
        ("rope_theta", C.c_float), ("rms_eps", C.c_float),
        ("max_positions", C.c_int32), ("max_envs", C.c_int32), ("max_frames", C.c_int32), ("dtype", C.c_int32),
    ]


class SvlnTurnArgs(C.Structure):
    """svln_turn_args of include/streamvln_hip.h"""
    _fields_ = [
        ("pixels", C.c_void_p), ("n_frames", C.c_int32), ("pixels_on_device", C.c_int32),
        ("env", C.c_int32),
        ("ids", C.c_void_p), ("n_ids", C.c_int32), ("n_memory", C.c_int32),
        ("new_window", C.c_int32), ("new_episode", C.c_int32),
        ("max_new_tokens", C.c_int32),
        ("eos_ids", C.c_void_p), ("n_eos", C.c_int32),
    ]


_P, _I, _F, _I64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
_PI32, _PI64, _PF, _PD = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_double)

#: every symbol include/streamvln_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "svln_create": (_I, [C.POINTER(SvlnConfig), _I, C.POINTER(_P)]),
    "svln_destroy": (None, [_P]),
    "svln_last_error": (C.c_char_p, []),
    "svln_sync": (_I, [_P]),
    "svln_synth_tensor": (_I, [_P, C.c_char_p, C.c_uint64, _F, _F]),
    "svln_set_tensor": (_I, [_P, C.c_char_p, _P, _I, _I64, _I]),
    "svln_weights_ready": (_I, [_P]),
    "svln_get_tensor_f32": (_I, [_P, C.c_char_p, _PF, _I64]),
    "svln_reset_env": (_I, [_P, _I]),
    "svln_kv_reset": (_I, [_P, _I]),
    "svln_env_state": (_I, [_P, _I, _PI32, _PI32]),
    "svln_encode_frames": (_I, [_P, _P, _I, _I]),
    "svln_preprocess_frames": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "svln_preprocess_frames_enqueue": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "svln_engine_stream": (_I, [_P, C.POINTER(C.c_void_p)]),
    "svln_preprocess_time": (_I, [_P, _PD, _PI64, _I]),
    "svln_frame_ring": (_I, [_P, _I, _I, _I, C.POINTER(C.c_void_p), _PI64]),
    "svln_frame_ring_wait": (_I, [_P, _I]),
    "svln_turn": (_I, [_P, C.POINTER(SvlnTurnArgs), _PI64, _I, _PI32, _PI32]),
    "svln_append_turn": (_I, [_P, _I, _PI64, _I, _I]),
    "svln_append_turn_at": (_I, [_P, _I, _PI64, _I, _I, _I]),
    "svln_generate_batch": (_I, [_P, _PI32, _I, _I, _PI64, _I, _PI64, _I, _PI32]),
    "svln_batch_submit": (_I, [_P, _I, _I, _PI64, _I, _PI32]),
    "svln_batch_step": (_I, [_P, _PI32, _PI32, _PI32]),
    "svln_batch_result": (_I, [_P, _I, _PI32, _PI64, _I, _PI32]),
    "svln_batch_cancel": (_I, [_P, _I]),
    "svln_set_turn_row_limit": (_I, [_P, _I]),
    "svln_set_repetition_penalty": (_I, [_P, _F]),
    "svln_get_hidden_batch": (_I, [_P, _I, _PF, _I, _PI32]),
    "svln_generate": (_I, [_P, _I, _I, _PI64, _I, _PI64, _I, _PI32]),
    "svln_generate_fixed": (_I, [_P, _I, _I, _PI64]),
    "svln_get_hidden": (_I, [_P, _PF, _I, _PI32]),
    "svln_get_embeds": (_I, [_P, _I, _I, _I, _PF]),
    "svln_get_frame_feats": (_I, [_P, _I, _I, _PF]),
    "svln_get_top2": (_I, [_P, _PF]),
    "svln_set_layer_taps": (_I, [_P, _I, _I]),
    "svln_get_layer_taps": (_I, [_P, _PF]),
    "svln_get_layer_probe": (_I, [_P, _I, _PF, _I64, _PI32, _PI32]),
    "svln_set_decode_graph": (_I, [_P, _I]),
    "svln_set_decode_persistent": (_I, [_P, _I]),
    "svln_probe_decode_layer": (_I, [_P, _I, C.POINTER(C.c_uint64), _I, _PI32]),
    "svln_set_fp8_decode": (_I, [_P, _I]),
    "svln_set_fp8_gemm": (_I, [_P, _I]),
    "svln_set_memory_prune": (_I, [_P, _I]),
    "svln_op_memory_prune": (_I, [_P, _P, _I, _I, _PI32, _PF]),
    "svln_probe_reset": (_I, [_P]),
    "svln_probe_read": (_I, [_P, _PD, _PI64, _PD]),
    "svln_probe_read_prefill": (_I, [_P, _PD, _PI64, _PD, _PD, _PD]),
    "svln_phase_times": (_I, [_P, _PD, _PD, _PD, _I]),
    "svln_set_feature_cache": (_I, [_P, _I]),
    "svln_feature_cache_stats": (_I, [_P, _PI64, _PI64]),
    "svln_op_gemm": (_I, [_P, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I]),
    "svln_op_gemm_norm": (_I, [_P, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _F, _I, _I, _I, _I, _PI32]),
    "svln_op_gemm_norm_q8": (_I, [_P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _F, _I, _I, _I, _I, _P, _P, _PI32]),
    "svln_op_gemv": (_I, [_P, _P, _I, _P, _P, _F, _P, _P, _P, _I, _I, _I, _PI32]),
    "svln_op_gemm_fp8": (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I]),
    "svln_op_gemv_batched": (_I, [_P, _P, _I, _P, _I, _P, _F, _P, _P, _I, _P, _I, _I, _I, _I, _I, _PI32]),
    "svln_op_quant_fp8": (_I, [_P, _P, _I64, _I, _P, _P]),
    "svln_op_gemv_fp8": (_I, [_P, _P, _P, _I, _P, _P, _F, _P, _P, _P, _I, _I, _I, _PI32]),
    "svln_op_rmsnorm": (_I, [_P, _P, _P, _P, _I, _I, _F]),
    "svln_op_layernorm": (_I, [_P, _P, _P, _P, _P, _I, _I, _F]),
    "svln_op_attention_llm": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _I, _I]),
    "svln_op_attention_vit": (_I, [_P, _P, _I, _I, _P, _I]),
    "svln_op_pool": (_I, [_P, _P, _P, _I]),
    "svln_op_patchify": (_I, [_P, _P, _P, _I]),
}

_lib = None


def load():
    """dlopen the engine (no GPU needed to load; compute calls need one)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with streamvln_amd/csrc/build.sh "
                               f"(or __graft_entry__.build()); there is no fallback path")
        # torch first: the engine must share torch's HIP runtime (same libamdhip64 instance) so that device
        # pointers and the GPU context are common to both; loading ours first can bind a second runtime copy.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class SvlnError(RuntimeError):
    pass


def check(rc: int):
    if rc != 0:
        raise SvlnError(load().svln_last_error().decode("utf-8", "replace"))
