"""ctypes binding of libactionmesh_amd.so (the C-ABI in include/actionmesh_amd.h).

The HIP library is the product.  There is NO CPU fallback: importing this module
never fails (so host-side logic stays testable without a GPU), but `lib()` raises
`HipLibraryMissing` when the shared object has not been built, and every compute
entry point raises `RuntimeError(am_last_error())` on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# ACTIONMESH_AMD_LIB: another build of the same library (kernel A/B variants, tools/build_variants.sh); never a fallback
LIB_PATH = os.environ.get("ACTIONMESH_AMD_LIB") or os.path.join(_HERE, "libactionmesh_amd.so")
# The float16 build of the same sources (csrc/Makefile: -DAM_F16): same symbols, the 16-bit storage / MFMA element type is IEEE half.
# The reference CLI's `--dtype float16` (inference/video_to_animated_mesh.py:153,222).  Loaded on first use, never a fallback.
LIB_PATH_F16 = os.environ.get("ACTIONMESH_AMD_LIB_F16") or os.path.join(_HERE, "libactionmesh_amd_f16.so")
ABI_VERSION = 2      # 2 (round 5): am_peer_ring owns its events; am_peer_alloc_flags, am_peer_ring_destroy


class HipLibraryMissing(RuntimeError):
    pass


class AmConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32),
        ("width", C.c_int32), ("ff_inner", C.c_int32), ("cross_dim", C.c_int32),
        ("inflated_mask_lo", C.c_uint32), ("inflated_mask_hi", C.c_uint32),
        ("max_batch", C.c_int32), ("max_frames_local", C.c_int32), ("max_tokens", C.c_int32),
        ("max_ctx_tokens", C.c_int32), ("world_size", C.c_int32), ("rank", C.c_int32),
        ("attn_defer_log2", C.c_int32), ("attn_fp8", C.c_int32), ("reserved", C.c_int32 * 6),
    ]


class AmGemmArgs(C.Structure):
    _fields_ = [
        ("A1", C.c_void_p), ("lda1", C.c_int32), ("K1", C.c_int32),
        ("A2", C.c_void_p), ("lda2", C.c_int32),
        ("W", C.c_void_p), ("ldw", C.c_int32),
        ("bias", C.c_void_p), ("residual", C.c_void_p),
        ("C", C.c_void_p), ("ldc", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("act", C.c_int32),
        ("a_G", C.c_int32), ("a_gs", C.c_int32), ("a_off", C.c_int32),
        ("c_G", C.c_int32), ("c_gs", C.c_int32), ("c_off", C.c_int32),
        ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p), ("ln_part", C.c_void_p),
    ]


class AmHeadPostArgs(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("ldx", C.c_int32),
        ("rows", C.c_int64), ("seq_len", C.c_int32), ("rows_per_frame", C.c_int32),
        ("heads", C.c_int32), ("nparts", C.c_int32), ("kinds", C.c_int32 * 3),
        ("w_q", C.c_void_p), ("w_k", C.c_void_p), ("eps", C.c_float),
        ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
        ("out_q", C.c_void_p), ("sq_pad", C.c_int32),
        ("out_k", C.c_void_p), ("out_vt", C.c_void_p), ("sk_pad", C.c_int32),
    ]


class AmNnArgs(C.Structure):
    _fields_ = [
        ("points", C.c_void_p), ("n_points", C.c_int64), ("points_bstride", C.c_int64),
        ("queries", C.c_void_p), ("n_queries", C.c_int64), ("queries_bstride", C.c_int64),
        ("batch", C.c_int32), ("precise", C.c_int32),
        ("out_index", C.c_void_p), ("out_d2", C.c_void_p),
    ]


class AmAttnArgs(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("Vt", C.c_void_p), ("O", C.c_void_p),
        ("nseq", C.c_int32), ("heads", C.c_int32), ("sq", C.c_int32), ("sq_pad", C.c_int32),
        ("sk", C.c_int32), ("sk_pad", C.c_int32), ("nchunks", C.c_int32),
        ("chunk_stride", C.c_int64), ("ldo", C.c_int32), ("scale", C.c_float),
        ("defer_log2", C.c_int32),
        ("chunk_first", C.c_int32), ("chunk_total", C.c_int32), ("rows", C.c_int32), ("state_mode", C.c_int32),
        ("state", C.c_void_p),
    ]


PEER_MAX_RANKS = 16


class AmPeerRing(C.Structure):          # include/actionmesh_amd_sharded.h
    _fields_ = [
        ("world", C.c_int32), ("rank", C.c_int32), ("chunk_bytes", C.c_uint64),
        ("kv", C.c_void_p), ("flags", C.c_void_p),
        ("peer_kv", C.c_void_p * PEER_MAX_RANKS), ("peer_flags", C.c_void_p * PEER_MAX_RANKS),
        ("side_stream", C.c_void_p), ("seq", C.c_uint32),
        ("ev_fork", C.c_void_p), ("ev_pushed", C.c_void_p),
    ]


# every symbol include/*.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "am_last_error": (C.c_char_p, []),
    "am_abi_version": (C.c_int, []),
    "am_create": (C.c_int, [C.POINTER(AmConfig), C.POINTER(_P)]),
    "am_destroy": (C.c_int, [_P]),
    "am_load_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_size_t]),
    "am_weights_missing": (C.c_int, [_P]),
    "am_set_context": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "am_set_branch_hints": (C.c_int, [_P, _P, C.c_int]),
    "am_denoise_forward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "am_denoise_forward_graph": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "am_graph_stats": (C.c_int, [_P, _P]),
    "am_forward_begin": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "am_point_embed": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "am_displacement": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int, _P, _P]),
    "am_attention_fallback_count": (C.c_int, [_P]),
    "am_patchify": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "am_nn_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "am_nn_search": (C.c_int, [C.POINTER(AmNnArgs), _P, C.c_size_t, _P]),
    "am_layer_pre_attn": (C.c_int, [_P, C.c_int, _P]),
    "am_layer_attn_local": (C.c_int, [_P, C.c_int, _P]),
    "am_layer_post_attn": (C.c_int, [_P, C.c_int, _P]),
    "am_forward_end": (C.c_int, [_P, _P, _P]),
    "am_kv_chunk_elems": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "am_bind_kv_buffers": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "am_bind_kv8_buffers": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "am_attention_counters": (C.c_int, [_P, _P]),
    "am_flow_step": (C.c_int, [_P, _P, C.c_int, _P, C.c_float, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    "am_step_flops": (C.c_double, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "am_gemm_bf16": (C.c_int, [C.POINTER(AmGemmArgs), _P]),
    "am_layernorm_bf16": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int, C.c_float, _P]),
    "am_add_layernorm_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_float, _P]),
    "am_row_stats_bf16": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_float, _P]),
    "am_row_stats_finalize": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64, C.c_float, _P]),
    "am_layernorm_stats_bf16": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int, C.c_float, _P, _P]),
    "am_ln_fold_weight": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "am_head_post": (C.c_int, [C.POINTER(AmHeadPostArgs), _P]),
    "am_gemm_headpost_bf16": (C.c_int, [C.POINTER(AmGemmArgs), C.POINTER(AmHeadPostArgs), _P]),
    "am_attention_bf16": (C.c_int, [C.POINTER(AmAttnArgs), _P]),
    "am_attention_quantize_fp8": (C.c_int, [C.POINTER(AmAttnArgs), _P, _P, _P, _P]),
    "am_attention_fp8": (C.c_int, [C.POINTER(AmAttnArgs), _P, _P, _P, _P]),
    "am_peer_alloc": (C.c_int, [C.c_size_t, C.POINTER(_P)]),
    "am_peer_alloc_flags": (C.c_int, [C.c_size_t, C.POINTER(_P), C.POINTER(C.c_int)]),
    "am_peer_free": (C.c_int, [_P]),
    "am_peer_export": (C.c_int, [_P, _P]),
    "am_peer_open": (C.c_int, [_P, C.POINTER(_P)]),
    "am_peer_close": (C.c_int, [_P]),
    "am_peer_copy": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "am_peer_signal": (C.c_int, [_P, C.c_uint32, _P]),
    "am_peer_wait": (C.c_int, [_P, C.c_uint32, _P, _P]),
    "am_debug_trace_begin": (C.c_int, [_P, C.c_int]),
    "am_debug_trace_end": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    "am_debug_trace_stage_name": (C.c_char_p, [C.c_int]),
    "am_f32_to_bf16": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "am_bf16_to_f32": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "am_timestep_sinusoid": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "am_forward_sharded_peer": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.POINTER(AmPeerRing), _P, C.c_int, _P]),
    "am_peer_ring_destroy": (C.c_int, [C.POINTER(AmPeerRing)]),
}

_libs = {}


def lib(kind: str = "bf16") -> C.CDLL:
    """Load (once) and return the shared library of the given 16-bit type ("bf16": the product default; "f16": the float16 build);
    fail loudly if it is absent."""
    if kind in _libs:
        return _libs[kind]
    if kind not in ("bf16", "f16"):
        raise ValueError(f"actionmesh_amd._lib.lib: kind must be 'bf16' or 'f16', got {kind!r}")
    path = LIB_PATH if kind == "bf16" else LIB_PATH_F16
    if not os.path.exists(path):
        raise HipLibraryMissing(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C actionmesh_amd/csrc`.  actionmesh_amd has no CPU fallback."
        )
    l = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(l, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if l.am_abi_version() != ABI_VERSION:
        raise HipLibraryMissing(f"ABI mismatch: library {l.am_abi_version()} != binding {ABI_VERSION}; rebuild")
    _libs[kind] = l
    return l


def kind_of(dtype) -> str:
    """'bf16' / 'f16' for torch.bfloat16 / torch.float16 (or their names); anything else is an error - there is no fp32 library."""
    name = str(dtype).replace("torch.", "")
    if name in ("bfloat16", "bf16"):
        return "bf16"
    if name in ("float16", "half", "f16", "fp16"):
        return "f16"
    raise ValueError(f"actionmesh_amd: the 16-bit type must be bfloat16 or float16, got {dtype!r}")


def autocast_kind(pinned: Optional[str] = None) -> str:
    """'bf16' or 'f16' for a module called like the reference's: the pinned kind if there is one, else the dtype of the caller's
    torch.autocast("cuda", dtype) region - the reference pipeline runs Stage I and Stage II inside one (pipeline.py:671, 679) with the
    CLI's --dtype {bfloat16, float16} - and bfloat16 outside any region.  float16 is chosen only when the caller asked for it."""
    if pinned is not None:
        return pinned
    import torch
    if hasattr(torch, "get_autocast_dtype"):
        enabled, dt = torch.is_autocast_enabled("cuda"), torch.get_autocast_dtype("cuda")
    else:                                    # older torch: the per-device getters (no silent bf16 under autocast(float16))
        enabled, dt = torch.is_autocast_enabled(), torch.get_autocast_gpu_dtype()
    return "f16" if (enabled and dt == torch.float16) else "bf16"


def torch_dtype(kind: str):
    import torch
    return torch.float16 if kind == "f16" else torch.bfloat16


def check(status: int, what: str = "", l: Optional[C.CDLL] = None) -> None:
    if status != 0:
        msg = (l if l is not None else lib()).am_last_error().decode(errors="replace")
        raise RuntimeError(f"libactionmesh_amd {what} failed (status {status}): {msg}")
