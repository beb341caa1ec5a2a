"""ctypes binding of include/captioner_hip.h (libcaptioner_hip.so).

There is no fallback: if the library cannot be found, built or loaded this module raises - the product path never
routes through PyTorch ops or the test oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcaptioner_hip.so")

CAP_F32, CAP_BF16, CAP_F32_SPLIT = 0, 1, 2
CAP_PIX_F32_NCHW, CAP_PIX_U8_NHWC = 0, 1


class CapConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("arch", C.c_int32), ("compute_dtype", C.c_int32),
        ("image_size", C.c_int32), ("patch_size", C.c_int32), ("v_hidden", C.c_int32), ("v_layers", C.c_int32),
        ("v_heads", C.c_int32), ("v_mlp", C.c_int32), ("v_eps", C.c_float),
        ("t_hidden", C.c_int32), ("t_layers", C.c_int32), ("t_heads", C.c_int32), ("t_ffn", C.c_int32),
        ("vocab", C.c_int32), ("max_pos", C.c_int32), ("t_eps", C.c_float),
        ("bos", C.c_int32), ("eos", C.c_int32), ("pad", C.c_int32),
        ("max_batch", C.c_int32), ("max_beams", C.c_int32), ("max_len", C.c_int32),
        ("pix_mean", C.c_float * 3), ("pix_std", C.c_float * 3),
        ("embed_dim", C.c_int32), ("pool_queries", C.c_int32), ("pool_heads", C.c_int32), ("mm_layers", C.c_int32),
        ("min_len", C.c_int32),
        ("q_hidden", C.c_int32), ("q_layers", C.c_int32), ("q_heads", C.c_int32), ("q_ffn", C.c_int32),
        ("q_cross_freq", C.c_int32), ("num_query_tokens", C.c_int32), ("q_eps", C.c_float),
        ("cross_kv_fp32", C.c_int32), ("weight_int8", C.c_int32),
    ]


_SIGNATURES = {
    "cap_last_error": (C.c_char_p, []),
    "cap_version": (C.c_int, []),
    "cap_create": (C.c_int, [C.POINTER(CapConfig), C.POINTER(C.c_void_p)]),
    "cap_create_shared": (C.c_int, [C.POINTER(CapConfig), C.c_void_p, C.POINTER(C.c_void_p)]),
    "cap_destroy": (C.c_int, [C.c_void_p]),
    "cap_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p]),
    "cap_finalize_weights": (C.c_int, [C.c_void_p]),
    "cap_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cap_generate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cap_embed_text": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cap_crop_resize_tables": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "cap_crop_resize_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cap_crop_resize_u8_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cap_generate_groups": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "cap_set_early_exit": (C.c_int, [C.c_void_p, C.c_int]),
    "cap_last_decode_steps": (C.c_int, [C.c_void_p]),
    "cap_set_decode_path": (C.c_int, [C.c_void_p, C.c_int]),
    "cap_last_decode_path": (C.c_int, [C.c_void_p]),
    "cap_set_row_compaction": (C.c_int, [C.c_void_p, C.c_int]),
    "cap_last_row_compaction": (C.c_int, [C.c_void_p]),
    "cap_cross_cache_kind": (C.c_int, [C.c_void_p]),
    "cap_device_bytes": (C.c_size_t, [C.c_void_p]),
    "cap_g8_saturations": (C.c_longlong, [C.c_int]),
    "cap_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "cap_profile_report": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "cap_op_gemm": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cap_op_pack_kv16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cap_op_gemm_crosskv": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cap_op_gemm_partial": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cap_op_layernorm": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_int, C.c_void_p]),
    "cap_op_vit_attention": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cap_op_reduce_layernorm": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cap_op_gemm_skinny": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p]),
    "cap_op_gemm_skinny_slices": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "cap_op_quant_i8_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "cap_op_gemm_skinny_i8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p]),
    "cap_op_gemm_skinny_i8_slices": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "cap_op_vit_attention_hd": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p]),
    "cap_op_decode_attention": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cap_op_beam_candidates": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "cap_op_convert": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cap_op_convert_weight": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)

_lib = None
_lock = threading.Lock()


class CaptionerHipError(RuntimeError):
    pass


def load_library(build_if_missing: bool = True) -> C.CDLL:
    """Load (building first if the .so is absent and hipcc exists).  Raises CaptionerHipError otherwise."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            if not build_if_missing:
                raise CaptionerHipError(f"{LIB_PATH} is missing - run `python -m embodied_captioning_amd.build`")
            from . import build as _build
            try:
                _build.build(verbose=False)
            except Exception as e:  # noqa: BLE001
                raise CaptionerHipError(f"HIP extension missing and could not be built: {e}") from e
        # torch (when present) must load its bundled HIP runtime first so both share one libamdhip64.so.7
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001
            pass
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise CaptionerHipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise CaptionerHipError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def last_error() -> str:
    return load_library().cap_last_error().decode(errors="replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise CaptionerHipError(f"{what} failed (rc={rc}): {last_error()}")
