"""ctypes binding of ``lib/libleclip_hip.so`` (the C ABI declared in ``include/leclip_hip.h``).

There is no CPU fallback: if the library is missing or a symbol is absent, importing the
product path on a GPU box raises ``HipLibraryError`` - loudly, by design.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

PACKAGE_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PACKAGE_DIR, "lib", "libleclip_hip.so")
ABI_VERSION = 9

F32, F16, BF16 = 0, 1, 2
ACT_NONE, ACT_QUICKGELU = 0, 1
MASK_NONE, MASK_CAUSAL = 0, 1


class HipLibraryError(RuntimeError):
    pass


class HipKernelError(RuntimeError):
    pass


# name -> (restype, argtypes); must list every function include/leclip_hip.h declares
SIGNATURES = {
    "leclip_abi_version": (c_int, []),
    "leclip_strerror": (c_char_p, [c_int]),
    "leclip_last_error": (c_char_p, []),
    "leclip_set_walk_order": (c_int, [c_int]),
    "leclip_set_gemm_family": (c_int, [c_int]),
    "leclip_gemm_kernel_name": (c_char_p, [c_int64, c_int, c_int, c_int]),
    "leclip_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64,
                                     c_float, c_int, c_int, c_void_p]),
    "leclip_gemm_bias_act_res_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                             c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "leclip_gemm_ln_fused_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                         c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "leclip_gemm_ln_partials_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_int,
                                            c_int, c_int, c_void_p]),
    "leclip_ln_stats_finalize_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "leclip_gemm_res_stats_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_int,
                                          c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "leclip_row_stats_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_float, c_int, c_void_p]),
    "leclip_patch_embed_workspace_bytes": (c_int64, [c_int64, c_int, c_int, c_int]),
    "leclip_patch_embed_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_void_p, c_void_p]),
    "leclip_patch_embed_ln_workspace_bytes": (c_int64, [c_int64, c_int, c_int, c_int, c_int]),
    "leclip_patch_embed_ln_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                          c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "leclip_attention_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_int64, c_int,
                                     c_float, c_int, c_void_p]),
    "leclip_attention_prefix_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_int64, c_int,
                                            c_float, c_int, c_int, c_void_p]),
    "leclip_gather_ln_proj_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                          c_int, c_int64, c_float, c_int, c_int, c_void_p]),
    "leclip_l2norm_logits_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "leclip_embed_tokens_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int,
                                        c_void_p]),
    "leclip_prompt_assemble_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                           c_int, c_int, c_int, c_void_p]),
    "leclip_add_pos_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "leclip_window_aggregate_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_float, c_void_p]),
    "leclip_cooccurrence_adjust_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "leclip_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "leclip_quickgelu_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "leclip_quickgelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "leclip_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_int64, c_int, c_float,
                                     c_int, c_void_p]),
    "leclip_eot_index_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "leclip_image_tail_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                      c_int, c_int, c_float, c_float, c_int, c_void_p]),
    "leclip_l2norm_logits_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "leclip_gather_rows_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_int, c_void_p]),
    "leclip_scatter_rows_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int64, c_int64, c_int, c_void_p]),
    "leclip_l2norm_rows_fwd": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p]),
    "leclip_local_pool_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int, c_float, c_float, c_void_p]),
    "leclip_topk_mix_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int64, c_void_p]),
    "leclip_local_pool_masked_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int, c_float, c_float, c_void_p]),
    "leclip_local_pool_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int, c_float, c_float,
                                      c_int64, c_void_p]),
    "leclip_transpose_f32_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p]),
    "leclip_l2norm_rows_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "leclip_bpe_open": (c_void_p, [c_char_p]),
    "leclip_bpe_close": (None, [c_void_p]),
    "leclip_bpe_vocab_size": (c_int64, [c_void_p]),
    "leclip_bpe_encode": (c_int64, [c_void_p, c_char_p, c_void_p, c_int64]),
    "leclip_bpe_tokenize": (c_int, [c_void_p, ctypes.POINTER(c_char_p), c_int64, c_int, c_int, c_void_p]),
    "leclip_crop_resize_fwd": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_int, ctypes.POINTER(c_float),
                                       ctypes.POINTER(c_float), c_int, c_void_p]),
}

_lib = None


def load(path: str = None):
    """Load the shared library once, bind every declared symbol, check the ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("LECLIP_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise HipLibraryError(
            f"{path} not found: build it with `python __graft_entry__.py build` "
            f"(make -C {os.path.join(PACKAGE_DIR, 'csrc')}); there is no CPU fallback for the scoring path")
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise HipLibraryError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.leclip_abi_version()
    if got != ABI_VERSION:
        raise HipLibraryError(f"{path}: ABI version {got}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        lib = load()
        raise HipKernelError(f"{what}: {lib.leclip_strerror(rc).decode()} ({rc}): {lib.leclip_last_error().decode()}")
