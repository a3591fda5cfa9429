"""ctypes binding of libesmk.so (the C-ABI declared in include/esmk.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libesmk.so")
# experiments build (make EXPERIMENTS=1): alternative GEMM kernels + ESMK_GEMM_* switches; tools/ and tests only
XLIB_PATH = os.path.join(_HERE, "libesmk_x.so")

F32, BF16 = 0, 1
PACK_PLAIN, PACK_SWIGLU = 0, 1
EPI_BF16, EPI_F32, EPI_SWIGLU, EPI_RESIDUAL, EPI_BIAS_GELU = 0, 1, 2, 3, 4
ATTN_QBLK = 128

vp = C.c_void_p


class EsmkEsm3Encoder(C.Structure):
    _fields_ = [(n, vp) for n in ("sequence_embed", "plddt_w", "plddt_b", "res_plddt_w", "res_plddt_b",
                                  "structure_embed", "ss8_embed", "sasa_embed", "function_embed", "residue_embed")]


class EsmkLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1_g", "ln1_b", "w_qkv", "qln_g", "kln_g", "w_o", "ln2_g", "ln2_b", "w_ffn1",
                                  "w_ffn2", "geom_ln_g", "w_geom_proj", "w_geom_out", "geom_dist_scale",
                                  "geom_rot_scale", "w_qkv_f", "qkv_u", "qkv_v", "w_ffn1_f", "ffn1_u", "ffn1_v",
                                  "w_geom_proj_f", "geom_u")]


class EsmkTrunk(C.Structure):
    _fields_ = [("d_model", C.c_int), ("n_heads", C.c_int), ("n_layers", C.c_int), ("ffn_hidden", C.c_int),
                ("v_heads", C.c_int), ("residual_scale", C.c_float), ("layers", C.POINTER(EsmkLayer)),
                ("final_norm_g", vp), ("rope_cos", vp), ("rope_sin", vp), ("rope_len", C.c_int),
                ("probe_start", C.POINTER(vp)), ("probe_stop", C.POINTER(vp)), ("small_tokens", C.c_int)]


class EsmkBatch(C.Structure):
    _fields_ = [("n_tokens", C.c_int), ("n_segments", C.c_int), ("cu_seqlens", vp), ("positions", vp),
                ("qblk_map", vp), ("n_qblk", C.c_int), ("affine", vp), ("affine_mask", vp), ("seq_id", vp),
                ("chain_id", vp), ("row_cu", vp), ("n_rows", C.c_int), ("max_row_len", C.c_int), ("pad_tail", C.c_int),
                ("max_position", C.c_int)]


class EsmkEsm3Inputs(C.Structure):
    _fields_ = [(n, vp) for n in ("sequence", "structure", "ss8", "sasa", "function_tokens", "residue_tokens",
                                  "average_plddt", "per_res_plddt")] + [("rows", C.c_int * 8)]


class EsmkHead(C.Structure):
    _fields_ = [("w0", vp), ("b0", vp), ("ln_g", vp), ("ln_b", vp), ("w3", vp), ("b3", vp), ("out_dim", C.c_int)]


# name -> (restype, argtypes); this table is also what tests/test_cabi_symbols.py checks against esmk.h
SIGNATURES = {
    "esmk_version": (C.c_int, []),
    "esmk_last_error": (C.c_char_p, []),
    "esmk_build_id": (C.c_char_p, []),
    "esmk_packed_weight_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "esmk_pack_weight": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "esmk_layernorm": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "esmk_add_layernorm": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "esmk_add2_layernorm": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "esmk_gemm": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_float, vp, vp]),
    "esmk_gemm_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "esmk_gemm_ws": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_float, vp, vp, C.c_size_t, vp]),
    "esmk_gemm_noalloc": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_float, vp, vp]),
    "esmk_qk_prep": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp]),
    "esmk_qk_prep_ex": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp]),
    "esmk_qkv_stats_parts": (C.c_int, [C.c_int, C.c_int]),
    "esmk_gemm_qkv_stats": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp]),
    "esmk_k_prep": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, C.c_int, vp, vp]),
    "esmk_attn_varlen_qfused": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp,
                                          vp, vp, vp, C.c_int, C.c_float, vp]),
    "esmk_attn_plan": (C.c_int, [vp, C.c_int, vp, C.c_int]),
    "esmk_attn_varlen": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "esmk_attn_l64_d128": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "esmk_embed": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "esmk_esm3_encode": (C.c_int, [C.POINTER(EsmkEsm3Encoder), vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "esmk_frames": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "esmk_geom_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "esmk_geom_attn": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "esmk_geom_attn_ex": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "esmk_knn_neighbourhoods": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "esmk_vq_quantize": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "esmk_trunk_workspace_bytes": (C.c_size_t, [C.POINTER(EsmkTrunk), C.c_int]),
    "esmk_trunk_forward": (C.c_int, [C.POINTER(EsmkTrunk), C.POINTER(EsmkBatch), vp, vp, vp, vp, C.c_size_t, vp]),
    "esmk_trunk_forward_ex": (C.c_int, [C.POINTER(EsmkTrunk), C.POINTER(EsmkBatch), vp, vp, vp, C.c_int, vp, C.c_size_t, vp]),
    "esmk_segments_from_tokens": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int, vp, vp, vp]),
    "esmk_res_stats_parts": (C.c_int, [C.c_int]),
    "esmk_row_stats": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp]),
    "esmk_fold_layernorm": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
    "esmk_gemm_lnfold": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_float, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "esmk_gemm_residual_stats": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, C.c_float, vp, vp, vp, vp, vp]),
    "esmk_esm3_encode_ex": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "esmk_sampling_mask": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int64, C.c_int, vp, vp]),
    "esmk_sample_sasa": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "esmk_compact_tokens": (C.c_int, [vp, C.c_int, C.c_int, C.c_int64, C.c_int, vp, vp, vp, vp, vp, vp]),
    "esmk_unpack_rows": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "esmk_segment_mean": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp]),
    "esmk_sample_track": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp, vp, vp]),
    "esmk_dim6_frames": (C.c_int, [vp, C.c_int, C.c_float, vp, vp, C.c_int, vp]),
    "esmk_categorical_mean": (C.c_int, [vp, C.c_int, C.c_int, C.c_float, C.c_float, vp, vp]),
    "esmk_pairwise_pae": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp]),
    "esmk_head_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "esmk_head_forward": (C.c_int, [C.POINTER(EsmkHead), vp, vp, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "esmk_head_ex_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "esmk_head_forward_ex": (C.c_int, [C.POINTER(EsmkHead), vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "esmk_event_create": (C.c_int, [C.POINTER(vp)]),
    "esmk_event_destroy": (C.c_int, [vp]),
    "esmk_event_record": (C.c_int, [vp, vp]),
    "esmk_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
}


class EsmkError(RuntimeError):
    pass


_lib = None
_xlib = None


def _load(path):
    if not os.path.exists(path):
        raise EsmkError(f"{path} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or `make -C esm_amd/csrc`). There is no non-HIP fallback.")
    # torch ships its own libamdhip64 (same soname): load it FIRST so that libesmk.so binds to the one HIP
    # runtime torch's allocator and streams live in; the other order leaves two runtimes in the process.
    import torch  # noqa: F401
    l = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args
    return l


def lib():
    """Load libesmk.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def lib_experiments():
    """libesmk_x.so: same C-ABI, built with -DESMK_EXPERIMENTS (alternative GEMM kernels selected by ESMK_GEMM_*
    environment variables).  For tools/ and tests; the product package never calls this."""
    global _xlib
    if _xlib is None:
        _xlib = _load(XLIB_PATH)
    return _xlib


def check(rc: int, what: str = ""):
    if rc < 0:
        msg = lib().esmk_last_error().decode(errors="replace")
        raise EsmkError(f"{what or 'esmk'} failed (code {rc}): {msg}")
    return rc


def ptr(t):
    """Device (or host) pointer of a torch tensor / None."""
    return None if t is None else t.data_ptr()
