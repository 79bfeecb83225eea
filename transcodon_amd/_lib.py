"""ctypes binding of libtranscodon_hip.so (declarations: include/transcodon_hip.h).

The library is built in-tree (``make -C transcodon_amd/csrc``).  There is no CPU
fallback: if the shared object is missing or a call fails, an exception is raised.
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TC_LIB_PATH points at another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("TC_LIB_PATH") or os.path.join(_HERE, "csrc", "libtranscodon_hip.so")

ABI_VERSION = 131          # TC_VERSION of include/transcodon_hip.h this binding restates
TC_F32, TC_BF16 = 0, 1
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL, EPI_QKV_ROPE = 0, 1, 2, 3
EPI_PLAIN_F32, EPI_BIAS_GELU_SAVE, EPI_DGELU = 4, 5, 6
PROF_GEMM, PROF_ATTN, PROF_LAYERNORM, PROF_EMBED, PROF_HEAD, PROF_DECODE, PROF_LN_ROWS = range(7)

vp = C.c_void_p


class LayerWeights(C.Structure):
    _fields_ = [(n, vp) for n in (
        "ln1_g", "ln1_b", "w_qkv", "b_qkv", "w_o", "b_o", "ln2_g", "ln2_b",
        "w_fc1", "b_fc1", "w_fc2", "b_fc2", "rope_cos", "rope_sin",
        "w_qkv_ln", "c_qkv_ln", "s_qkv_ln", "w_fc1_ln", "c_fc1_ln", "s_fc1_ln")]


class EncoderDesc(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("dtype", "num_layers", "E", "F", "H", "vocab", "n_species",
                                  "n_struct", "pad_idx", "rope_table_len")]
        + [("ln_eps", C.c_float)]
        + [(n, vp) for n in ("tok_table", "species_table")]
        + [("layers", C.POINTER(LayerWeights))]
        + [(n, vp) for n in ("lnf_g", "lnf_b", "w_heads", "b_heads", "lm_ln_g", "lm_ln_b",
                             "lm_bias", "st_ln_g", "st_ln_b", "st_w", "st_bias",
                             "w_heads_ln", "c_heads_ln", "s_heads_ln")]
    )


LAYER_GRAD_FIELDS = ("ln1_g", "ln1_b", "w_qkv", "b_qkv", "w_o", "b_o", "ln2_g", "ln2_b",
                     "w_fc1", "b_fc1", "w_fc2", "b_fc2")
HEAD_GRAD_FIELDS = ("lnf_g", "lnf_b", "w_heads", "b_heads", "lm_ln_g", "lm_ln_b", "lm_bias",
                    "st_ln_g", "st_ln_b", "st_w", "st_bias")


class LayerGrads(C.Structure):
    _fields_ = [(n, vp) for n in LAYER_GRAD_FIELDS]


class EncoderGrads(C.Structure):
    _fields_ = ([(n, vp) for n in ("tok_table", "species_table")]
                + [("layers", C.POINTER(LayerGrads))]
                + [(n, vp) for n in HEAD_GRAD_FIELDS])


GRADS_READY_FN = C.CFUNCTYPE(None, C.c_int, C.c_void_p)


class HipLibraryError(RuntimeError):
    pass


_SIGNATURES = {
    "tc_version": (C.c_int, []),
    "tc_struct_bytes": (C.c_size_t, [C.c_int]),
    "tc_init_device": (C.c_int, []),
    "tc_last_error": (C.c_char_p, []),
    "tc_embed_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp] + [C.c_int] * 6 + [vp]),
    "tc_embed_fwd_ln": (C.c_int, [vp, vp, vp, vp, vp, vp, vp] + [C.c_int] * 6 + [vp, vp, vp, C.c_float, vp]),
    "tc_gemm_residual_split": (C.c_int, [C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tc_layernorm_fwd_split": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    "tc_split_to_f32": (C.c_int, [vp, vp, vp, C.c_long, vp]),
    "tc_ln_rows_finalize": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    "tc_ln_rows_clear": (C.c_int, [vp, C.c_int, vp]),
    "tc_copy_f32": (C.c_int, [vp, vp, C.c_long, vp]),
    "tc_splitk_bias_residual": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp]),
    "tc_ln_fold_weights": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "tc_gemm_ln": (C.c_int, [C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_float, vp, vp, C.c_int, C.c_int, vp]),
    "tc_gemm_residual_ln": (C.c_int, [C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                      C.c_int, vp]),
    "tc_ln_fold_supported": (C.c_int, [C.c_int] * 5),
    "tc_ln_fold_rows_supported": (C.c_int, [C.c_int, C.c_long, C.c_int, C.c_int]),
    "tc_layernorm_fwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    "tc_gemm": (C.c_int, [C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int,
                          C.c_int, C.c_int, C.c_float, vp, vp, C.c_int, C.c_int, vp]),
    "tc_key_mask_words": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "tc_attn_fwd": (C.c_int, [C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tc_attn_fwd_q2": (C.c_int, [C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tc_attn_scores": (C.c_int, [C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tc_head_logits_fwd": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp,
                                     C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "tc_codon_decode": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "tc_codon_stats": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, vp]),
    "tc_codon_minmax": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_long, vp]),
    "tc_ce_loss_fwd_bwd": (C.c_int, [vp, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp]),
    "tc_mean_pool": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "tc_gemm_train": (C.c_int, [C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, vp]),
    "tc_transpose_batch": (C.c_int, [C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "tc_transpose_cast": (C.c_int, [C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp,
                                    C.c_int, C.c_int, vp]),
    "tc_gemm_tn": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tc_colsum": (C.c_int, [C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "tc_reduce_slabs": (C.c_int, [vp, vp, C.c_int, C.c_long, C.c_int, vp]),
    "tc_layernorm_bwd": (C.c_int, [C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int,
                                   C.c_float, vp]),
    "tc_layernorm_bwd_dy": (C.c_int, [C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp,
                                      C.c_int, C.c_int, C.c_float, vp]),
    "tc_dgelu": (C.c_int, [C.c_int, vp, vp, vp, C.c_long, vp]),
    "tc_head_logits_bwd": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp,
                                     C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "tc_embed_bwd": (C.c_int, [vp, vp, vp, vp, vp] + [C.c_int] * 6 + [vp]),
    "tc_rope_bwd": (C.c_int, [C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_float, vp]),
    "tc_attn_fwd_train": (C.c_int, [C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tc_attn_bwd": (C.c_int, [C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_float,
                              C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tc_attn_bwd_static": (C.c_int, [C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_float,
                                     C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tc_gemm_qkv_packed": (C.c_int, [C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                     vp, vp, vp, C.c_int, C.c_int, vp]),
    "tc_attn_packed_block_rows": (C.c_int, [C.c_int]),
    "tc_attn_fwd_packed": (C.c_int, [C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_double, vp]),
    "tc_attn_fwd_packed_q2": (C.c_int, [C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_double, vp]),
    "tc_mean_pool_packed": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "tc_encoder_forward_packed": (C.c_int, [C.POINTER(EncoderDesc), vp, vp, vp, vp, C.c_int, vp, C.c_int, vp, C.c_int,
                                            vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "tc_gemm_ln_qkv_panels": (C.c_int, [C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_float, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp]),
    "tc_workspace_bytes": (C.c_size_t, [C.POINTER(EncoderDesc), C.c_int, C.c_int]),
    "tc_encoder_forward": (C.c_int, [C.POINTER(EncoderDesc), vp, vp, C.c_int, C.c_int, vp, vp, vp,
                                     C.c_int, C.POINTER(C.c_int32), C.POINTER(vp), vp, vp,
                                     C.c_size_t, vp]),
    "tc_train_workspace_bytes": (C.c_size_t, [C.POINTER(EncoderDesc), C.c_int, C.c_int]),
    "tc_encoder_forward_train": (C.c_int, [C.POINTER(EncoderDesc), vp, vp, C.c_int, C.c_int, vp, vp, vp, vp,
                                           C.c_size_t, vp]),
    "tc_encoder_backward": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderGrads), vp, vp, C.c_int, C.c_int,
                                      vp, vp, vp, C.c_size_t, vp]),
    "tc_encoder_backward_notify": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderGrads), vp, vp, C.c_int,
                                             C.c_int, vp, vp, vp, C.c_size_t, vp, vp, vp]),
    "tc_mlm_loss": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, vp]),
    "tc_mlm_mask": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_double,
                              C.c_int, vp, C.c_int, vp]),
    "tc_crop_pad": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "tc_adamw_step": (C.c_int, [vp, vp, vp, vp, C.c_long] + [C.c_float] * 5 + [C.c_int, C.c_float, vp]),
    "tc_prof_enable": (C.c_int, [C.c_int]),
    "tc_prof_reset": (C.c_int, []),
    "tc_prof_collect": (C.c_int, [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
# bumped by optimisers that update parameter memory without torch noticing; part of every
# packed-weight cache key
weights_epoch = [0]
_lib = None


def load():
    """Load the shared library once; raises HipLibraryError when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `make -C transcodon_amd/csrc` "
                "(there is no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.tc_version() != ABI_VERSION:
            raise HipLibraryError(f"{LIB_PATH} has ABI version {lib.tc_version()}, this binding was written for "
                                  f"{ABI_VERSION}: rebuild it with `make -C transcodon_amd/csrc`")
        for which, struct in enumerate((LayerWeights, EncoderDesc, LayerGrads, EncoderGrads)):
            if lib.tc_struct_bytes(which) != C.sizeof(struct):
                raise HipLibraryError(f"{struct.__name__}: {C.sizeof(struct)} bytes here, "
                                      f"{lib.tc_struct_bytes(which)} in {LIB_PATH}")
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().tc_last_error().decode(errors="replace")
        raise HipLibraryError(f"{what or 'libtranscodon_hip'} failed ({rc}): {msg}")


def prof_collect(kind):
    n, ms, work = C.c_int64(0), C.c_double(0), C.c_double(0)
    check(load().tc_prof_collect(kind, C.byref(n), C.byref(ms), C.byref(work)), "tc_prof_collect")
    return n.value, ms.value, work.value
