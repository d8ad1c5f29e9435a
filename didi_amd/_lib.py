"""ctypes binding of libdidi_hip.so (the C ABI in include/didi_hip.h).

The structures below mirror the header field-for-field (same order, same types).  The library is
built in-tree by ``__graft_entry__.build()`` / ``make -C didi_amd/csrc``; if it is missing or
fails to load, importing this module raises: there is no fallback path.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- first: its bundled libamdhip64.so.7 must be the one HIP runtime in the process

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIDI_HIP_LIB") or os.path.join(_HERE, "libdidi_hip.so")  # DIDI_HIP_LIB: A/B runs against another build

u16p = C.POINTER(C.c_uint16)
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)

OK = 0
PREC_BF16, PREC_FP16, PREC_BF16X3, PREC_FP16X2, PREC_FP16X3 = 1, 2, 3, 4, 5
FMT_BF16, FMT_FP16 = 0, 1
ACT_NONE, ACT_GELU, ACT_TANH, ACT_RELU, ACT_GELU_NEW = 0, 1, 2, 3, 4
STEP_NONE, STEP_INIT, STEP_DDPM, STEP_EULER = 0, 1, 2, 3
ERR_FAULT = -5
FAULT_LN_EXCHANGE, FAULT_FP16_RANGE = 1, 2
ABI_VERSION = 7
# didi_set_option keys (tuning switches for A/B runs; defaults = production path)
OPTIONS = {"GEMM_LN": 0, "GEMM_SMALL": 1, "GEMM_BIG": 2, "GEMM_PP": 3, "GEMM_MT": 4, "BIG_MIN_M": 5, "SMALL_MAX_M": 6, "X2_GEMMS": 7, "FUSED_ATTN": 8, "HI_RESIDUAL": 9, "LO8_RESIDUAL": 10}


class StepArgs(C.Structure):
    _fields_ = [
        ("mode", C.c_int), ("M", C.c_int), ("S", C.c_int), ("H", C.c_int),
        ("X", C.c_void_p), ("X0", C.c_void_p), ("E", C.c_void_p),
        ("cx0", C.c_float), ("ce", C.c_float), ("ratio", C.c_float), ("cin", C.c_float),
        ("seed", C.c_uint64), ("offset", C.c_uint64),
        ("x_out", C.c_void_p), ("xin_out", C.c_void_p),
        ("time_embeds", C.c_void_p), ("t", C.c_int), ("time_idx", C.c_void_p),
        ("pos_emb", C.c_void_p), ("tok_type0", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p),
        ("ln_eps", C.c_float), ("h_hi", C.c_void_p), ("h_lo", C.c_void_p), ("plane_format", C.c_int), ("act_dither", C.c_int),
    ]


class GemmArgs(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("precision", C.c_int),
        ("A_hi", C.c_void_p), ("A_lo", C.c_void_p), ("lda", C.c_int),
        ("W_hi", C.c_void_p), ("W_lo", C.c_void_p), ("ldw", C.c_int),
        ("bias", C.c_void_p), ("act", C.c_int),
        ("R_f32", C.c_void_p), ("R_hi", C.c_void_p), ("R_lo", C.c_void_p), ("ldr", C.c_int),
        ("C_f32", C.c_void_p), ("C_hi", C.c_void_p), ("C_lo", C.c_void_p), ("ldc", C.c_int),
        ("fault", C.c_void_p), ("act_dither", C.c_int),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("heads", C.c_int), ("Sq", C.c_int), ("Sk", C.c_int), ("D", C.c_int),
        ("precision", C.c_int), ("causal", C.c_int), ("key_mask", C.c_void_p),
        ("Q_hi", C.c_void_p), ("Q_lo", C.c_void_p), ("ldq", C.c_int),
        ("K_hi", C.c_void_p), ("K_lo", C.c_void_p), ("ldk", C.c_int),
        ("V_hi", C.c_void_p), ("V_lo", C.c_void_p), ("ldv", C.c_int),
        ("O_hi", C.c_void_p), ("O_lo", C.c_void_p), ("ldo", C.c_int),
        ("scale", C.c_float), ("pos_bias", C.c_void_p), ("key_count", C.c_void_p), ("act_dither", C.c_int),
    ]


class AttnTrainArgs(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("dO", C.c_void_p), ("key_mask", C.c_void_p),
        ("B", C.c_int), ("heads", C.c_int), ("Sq", C.c_int), ("Sk", C.c_int), ("D", C.c_int),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("ldo", C.c_int), ("causal", C.c_int),
        ("scale", C.c_float), ("drop_p", C.c_float), ("seed", C.c_uint64), ("offset", C.c_uint64),
        ("O", C.c_void_p), ("dQ", C.c_void_p), ("dK", C.c_void_p), ("dV", C.c_void_p), ("pos_bias", C.c_void_p),
    ]


class Linear(C.Structure):
    _fields_ = [("w_hi", C.c_void_p), ("w_lo", C.c_void_p), ("bias", C.c_void_p), ("N", C.c_int), ("K", C.c_int)]


class DecoderLayer(C.Structure):
    _fields_ = [
        ("qkv", Linear), ("attn_out", Linear), ("ln1_g", C.c_void_p), ("ln1_b", C.c_void_p),
        ("cq", Linear), ("ckv", Linear), ("cross_out", Linear), ("ln2_g", C.c_void_p), ("ln2_b", C.c_void_p),
        ("ffn_up", Linear), ("ffn_down", Linear), ("ln3_g", C.c_void_p), ("ln3_b", C.c_void_p),
    ]


class Decoder(C.Structure):
    _fields_ = [
        ("H", C.c_int), ("heads", C.c_int), ("I", C.c_int), ("L", C.c_int), ("ln_eps", C.c_float),
        ("time_embeds", C.c_void_p), ("pos_emb", C.c_void_p), ("tok_type0", C.c_void_p),
        ("emb_ln_g", C.c_void_p), ("emb_ln_b", C.c_void_p), ("layers_host", C.POINTER(DecoderLayer)),
    ]


class ForwardArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("S", C.c_int), ("C", C.c_int), ("precision", C.c_int),
        ("step", StepArgs),
        ("kv_hi", C.c_void_p), ("kv_lo", C.c_void_p), ("ctx_mask", C.c_void_p), ("ctx_count", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("x0_f32", C.c_void_p), ("x0_hi", C.c_void_p), ("x0_lo", C.c_void_p),
        ("prof_events_host", C.POINTER(C.c_void_p)), ("prof_capacity", C.c_int), ("prof_used_host", C.POINTER(C.c_int)),
        ("fault", C.c_void_p), ("act_dither", C.c_int),
    ]


class SampleArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("S", C.c_int), ("C", C.c_int), ("precision", C.c_int), ("euler", C.c_int), ("n_calls", C.c_int),
        ("time_row_host", i32p), ("c0_host", f32p), ("c1_host", f32p), ("c2_host", f32p),
        ("init_scale", C.c_float), ("noise", C.c_void_p), ("seed", C.c_uint64), ("base_offset", C.c_uint64),
        ("xt_trace", C.c_void_p), ("kv_hi", C.c_void_p), ("kv_lo", C.c_void_p), ("ctx_mask", C.c_void_p), ("ctx_count", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("x_state", C.c_void_p), ("x0_f32", C.c_void_p), ("x0_hi", C.c_void_p), ("x0_lo", C.c_void_p),
        ("cls_hi", C.c_void_p), ("cls_lo", C.c_void_p), ("V", C.c_int), ("argmax_partial", C.c_void_p),
        ("ids64", C.c_void_p),
        ("prof_calls_host", C.POINTER(C.c_int)), ("n_prof_calls", C.c_int), ("prof_events_host", C.POINTER(C.c_void_p)), ("prof_capacity", C.c_int),
        ("prof_used_host", C.POINTER(C.c_int)), ("clamp_emb", C.c_void_p), ("fault", C.c_void_p),
        ("precision_host", C.POINTER(C.c_int)), ("dec_variants_host", C.POINTER(C.c_void_p)), ("n_variants", C.c_int), ("act_dither", C.c_int),
    ]


# every symbol include/didi_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("didi_version", C.c_int, []),
    ("didi_error_string", C.c_char_p, [C.c_int]),
    ("didi_fault_read", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]),
    ("didi_fault_string", C.c_char_p, [C.c_uint32]),
    ("didi_set_option", C.c_int, [C.c_int, C.c_int]),
    ("didi_get_option", C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    ("didi_split_bf16", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("didi_split_fp16", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("didi_philox_normal", C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p]),
    ("didi_step_embed_ln", C.c_int, [C.POINTER(StepArgs), C.c_void_p]),
    ("didi_embed_gather", C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p]),
    ("didi_layernorm", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("didi_gemm", C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    ("didi_gemm_ln_supported", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("didi_gemm_ln_stats_bytes", C.c_size_t, [C.c_int]),
    ("didi_gemm_ln", C.c_int, [C.POINTER(GemmArgs), C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_uint32, C.c_void_p]),
    ("didi_gemm_ln_lo8", C.c_int, [C.POINTER(GemmArgs), C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    ("didi_qkv_self_attention_supported", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("didi_qkv_self_attention", C.c_int, [C.POINTER(GemmArgs), C.c_int, C.c_int, C.c_float, C.c_void_p]),
    ("didi_q_cross_attention_supported", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("didi_q_cross_attention", C.c_int, [C.POINTER(GemmArgs), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_float, C.c_void_p]),
    ("didi_gemm_small_workspace_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("didi_gemm_small", C.c_int, [C.POINTER(GemmArgs), C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    ("didi_rmsnorm", C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("didi_mul_split", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("didi_round_argmax_bytes", C.c_size_t, [C.c_int, C.c_int]),
    ("didi_round_ce_bytes", C.c_size_t, [C.c_int, C.c_int]),
    ("didi_round_ce", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("didi_round_argmax", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("didi_round_argmax_static_max_bytes", C.c_size_t, []),
    ("didi_round_argmax_static", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("didi_first_eos", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    ("didi_attention", C.c_int, [C.POINTER(AttnArgs), C.c_void_p]),
    ("didi_key_counts", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("didi_decoder_workspace_bytes", C.c_size_t, [C.POINTER(Decoder), C.c_int, C.c_int, C.c_int]),
    ("didi_decoder_kv_plane_bytes", C.c_size_t, [C.POINTER(Decoder), C.c_int, C.c_int]),
    ("didi_decoder_prepare_context", C.c_int, [C.POINTER(Decoder), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                               C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("didi_decoder_forward", C.c_int, [C.POINTER(Decoder), C.POINTER(ForwardArgs), C.c_void_p]),
    ("didi_sample", C.c_int, [C.POINTER(Decoder), C.POINTER(SampleArgs), C.c_void_p]),
    ("didi_train_noise", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("didi_train_loss_terms", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("didi_train_mse_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p]),
    ("didi_layernorm_backward_bytes", C.c_size_t, [C.c_int, C.c_int]),
    ("didi_layernorm_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    ("didi_colsum_bytes", C.c_size_t, [C.c_int, C.c_int]),
    ("didi_colsum", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("didi_transpose_split", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("didi_transpose_split_colsum_bytes", C.c_size_t, [C.c_int, C.c_int]),
    ("didi_transpose_split_colsum", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    ("didi_split_both_ex", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    ("didi_gemm_wgrad_workspace_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("didi_gemm_wgrad", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    ("didi_gemm_dgrad_supported", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("didi_gemm_dgrad", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    ("didi_split_both", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    ("didi_act_backward", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    ("didi_axpby", C.c_int, [C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("didi_ce_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    ("didi_act_forward", C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("didi_embedding_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_int64, C.c_void_p, C.c_void_p]),
    ("didi_attention_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    ("didi_attention_train_forward", C.c_int, [C.c_void_p, C.c_void_p]),
    ("didi_attention_train_backward", C.c_int, [C.c_void_p, C.c_void_p]),
    ("didi_dropout_add", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    ("didi_dropout_add_layernorm", C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                             C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("didi_adamw_step", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    ("didi_sumsq_accumulate", C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    ("didi_train_embed_sum", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    ("didi_seq_sum", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("didi_train_target_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("didi_event_create", C.c_int, [C.POINTER(C.c_void_p)]),
    ("didi_event_elapsed_ms", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    ("didi_event_destroy", C.c_int, [C.c_void_p]),
]


class DidiHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C didi_amd/csrc`.  didi_amd has no non-HIP fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.didi_version() != ABI_VERSION:
        raise ImportError(f"libdidi_hip ABI version {lib.didi_version()} != {ABI_VERSION}")
    # the library never reads the environment; the tools' DIDI_<OPTION>=<int> variables are applied here, through the ABI
    for name, key in OPTIONS.items():
        v = os.environ.get("DIDI_" + name)
        if v is not None and lib.didi_set_option(key, int(v)) != OK:
            raise ImportError(f"DIDI_{name}={v}: rejected by didi_set_option")
    return lib


lib = _load()


def check(rc, what=""):
    if rc != OK:
        raise DidiHipError(f"{what}: {lib.didi_error_string(rc).decode()} ({rc})")


def set_option(name, value):
    check(lib.didi_set_option(OPTIONS[name], int(value)), f"set_option {name}")


def get_option(name):
    v = C.c_int()
    check(lib.didi_get_option(OPTIONS[name], C.byref(v)), f"get_option {name}")
    return v.value


def check_fault(fault, what=""):
    """Wait for the current stream and raise if the device fault word (uint32 tensor of one element) is set."""
    bits = C.c_uint32()
    rc = lib.didi_fault_read(fault.data_ptr(), C.byref(bits), current_stream())
    if rc == ERR_FAULT:
        raise DidiHipError(f"{what}: device fault 0x{bits.value:x}: {lib.didi_fault_string(bits.value).decode()}")
    check(rc, what)


def ptr(t):
    """Device (or host) address of a torch tensor, or None."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
