"""ctypes binding of libmesm_gfx950.so (the C-ABI declared in include/mesm_gfx950.h).

There is no CPU fallback: if the shared object is missing or a call returns a non-zero
status the caller gets an exception.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MESM_LIB_PATH") or os.path.join(_HERE, "libmesm_gfx950.so")  # override: tuning builds (tools/build_variant.sh)

ACT_NONE, ACT_RELU, ACT_PRELU = 0, 1, 2
LAYOUT_REDUCE_CONTIG, LAYOUT_OUTER_CONTIG = 0, 1
MASK_KPAD, MASK_T2V_QUIRK, MASK_CAUSAL = 0, 1, 2

c_f32p = ctypes.c_void_p  # device pointers travel as opaque addresses
c_ptr = ctypes.c_void_p


class MesmError(RuntimeError):
    pass


class GemmArgs(ctypes.Structure):
    _fields_ = [
        ("A", c_ptr), ("A2", c_ptr), ("B", c_ptr), ("B2", c_ptr), ("C", c_ptr),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("a_layout", ctypes.c_int32), ("b_layout", ctypes.c_int32),
        ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("ldc", ctypes.c_int64),
        ("bias", c_ptr), ("residual", c_ptr), ("ldr", ctypes.c_int64),
        ("aux", c_ptr), ("ldaux", ctypes.c_int64),
        ("slope", c_ptr), ("dslope", c_ptr), ("colsum", c_ptr),
        ("a_act", ctypes.c_int32), ("b_act", ctypes.c_int32),
        ("a_drop_p", ctypes.c_float), ("b_drop_p", ctypes.c_float),
        ("a_drop_seed", ctypes.c_uint32), ("b_drop_seed", ctypes.c_uint32),
        ("e_act", ctypes.c_int32), ("e_actgrad", ctypes.c_int32),
        ("e_drop_p", ctypes.c_float), ("e_drop_seed", ctypes.c_uint32),
        ("out_scale", ctypes.c_float),
        ("accumulate", ctypes.c_int32), ("split_k", ctypes.c_int32),
        ("seed_offset", c_ptr), ("dslope_ws", c_ptr),
        ("pre_out", c_ptr), ("ldpre", ctypes.c_int64),
        ("e_drop_row0", ctypes.c_int32), ("reserved0", ctypes.c_int32),
    ]


class AttnArgs(ctypes.Structure):
    _fields_ = [
        ("q", c_ptr), ("k", c_ptr), ("v", c_ptr), ("o", c_ptr), ("lse", c_ptr),
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("Lq", ctypes.c_int32),
        ("Lk", ctypes.c_int32), ("dk", ctypes.c_int32), ("dv", ctypes.c_int32),
        ("q_bs", ctypes.c_int64), ("q_ls", ctypes.c_int64),
        ("k_bs", ctypes.c_int64), ("k_ls", ctypes.c_int64),
        ("v_bs", ctypes.c_int64), ("v_ls", ctypes.c_int64),
        ("o_bs", ctypes.c_int64), ("o_ls", ctypes.c_int64),
        ("kpad", c_ptr), ("qpad", c_ptr), ("mask_mode", ctypes.c_int32),
        ("scale", ctypes.c_float), ("drop_p", ctypes.c_float),
        ("drop_seed", ctypes.c_uint32),
        ("d_o", c_ptr), ("dq", c_ptr), ("dk_", c_ptr), ("dv_", c_ptr),
        ("seed_offset", c_ptr), ("mask_group", ctypes.c_int32),
        ("q2", c_ptr), ("k2", c_ptr), ("dq2", c_ptr), ("dk2", c_ptr), ("k_add", c_ptr),
        ("mask_mod", c_ptr),
    ]


class LnArgs(ctypes.Structure):
    _fields_ = [
        ("x", c_ptr), ("gamma", c_ptr), ("beta", c_ptr), ("y", c_ptr), ("mean", c_ptr), ("rstd", c_ptr),
        ("rows", ctypes.c_int64), ("D", ctypes.c_int32), ("eps", ctypes.c_float),
        ("drop_p", ctypes.c_float), ("drop_seed", ctypes.c_uint32), ("seed_offset", c_ptr),
        ("add", c_ptr), ("y2", c_ptr),
        ("dy", c_ptr), ("dx", c_ptr), ("dgamma", c_ptr), ("dbeta", c_ptr),
        ("accumulate_dx", ctypes.c_int32), ("drop2_p", ctypes.c_float), ("drop2_seed", ctypes.c_uint32),
        ("relu_in", ctypes.c_int32),
        ("dx2", c_ptr), ("dyb", c_ptr), ("addend", c_ptr),
    ]


class GlueArgs(ctypes.Structure):
    _fields_ = [("op", ctypes.c_int32), ("reserved0", ctypes.c_int32), ("i", ctypes.c_int32 * 6), ("n", ctypes.c_int64 * 2),
                ("p", c_ptr * 8)]


GLUE_TOKEN_MIX_FWD, GLUE_TOKEN_MIX_BWD, GLUE_GATHER_ROWS_FWD, GLUE_GATHER_ROWS_BWD = 1, 2, 3, 4
GLUE_UNSTACK_ROWS, GLUE_STACK_ROWS, GLUE_ADD_TILE, GLUE_GATHER_ADD = 5, 6, 7, 8


class CritFwdArgs(ctypes.Structure):
    _fields_ = [
        ("weights", c_ptr), ("lv", c_ptr), ("total", c_ptr), ("n_valid", c_ptr),
        ("N", ctypes.c_int32), ("n_slots", ctypes.c_int32),
        ("n_set", ctypes.c_int32), ("Q", ctypes.c_int32), ("Tmax", ctypes.c_int32),
        ("w_span", ctypes.c_float), ("w_giou", ctypes.c_float), ("w_class", ctypes.c_float), ("eos_coef", ctypes.c_float),
        ("reserved0", ctypes.c_int32),
        ("tgt_cxw", c_ptr), ("tgt_xx", c_ptr), ("tgt_off", c_ptr),
        ("set_logits", c_ptr * 8), ("set_spans", c_ptr * 8), ("set_match", c_ptr * 8), ("set_slot", ctypes.c_int32 * 8),
        ("sal_on", ctypes.c_int32), ("sal_L", ctypes.c_int32), ("sal_P", ctypes.c_int32), ("sal_slot", ctypes.c_int32),
        ("rank_coef", ctypes.c_float), ("margin", ctypes.c_float),
        ("s_pos", c_ptr), ("s_neg", c_ptr), ("sal_label", c_ptr), ("vmask", c_ptr), ("pos_idx", c_ptr), ("neg_idx", c_ptr),
        ("fw_on", ctypes.c_int32), ("fw_Lw", ctypes.c_int32), ("fw_C", ctypes.c_int32), ("fw_slot", ctypes.c_int32),
        ("fw_eps", ctypes.c_float), ("reserved1", ctypes.c_int32),
        ("logit", c_ptr), ("label", c_ptr), ("words_mask", c_ptr), ("row_loss", c_ptr), ("row_lse", c_ptr), ("correct", c_ptr),
        ("ss_on", ctypes.c_int32), ("ss_D", ctypes.c_int32), ("ss_Lv", ctypes.c_int32), ("ss_Le", ctypes.c_int32),
        ("ss_slot", ctypes.c_int32), ("ss_tau", ctypes.c_float),
        ("pv", c_ptr), ("cmask", c_ptr), ("ew", c_ptr), ("wmask", c_ptr), ("ss_pos", c_ptr),
        ("cn", c_ptr), ("wn", c_ptr), ("stats", c_ptr), ("sim", c_ptr),
    ]


class CritBwdArgs(ctypes.Structure):
    _fields_ = [
        ("g_total", c_ptr), ("weights", c_ptr), ("n_valid", c_ptr),
        ("N", ctypes.c_int32), ("Q", ctypes.c_int32), ("n_set", ctypes.c_int32), ("eos_coef", ctypes.c_float),
        ("tgt_cxw", c_ptr), ("tgt_xx", c_ptr), ("tgt_off", c_ptr),
        ("set_logits", c_ptr * 8), ("set_spans", c_ptr * 8), ("set_match", c_ptr * 8),
        ("set_dlogits", c_ptr * 8), ("set_dspans", c_ptr * 8), ("set_slot", ctypes.c_int32 * 8),
        ("sal_on", ctypes.c_int32), ("sal_L", ctypes.c_int32), ("sal_P", ctypes.c_int32), ("sal_slot", ctypes.c_int32),
        ("rank_coef", ctypes.c_float), ("margin", ctypes.c_float),
        ("s_pos", c_ptr), ("s_neg", c_ptr), ("sal_label", c_ptr), ("vmask", c_ptr), ("pos_idx", c_ptr), ("neg_idx", c_ptr),
        ("ds_pos", c_ptr), ("ds_neg", c_ptr),
        ("fw_on", ctypes.c_int32), ("fw_Lw", ctypes.c_int32), ("fw_C", ctypes.c_int32), ("fw_slot", ctypes.c_int32),
        ("fw_eps", ctypes.c_float), ("reserved0", ctypes.c_int32),
        ("logit", c_ptr), ("label", c_ptr), ("row_lse", c_ptr), ("words_mask", c_ptr), ("dlogit", c_ptr),
        ("ss_on", ctypes.c_int32), ("ss_D", ctypes.c_int32), ("ss_Lv", ctypes.c_int32), ("ss_Le", ctypes.c_int32),
        ("ss_slot", ctypes.c_int32), ("ss_tau", ctypes.c_float),
        ("cn", c_ptr), ("wn", c_ptr), ("ss_pos", c_ptr), ("sim", c_ptr), ("stats", c_ptr), ("cmask", c_ptr), ("wmask", c_ptr),
        ("dpv", c_ptr), ("dew", c_ptr),
    ]


# name -> (restype, argtypes); mirrors include/mesm_gfx950.h one to one.
_i32, _i64, _f32, _u32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_uint32
PROTOTYPES = {
    "mesm_abi_version": (ctypes.c_int, []),
    "mesm_arch": (ctypes.c_char_p, []),
    "mesm_gemm_f32": (ctypes.c_int, [ctypes.POINTER(GemmArgs), c_ptr]),
    "mesm_gemm_group": (ctypes.c_int, [ctypes.POINTER(GemmArgs), _i32, c_ptr]),
    "mesm_gemm_flush_side": (ctypes.c_int, [c_ptr]),
    "mesm_gemm_get_bf16x": (ctypes.c_int, []),
    "mesm_gemm_set_switches": (ctypes.c_int, [_i32, _i32]),
    "mesm_gemm_drop_side": (ctypes.c_int, []),
    "mesm_gemm_tape": (ctypes.c_int, [_i32]),
    "mesm_gemm_tape_entry": (ctypes.c_int, [c_ptr, _i32, _i32, ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "mesm_gemm_tape_size": (ctypes.c_int, []),
    "mesm_gemm_tape_bytes": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "mesm_gemm_tape_replay": (ctypes.c_int, [c_ptr, _i32, ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_double)]),
    "mesm_layernorm_fwd": (ctypes.c_int, [c_ptr] * 6 + [_i64, _i32, _f32, _f32, _u32, c_ptr, c_ptr]),
    "mesm_layernorm_bwd": (ctypes.c_int, [c_ptr] * 8 + [_i64, _i32, _i32, _f32, _u32, c_ptr, c_ptr]),
    "mesm_layernorm_bwd2": (ctypes.c_int, [c_ptr] * 8 + [_i64, _i32, _i32, _f32, _u32, c_ptr, c_ptr, _f32, _u32, c_ptr]),
    "mesm_layernorm_fwd2": (ctypes.c_int, [c_ptr] * 6 + [_i64, _i32, _f32, _f32, _u32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mesm_layernorm_bwd3": (ctypes.c_int, [c_ptr] * 8 + [_i64, _i32, _i32, _f32, _u32, c_ptr, c_ptr, _f32, _u32,
                                                        c_ptr, c_ptr, c_ptr]),
    "mesm_layernorm_fwd_group": (ctypes.c_int, [ctypes.POINTER(LnArgs), _i32, c_ptr]),
    "mesm_layernorm_bwd_group": (ctypes.c_int, [ctypes.POINTER(LnArgs), _i32, c_ptr]),
    "mesm_attn_bwd_accumulates_dq": (ctypes.c_int, [_i32] * 7),
    "mesm_attn_fwd_group": (ctypes.c_int, [ctypes.POINTER(AttnArgs), _i32, c_ptr]),
    "mesm_attn_bwd_group": (ctypes.c_int, [ctypes.POINTER(AttnArgs), _i32, c_ptr]),
    "mesm_attn_fwd": (ctypes.c_int, [ctypes.POINTER(AttnArgs), c_ptr]),
    "mesm_attn_bwd": (ctypes.c_int, [ctypes.POINTER(AttnArgs), c_ptr]),
    "mesm_sine_pos_fwd": (ctypes.c_int, [c_ptr, c_ptr, _i32, _i32, _i32, c_ptr]),
    "mesm_query_sine_fwd": (ctypes.c_int, [c_ptr, c_ptr, _i64, _i32, c_ptr]),
    "mesm_query_sine_bwd": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, _i64, _i32, c_ptr]),
    "mesm_dropout": (ctypes.c_int, [c_ptr, c_ptr, _i64, _f32, _u32, c_ptr, c_ptr]),
    "mesm_windows": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, _i32, _i32, c_ptr]),
    "mesm_ref_update_fwd": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, _i64, _f32, c_ptr]),
    "mesm_ref_update_bwd": (ctypes.c_int, [c_ptr] * 5 + [_i64, _f32, c_ptr]),
    "mesm_ref_init_fwd": (ctypes.c_int, [c_ptr, c_ptr, _i32, _i32, c_ptr]),
    "mesm_ref_init_bwd": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, _i32, _i32, c_ptr]),
    "mesm_ref_step_fwd": (ctypes.c_int, [c_ptr, _i32, c_ptr, c_ptr, _f32] + [c_ptr] * 5 + [_i64, _i32, c_ptr]),
    "mesm_ref_step_bwd": (ctypes.c_int, [c_ptr] * 3 + [_f32] + [c_ptr] * 8 + [_i64, _i32, c_ptr]),
    "mesm_ref_init_sine_bwd": (ctypes.c_int, [c_ptr] * 7 + [_i32, _i32, _i32, c_ptr]),
    "mesm_qsine_scale_fwd": (ctypes.c_int, [c_ptr] * 5 + [_i64, _i32, c_ptr]),
    "mesm_qsine_scale_bwd": (ctypes.c_int, [c_ptr] * 9 + [_i64, _i32, c_ptr]),
    "mesm_act_dropout": (ctypes.c_int, [c_ptr, c_ptr, _i64, _i32, c_ptr, _f32, _u32, c_ptr, c_ptr]),
    "mesm_act_bias_bwd": (ctypes.c_int, [c_ptr] * 6 + [_i64, _i32, _i32, c_ptr]),
    "mesm_nll_smooth_fwd": (ctypes.c_int, [c_ptr] * 6 + [_i64, _i32, _f32, c_ptr]),
    "mesm_nll_smooth_bwd": (ctypes.c_int, [c_ptr] * 5 + [_i64, _i32, _f32, c_ptr]),
    "mesm_saliency_loss_fwd": (ctypes.c_int, [c_ptr] * 6 + [_i32, _i32, _i32, _f32, _f32, c_ptr, c_ptr]),
    "mesm_saliency_loss_bwd": (ctypes.c_int, [c_ptr] * 6 + [_i32, _i32, _i32, _f32, _f32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mesm_match": (ctypes.c_int, [c_ptr] * 5 + [_i32, _i32, _i32, _f32, _f32, _f32, c_ptr, c_ptr, c_ptr]),
    "mesm_grad_sumsq": (ctypes.c_int, [c_ptr, _i64, c_ptr, ctypes.POINTER(ctypes.c_int32), c_ptr, c_ptr]),
    "mesm_clip_grad": (ctypes.c_int, [c_ptr, _i64, c_ptr, _i32, _f32, c_ptr, c_ptr]),
    "mesm_adamw_step": (ctypes.c_int, [c_ptr] * 5 + [_i64, c_ptr, _i32, _f32, c_ptr, _f32, _f32, _f32, _f32,
                                       c_ptr, c_ptr, c_ptr]),
    "mesm_set_loss_fwd_layers": (ctypes.c_int, [c_ptr, c_ptr, _i32, c_ptr, c_ptr, c_ptr, _i32, _i32, _i32, _f32, _f32, _f32,
                                                _f32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mesm_set_loss_fwd": (ctypes.c_int, [c_ptr] * 5 + [_i32, _i32, _i32, _f32, _f32, _f32, _f32, c_ptr, c_ptr, c_ptr]),
    "mesm_set_loss_bwd": (ctypes.c_int, [c_ptr] * 6 + [_i32, _i32, _f32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mesm_rec_ss_fwd": (ctypes.c_int, [c_ptr, c_ptr, _i32, c_ptr, c_ptr, _i32, c_ptr, _i32, _i32, _f32] + [c_ptr] * 6),
    "mesm_rec_ss_bwd": (ctypes.c_int, [c_ptr] * 7 + [_i32, _i32, _i32, _i32, _f32] + [c_ptr] * 4),
    "mesm_rec_fw_reduce": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, _i32, _i32, c_ptr, c_ptr]),
    "mesm_rec_fw_rowgrad": (ctypes.c_int, [c_ptr, _i32, _i32, c_ptr, c_ptr, c_ptr]),
    "mesm_rowdot_fwd": (ctypes.c_int, [c_ptr, c_ptr, _i32, _i32, _i32, _f32, c_ptr, c_ptr]),
    "mesm_rowdot_bwd": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, _i32, _i32, _i32, _f32, c_ptr, c_ptr, c_ptr]),
    "mesm_text_prep": (ctypes.c_int, [c_ptr, _i32, _i32, _i32, _i32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mesm_stack_rows": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, _i32, c_ptr, _i32, c_ptr]),
    "mesm_unstack_rows": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, _i32, _i64, c_ptr]),
    "mesm_prepend_fwd": (ctypes.c_int, [c_ptr] * 9 + [_i32, _i32, _i32, _i32, _i32, c_ptr]),
    "mesm_prepend_bwd": (ctypes.c_int, [c_ptr] * 6 + [_i32, _i32, _i32, _i32, c_ptr]),
    "mesm_split_token_fwd": (ctypes.c_int, [c_ptr] * 4 + [_i32, _i32, _i32, _i32, c_ptr]),
    "mesm_split_token_bwd": (ctypes.c_int, [c_ptr] * 4 + [_i32, _i32, _i32, _i32, c_ptr]),
    "mesm_token_mix_fwd": (ctypes.c_int, [c_ptr] * 6 + [_i64, _i32, c_ptr]),
    "mesm_token_mix_bwd": (ctypes.c_int, [c_ptr] * 6 + [_i64, _i32, c_ptr]),
    "mesm_gather_rows_fwd": (ctypes.c_int, [c_ptr] * 5 + [_i64, _i32, _i32, c_ptr]),
    "mesm_gather_rows_bwd": (ctypes.c_int, [c_ptr] * 6 + [_i64, _i32, _i32, c_ptr]),
    "mesm_add_wrap": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, _i64, _i64, c_ptr]),
    "mesm_add_n": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _i32, c_ptr, _i64, c_ptr]),
    "mesm_step_begin": (ctypes.c_int, [c_ptr, _i32, _i32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mesm_skinny_linear_bwd": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, _i64, _i32, _i32, _i32, c_ptr]),
    "mesm_clip_embed": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, _i64, _i32, _i32, _i32, c_ptr]),
    "mesm_layernorm_f16": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, _i64, _i32, _f32, c_ptr]),
    "mesm_gemm_f16": (ctypes.c_int, [c_ptr, _i32, _i64, c_ptr, _i64, c_ptr, c_ptr, _i64, c_ptr, _i32, _i64,
                                     _i32, _i32, _i32, _i32, c_ptr]),
    "mesm_text_pool": (ctypes.c_int, [c_ptr, _i32, c_ptr, _i32, _i32, _i32, _i32, _i32, _i32, c_ptr, c_ptr, c_ptr]),
    "mesm_embed_rows": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, _i64, _i32, _i32, c_ptr]),
    "mesm_set_loss_fwd_nv": (ctypes.c_int, [c_ptr] * 5 + [_i32, _i32, _i32, _f32, _f32, _f32, _f32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mesm_set_loss_bwd_nv": (ctypes.c_int, [c_ptr] * 6 + [_i32, _i32, _f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mesm_saliency_loss_fwd_nv": (ctypes.c_int, [c_ptr] * 6 + [_i32, _i32, _i32, _f32, _f32, c_ptr, c_ptr, c_ptr]),
    "mesm_saliency_loss_bwd_nv": (ctypes.c_int, [c_ptr] * 6 + [_i32, _i32, _i32, _f32, _f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mesm_rec_ss_fwd_nv": (ctypes.c_int, [c_ptr, c_ptr, _i32, c_ptr, c_ptr, _i32, c_ptr, _i32, _i32, _f32] + [c_ptr] * 7),
    "mesm_rec_ss_bwd_nv": (ctypes.c_int, [c_ptr] * 7 + [_i32, _i32, _i32, _i32, _f32] + [c_ptr] * 5),
    "mesm_rec_fw_reduce_nv": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, _i32, _i32, c_ptr, c_ptr, c_ptr]),
    "mesm_rec_fw_rowgrad_nv": (ctypes.c_int, [c_ptr, _i32, _i32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mesm_ddp_unique_id": (ctypes.c_int, [c_ptr]),
    "mesm_ddp_init": (ctypes.c_int, [c_ptr, _i32, _i32, ctypes.POINTER(ctypes.c_void_p)]),
    "mesm_ddp_allreduce": (ctypes.c_int, [c_ptr, c_ptr, _i64, c_ptr, _i32]),
    "mesm_ddp_wait": (ctypes.c_int, [c_ptr, c_ptr]),
    "mesm_ddp_destroy": (ctypes.c_int, [c_ptr]),
    "mesm_ddp_count": (ctypes.c_int, [c_ptr, ctypes.POINTER(ctypes.c_int32)]),
    "mesm_ddp_last_error": (ctypes.c_char_p, []),
    "mesm_weighted_sum": (ctypes.c_int, [c_ptr, c_ptr, _i32, c_ptr, c_ptr]),
    "mesm_scale_vec": (ctypes.c_int, [c_ptr, c_ptr, _i32, c_ptr, c_ptr]),
    "mesm_fill_ranges": (ctypes.c_int, [c_ptr, c_ptr, _i32, c_ptr]),
    "mesm_glue_group": (ctypes.c_int, [ctypes.POINTER(GlueArgs), _i32, c_ptr]),
    "mesm_criterion_fwd": (ctypes.c_int, [ctypes.POINTER(CritFwdArgs), c_ptr]),
    "mesm_criterion_bwd": (ctypes.c_int, [ctypes.POINTER(CritBwdArgs), c_ptr]),
}

_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MesmError(
                "libmesm_gfx950.so is missing (%s): run `python -m mesm_amd.build` — "
                "there is no fallback path" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise MesmError("%s failed with status %d" % (what, rc))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def set_device_index(idx):
    """(kept for callers of round 5: the device is no longer remembered -- ADVICE r5: a sticky index sent the launches of a
    second model on another device, or of direct kernel calls after a forward, to the wrong device's stream)"""


def stream_ptr():
    """the raw HIP stream torch would launch on right now: the CURRENT device's current stream, through the two C entry
    points (torch.cuda.current_stream() builds a Stream object and looks the device up in Python: 1.9 ms per eager step
    over its ~1,900 launches; these two calls cost a third of a microsecond)"""
    if _raw_stream is not None and _get_device is not None:
        return ctypes.c_void_p(_raw_stream(_get_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device address of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MesmError("mesm_amd kernels need tensors on an MI355X device; got a %s tensor "
                            "(the CPU oracle lives in oracle/, it is not a fallback)" % t.device)
