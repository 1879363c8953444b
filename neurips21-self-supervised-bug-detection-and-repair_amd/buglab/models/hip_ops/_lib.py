"""ctypes table of libbuglab_hip.so (include/buglab_hip.h): structures, prototypes, loader, argument helpers.
Part of buglab.models.hip_ops (split in round 6: this file = the C ABI as Python sees it; _streams.py = stream policy and
live timing; __init__.py = the operators and their autograd Functions).  No CPU fallback: HipOpsUnavailable."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int8, c_int32, c_int64, c_uint32, c_void_p
from typing import List, NamedTuple, Optional, Sequence, Tuple

import torch

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, ACT_GELU, ACT_GELU_AGG = 0, 1, 2, 3, 4, 5
# "gelu_aggregated": GELU on the aggregate of a segmented max (gelu(max x)) instead of on every item (max gelu(x)) -- only
# meaningful as a message-passing layer's `msg_act` / segment_max's `act`
_ACTS = {"none": ACT_NONE, "relu": ACT_RELU, "sigmoid": ACT_SIGMOID, "tanh": ACT_TANH, "gelu": ACT_GELU,
         "gelu_aggregated": ACT_GELU_AGG}


def message_activation_code(activation: str, placement: str = "aggregated") -> str:
    """(message_activation, message_activation_placement) of an MlpMessagePassingLayer -> the `msg_act` name of mp_layer()."""
    if activation not in ("gelu", "none"):
        raise ValueError(f"message_activation must be 'gelu' or 'none' (got {activation!r})")
    if placement not in ("aggregated", "message"):
        raise ValueError(f"message_activation_placement must be 'aggregated' or 'message' (got {placement!r})")
    return "gelu_aggregated" if (activation == "gelu" and placement == "aggregated") else activation
LIB_NAME = "libbuglab_hip.so"
LIB_PATH = os.environ.get("BL_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)  # BL_HIP_LIB: tuning builds


class HipOpsUnavailable(RuntimeError):
    pass


class bl_rows_t(Structure):
    _fields_ = [("x", c_void_p * 3), ("idx", c_void_p * 3), ("ld", c_int32 * 3), ("width", c_int32 * 3), ("nsrc", c_int32)]


class bl_rows_packed_t(Structure):
    _fields_ = [("xp", c_void_p * 3), ("idx", c_void_p * 3), ("width", c_int32 * 3), ("nsrc", c_int32)]


class bl_dropout_t(Structure):
    _fields_ = [("p", c_float), ("seed", c_uint32), ("stream", c_uint32)]


class bl_mp_layer_t(Structure):
    _fields_ = [("N", c_int32), ("E", c_int32), ("T", c_int32), ("Din", c_int32), ("Dm", c_int32), ("Dout", c_int32),
                ("msg_src", c_void_p), ("msg_tgt", c_void_p), ("type_ptr", c_void_p), ("tgt_ptr", c_void_p), ("tgt_msgs", c_void_p),
                ("src_ptr", c_void_p), ("src_msgs", c_void_p), ("node_order", c_void_p),
                ("W", c_void_p), ("ln_g", c_void_p), ("ln_b", c_void_p), ("Wd", c_void_p), ("bd", c_void_p),
                ("msg_act", c_int32), ("ln_eps", c_float), ("drop", bl_dropout_t), ("Wt", c_void_p),
                ("Wd_packed", c_void_p), ("Wd_packed_bwd", c_void_p), ("num_hub_slots", c_int32), ("aggregation", c_int32)]


class bl_x6_epi_t(Structure):
    _fields_ = [("form", c_int32), ("bias", c_void_p), ("act", c_int32), ("drop", bl_dropout_t), ("res", c_void_p), ("ld_res", c_int32),
                ("y_packed", c_void_p), ("mask_scale", c_float), ("colsum", c_void_p), ("c_packed", c_void_p)]


class bl_head_view_t(Structure):
    _fields_ = [("p", c_void_p), ("sb", c_int64), ("sh", c_int32), ("sl", c_int32)]


class bl_packed_head_view_t(Structure):
    _fields_ = [("p", c_void_p), ("W", c_int32), ("col0", c_int32), ("hs", c_int32)]


class bl_great_layer_t(Structure):
    _fields_ = [("B", c_int32), ("L", c_int32), ("H", c_int32), ("dk", c_int32), ("T", c_int32), ("FF", c_int32),
                ("row_ptr", c_void_p), ("ekey", c_void_p), ("ecode", c_void_p), ("lens", c_void_p),
                ("bias_f", c_void_p), ("bias_r", c_void_p), ("norm_g", c_void_p), ("norm_b", c_void_p), ("lin1_b", c_void_p), ("lin2_b", c_void_p),
                ("qkv_w", c_void_p), ("out_w", c_void_p), ("lin1_w", c_void_p), ("lin2_w", c_void_p),
                ("qkv_w_bwd", c_void_p), ("out_w_bwd", c_void_p), ("lin1_w_bwd", c_void_p), ("lin2_w_bwd", c_void_p),
                ("ln_eps", c_float), ("drop_attn", bl_dropout_t), ("drop_att_out", bl_dropout_t), ("drop_ff_hidden", bl_dropout_t),
                ("drop_ff_out", bl_dropout_t)]


class bl_great_layer_grads_t(Structure):
    _fields_ = [("qkv_w", c_void_p), ("out_w", c_void_p), ("lin1_w", c_void_p), ("lin1_b", c_void_p), ("lin2_w", c_void_p), ("lin2_b", c_void_p),
                ("norm_g", c_void_p), ("norm_b", c_void_p), ("bias_f", c_void_p), ("bias_r", c_void_p)]


class bl_pack_job_t(Structure):
    _fields_ = [("w", c_void_p), ("out", c_void_p), ("kind", c_int32), ("G", c_int32), ("K", c_int32), ("N", c_int32),
                ("first_block", c_int32), ("pad_", c_int32)]


class bl_bug_loss_t(Structure):
    _fields_ = [("B", c_int32), ("C", c_int32), ("Rt", c_int32), ("Rv", c_int32), ("Rs", c_int32), ("G", c_int32),
                ("loc_scores", c_void_p), ("repair_logits", c_void_p), ("loc_group_ptr", c_void_p), ("loc_group_items", c_void_p),
                ("candidate_ptr", c_void_p), ("has_bug", c_void_p), ("correct_candidate_idxs", c_void_p),
                ("repair_group_ptr", c_void_p), ("repair_group_items", c_void_p), ("logit_group", c_void_p * 3),
                ("target", c_void_p * 3), ("ntarget", c_int32 * 3), ("w_buggy", c_float), ("abstain_weight", c_float)]


_SIGNATURES = {
    "bl_version": ([], ctypes.c_int),
    "bl_set_deterministic": ([c_int32], None),
    "bl_get_deterministic": ([], c_int32),
    "bl_set_wgrad_tile": ([c_int32], c_int32),
    "bl_set_wgrad_kchunk_cap": ([c_int32], c_int32),
    "bl_node_update_bwd_ok": ([c_int32, c_int32], c_int32),
    "bl_node_update_bwd": ([c_void_p, c_void_p, c_int32, c_int32, bl_dropout_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_set_fused_node_bwd": ([c_int32], c_int32),
    "bl_last_error": ([], ctypes.c_char_p),
    "bl_embed_subtoken_pool_fwd": ([c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, bl_dropout_t, c_int32, c_void_p, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_embed_subtoken_pool_bwd": ([c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, bl_dropout_t, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_embed_subtoken_pool_bwd_sorted": ([c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, bl_dropout_t, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_embed_subtoken_max_fwd": ([c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, bl_dropout_t, c_int32, c_void_p, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_embed_subtoken_max_bwd": ([c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, bl_dropout_t, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_embed_subtoken_max_bwd_sorted": ([c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, bl_dropout_t, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_gemm_rows": ([POINTER(bl_rows_t), c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, bl_dropout_t, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "bl_gemm_rows_routed": ([POINTER(bl_rows_t), c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "bl_pack_bf16x3": ([c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_pack_weights_x6": ([c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_pack_job_blocks": ([c_int32, c_int32, c_int32, c_int32], c_int64),
    "bl_pack_weights_multi": ([c_void_p, c_int32, c_int32, c_void_p], ctypes.c_int),
    "bl_gemm_rows_x6": ([POINTER(bl_rows_packed_t), c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "bl_gemm_rows_x6w_ok": ([c_int32, c_int32], c_int32),
    "bl_set_rows_tile": ([c_int32], c_int32),
    "bl_packed_weight_elems_x6w": ([c_int32, c_int32, c_int32], c_int64),
    "bl_pack_weights_x6w": ([c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_gemm_rows_x6w": ([POINTER(bl_rows_packed_t), c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "bl_gemm_rows_x6_epi": ([POINTER(bl_rows_packed_t), c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32,
                             bl_dropout_t, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "bl_gemm_rows_x6_epi2": ([POINTER(bl_rows_packed_t), c_void_p, c_int32, c_int32, c_int32, POINTER(bl_x6_epi_t), c_void_p, c_int32, c_void_p], ctypes.c_int),
    "bl_gemm_wgrad_x6": ([POINTER(bl_rows_packed_t), c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64,
                          c_int32, c_void_p], ctypes.c_int),
    "bl_routed_dgrad_vec_ok": ([c_int32, c_int32], c_int32),
    "bl_routed_dgrad_vec": ([c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "bl_routed_dgrad_nodes": ([c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32,
                               c_void_p, c_int32, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "bl_routed_dgrad_nodes_rows": ([c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32,
                                    c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "bl_gemm_wgrad_routed_x6": ([POINTER(bl_rows_packed_t), c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_int32, c_void_p], ctypes.c_int),
    "bl_gemm_wgrad": ([POINTER(bl_rows_t), c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_int32, c_void_p], ctypes.c_int),
    "bl_gemm_wgrad_routed": ([POINTER(bl_rows_t), c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_int32, c_void_p], ctypes.c_int),
    "bl_pack_bf16x3_cols": ([c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_mp_scatter_grad_split": ([c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_mp_layer_saved_bytes": ([c_int32, c_int32, c_int32, c_int32, c_int32], c_int64),
    "bl_mp_layer_workspace_bytes": ([c_int32, c_int32, c_int32, c_int32, c_int32, c_int32], c_int64),
    "bl_mp_layer_packed_weight_elems": ([c_int32, c_int32, c_int32, c_int32], c_int64),
    "bl_mp_layer_weight_image": ([c_int32, c_int32, c_int32], c_int32),
    "bl_mp_layer_fwd": ([POINTER(bl_mp_layer_t), c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_mp_layer_bwd": ([POINTER(bl_mp_layer_t), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32,
                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32], ctypes.c_int),
    "bl_gather_concat_mlp_score_fwd": ([POINTER(bl_rows_t), c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_gather_concat_mlp_score_workspace_bytes": ([c_int32, c_int32, c_int32], c_int64),
    "bl_gather_concat_mlp_score_bwd": ([POINTER(bl_rows_t), c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_int32), c_void_p], ctypes.c_int),
    "bl_localization_scores_saved_bytes": ([c_int32, c_int32, c_int32], c_int64),
    "bl_localization_scores_workspace_bytes": ([c_int32, c_int32, c_int32, c_int32], c_int64),
    "bl_localization_scores_fwd": ([c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_localization_scores_bwd": ([c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_bug_loss_fwd": ([POINTER(bl_bug_loss_t), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_bug_loss_bwd": ([POINTER(bl_bug_loss_t), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_add_layernorm_fwd": ([c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_add_layernorm_fwd_packed": ([c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p], ctypes.c_int),
    "bl_layernorm_bwd_branch": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, bl_dropout_t,
                                 c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_rel_attn_probs_fwd_v": ([POINTER(bl_head_view_t), c_float, POINTER(bl_head_view_t), c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                 c_int32, c_int32, c_void_p, c_void_p, c_void_p, bl_dropout_t, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_rel_attn_probs_bwd_v": ([POINTER(bl_head_view_t), POINTER(bl_head_view_t), c_void_p, POINTER(bl_head_view_t), c_float, c_void_p, c_void_p,
                                 c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, bl_dropout_t, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p], ctypes.c_int),
    "bl_attn_rows_times_v": ([c_void_p, POINTER(bl_head_view_t), c_int32, c_int32, c_int32, c_int32, c_void_p, c_float, POINTER(bl_head_view_t),
                              bl_dropout_t, POINTER(bl_packed_head_view_t), c_void_p], ctypes.c_int),
    "bl_attn_transposed_times_v": ([c_void_p, POINTER(bl_head_view_t), c_float, c_int32, c_int32, c_int32, c_int32, POINTER(bl_head_view_t),
                                    bl_dropout_t, POINTER(bl_packed_head_view_t), c_void_p], ctypes.c_int),
    "bl_great_layer_ok": ([c_int32, c_int32, c_int32, c_int32, c_int32, c_int32], c_int32),
    "bl_great_layer_saved_bytes": ([c_int32, c_int32, c_int32, c_int32, c_int32, c_int32], c_int64),
    "bl_great_layer_workspace_bytes": ([c_int32, c_int32, c_int32, c_int32, c_int32, c_int32], c_int64),
    "bl_great_layer_fwd": ([POINTER(bl_great_layer_t), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_great_layer_bwd": ([POINTER(bl_great_layer_t), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(bl_great_layer_grads_t), c_void_p,
                            c_void_p], ctypes.c_int),
    "bl_rel_attn_bias_fwd": ([c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_rel_attn_bias_bwd": ([c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_masked_softmax_fwd": ([c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_softmax_bwd": ([c_void_p, c_void_p, c_int32, c_int32, c_void_p], ctypes.c_int),
    "bl_rel_attn_probs_ok": ([c_int32, c_int32, c_int32], c_int32),
    "bl_rel_attn_probs_fwd": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                               c_void_p, bl_dropout_t, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_rel_attn_probs_bwd": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                               c_void_p, c_void_p, bl_dropout_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_attn_mm32_ok": ([c_int32, c_int32], c_int32),
    "bl_attn_rows_times": ([c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_float, c_void_p, c_void_p], ctypes.c_int),
    "bl_attn_transposed_times": ([c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_masked_softmax_dropout_fwd": ([c_void_p, c_int32, c_int32, c_int32, c_void_p, bl_dropout_t, c_void_p, c_void_p], ctypes.c_int),
    "bl_softmax_dropout_bwd": ([c_void_p, c_void_p, c_int32, c_int32, bl_dropout_t, c_void_p], ctypes.c_int),
    "bl_rel_value_bias_fwd": ([c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_rel_value_bias_bwd": ([c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_dropout_inplace": ([c_void_p, c_int64, bl_dropout_t, c_void_p], ctypes.c_int),
    "bl_h3_saturation_events": ([c_int32], c_int64),
    "bl_pack_f16x2": ([c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_amax": ([c_void_p, c_int64, c_void_p, c_void_p], ctypes.c_int),
    "bl_packed_weight_elems_h3": ([c_int32, c_int32, c_int32], c_int64),
    "bl_pack_weights_h3": ([c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p], ctypes.c_int),
    "bl_gemm_rows_h3": ([POINTER(bl_rows_packed_t), c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                         c_int32, c_float, c_void_p, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "bl_gemm_wgrad_h3": ([POINTER(bl_rows_packed_t), c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                          c_int32, c_float, c_void_p, c_void_p, c_int64, c_int32, c_void_p], ctypes.c_int),
    "bl_set_msg_gemm_mode": ([c_int32], c_int32),
    "bl_get_msg_gemm_mode": ([], c_int32),
    "bl_calib_mfma_bf16": ([c_int32, c_int32, c_void_p, POINTER(ctypes.c_double), c_void_p], ctypes.c_int),
    "bl_calib_stream_copy": ([c_void_p, c_void_p, c_int64, c_void_p], ctypes.c_int),
    "bl_prof_enable": ([c_int32], ctypes.c_int),
    "bl_prof_reset": ([], ctypes.c_int),
    "bl_prof_num_kinds": ([], ctypes.c_int),
    "bl_prof_kind_name": ([c_int32], ctypes.c_char_p),
    "bl_prof_read": ([c_int32, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int64), POINTER(c_int32)], ctypes.c_int),
    "bl_prof_read_bytes": ([c_int32], ctypes.c_double),
    "bl_segment_max_fwd": ([c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_segment_max_bwd": ([c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_layernorm_bwd": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_act_bwd": ([c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, bl_dropout_t, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_act_bwd_packed": ([c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, bl_dropout_t, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_mp_scatter_grad": ([c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_gru_scan_saved_elems": ([c_int32, c_int32, c_int32], c_int64),
    "bl_gru_scan_fwd": ([c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_gru_scan_bwd": ([c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_gru_cell_fwd": ([c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, bl_dropout_t, c_void_p, c_void_p], ctypes.c_int),
    "bl_gru_cell_bwd": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, bl_dropout_t, c_void_p, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_segment_log_softmax_fwd": ([c_void_p, c_void_p, c_void_p, c_int32, c_float, c_void_p, c_void_p], ctypes.c_int),
    "bl_segment_log_softmax_bwd": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_rowdot_fwd": ([c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p], ctypes.c_int),
    "bl_rowdot_bwd": ([c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_scatter_add_rows": ([c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "bl_gather_rows": ([c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p], ctypes.c_int),
    "bl_sqnorm_scratch_bytes": ([], ctypes.c_int64),
    "bl_sqnorm": ([c_void_p, c_int64, c_void_p, c_void_p, c_void_p], ctypes.c_int),
    "bl_adam_clip_step_dp": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_float, c_int32, c_void_p], ctypes.c_int),
    "bl_adam_clip_step": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_float, c_float, c_float, c_float, c_float, c_int32, c_void_p], ctypes.c_int),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libbuglab_hip.so and declare every prototype.  Loud failure, never a fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise HipOpsUnavailable(
            f"{p} not found: build it with `make -C neurips21-self-supervised-bug-detection-and-repair_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU fallback for the hot path."
        )
    lib = ctypes.CDLL(p)
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = restype
    if path is None:
        _lib = lib
    return lib


CALL_COUNT = 0  # calls into the library so far (bench.py reports calls per training step)


def _check(rc: int, what: str):
    global CALL_COUNT
    CALL_COUNT += 1
    if rc != 0:
        msg = load_library().bl_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise HipOpsUnavailable(f"{name}: tensor is on {t.device}; the BugLab hot path only runs on a ROCm GPU (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def _f32(t, name="tensor"):
    return _req(t, torch.float32, name)


def _i32(t, name="index"):
    return _req(t, torch.int32, name)


class Dropout(NamedTuple):
    p: float = 0.0
    seed: int = 0
    stream: int = 0

    def c(self) -> bl_dropout_t:
        return bl_dropout_t(float(self.p), int(self.seed) & 0xFFFFFFFF, int(self.stream) & 0xFFFFFFFF)


NO_DROPOUT = Dropout()

RowSource = Tuple[torch.Tensor, Optional[torch.Tensor]]  # (matrix [*, width], row index or None)


def _rows(sources: Sequence[RowSource]) -> Tuple[bl_rows_t, int]:
    r = bl_rows_t()
    assert 1 <= len(sources) <= 3
    K = 0
    for j, (x, idx) in enumerate(sources):
        _f32(x, f"rows source {j}")
        assert x.dim() == 2
        r.x[j] = x.data_ptr()
        r.idx[j] = _i32(idx, f"rows index {j}").data_ptr() if idx is not None else None
        r.ld[j] = x.stride(0)
        r.width[j] = x.shape[1]
        K += x.shape[1]
    r.nsrc = len(sources)
    return r, K


