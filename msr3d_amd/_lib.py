"""ctypes loader for libmsr3d_hip.so -- the only door from Python to the HIP kernels.

There is no CPU fallback anywhere in this package: if the library cannot be
loaded the import fails loudly.  torch is imported first so that the library's
`libamdhip64.so.7` dependency resolves to the HIP runtime torch already loaded
(one runtime per process: torch's streams must be valid handles for our
launches).
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsr3d_hip.so")

_c_int = ctypes.c_int
_c_float = ctypes.c_float
_ptr = ctypes.c_void_p

_fp = ctypes.POINTER(ctypes.c_float)


class StripGemm(ctypes.Structure):
    """msr3d_strip_gemm_t (include/msr3d_hip.h)."""
    _fields_ = ([("M", _c_int), ("N", _c_int), ("pro", _c_int), ("epi", _c_int), ("b_kc", _c_int),
                 ("groups_per_wg", _c_int)]
                + [(k, _ptr) for k in ("a0", "a1", "a2", "st1", "st2", "g1", "b1", "g2", "b2")]
                + [("eps1", _c_float), ("eps2", _c_float), ("p1", _c_float), ("p2", _c_float),
                   ("salt1", ctypes.c_uint), ("salt2", ctypes.c_uint), ("seed", _ptr)]
                + [(k, _ptr) for k in ("o0", "o1", "o2", "ost1", "ost2", "dg1", "db1", "dg2", "db2")]
                + [("W", _ptr), ("ldw", _c_int), ("bias", _ptr), ("C", _ptr), ("ldc", _c_int),
                   ("Cpre", _ptr), ("pre_in", _ptr), ("p_drop", _c_float), ("salt", ctypes.c_uint)])


class GemmProblem(ctypes.Structure):
    """msr3d_gemm_problem_t (include/msr3d_hip.h)."""
    _fields_ = [("a_kc", _c_int), ("b_kc", _c_int), ("M", _c_int), ("N", _c_int), ("K", _c_int),
                ("A", _ptr), ("lda", _c_int), ("B", _ptr), ("ldb", _c_int), ("C", _ptr), ("ldc", _c_int),
                ("bias", _ptr), ("beta", _c_float), ("colsum", _ptr), ("single_run", _c_int)]


class PackJob(ctypes.Structure):
    """msr3d_pack_job_t (include/msr3d_hip.h)."""
    _fields_ = [("src", _ptr), ("ld", _c_int), ("transposed", _c_int), ("rows", _c_int), ("k", _c_int),
                ("nseg", _c_int), ("seg_dst", _c_int * 4), ("seg_len", _c_int * 4), ("seg_src", _c_int * 4),
                ("dst", _ptr)]


class LoraGradJob(ctypes.Structure):
    """msr3d_lora_grad_job_t (include/msr3d_hip.h)."""
    _fields_ = [("C", _c_int), ("P", _ptr), ("ldp", _c_int), ("Q", _ptr), ("ldq", _c_int), ("out", _ptr),
                ("transpose_out", _c_int)]


class LoraShadowJob(ctypes.Structure):
    """msr3d_lora_shadow_job_t (include/msr3d_hip.h)."""
    _fields_ = [("A", _ptr), ("B", _ptr), ("a_pad", _ptr), ("b2", _ptr), ("bt_pad", _ptr), ("at2", _ptr),
                ("r", _c_int), ("K", _c_int), ("N", _c_int), ("pad_", _c_int)]


class SceneBlock(ctypes.Structure):
    """msr3d_scene_block_t (include/msr3d_hip.h)."""
    _fields_ = [("kind", _c_int), ("B", _c_int), ("L", _c_int), ("xp", _ptr), ("a0", _ptr), ("lda0", _c_int),
                ("w1", _ptr), ("w1_bytes", ctypes.c_uint), ("bias1", _ptr), ("w2", _ptr), ("w2_bytes", ctypes.c_uint),
                ("part", _ptr), ("part_stride", ctypes.c_longlong),
                ("pre", _ptr), ("h", _ptr), ("ff", _c_int), ("p_drop", _c_float), ("salt", ctypes.c_uint),
                ("seed", _ptr),
                ("qkvc", _ptr), ("ldq", _c_int), ("dqkvc", _ptr), ("ploc", _ptr), ("pad", _ptr),
                ("probs", _ptr), ("ctx", _ptr), ("H", _c_int),
                ("C", _ptr), ("ldc", _c_int), ("N", _c_int), ("rows_total", _c_int)]


class SceneRows(ctypes.Structure):
    """msr3d_scene_rows_t (include/msr3d_hip.h)."""
    _fields_ = ([("M", _c_int), ("L", _c_int), ("pro", _c_int), ("a0", _ptr), ("part", _ptr), ("nslab", _c_int),
                 ("part_stride", ctypes.c_longlong), ("extra", _ptr), ("a0_bias", _ptr), ("sum_out", _ptr)]
                + [(k, _ptr) for k in ("a1", "a2", "st1", "st2", "g1", "b1", "g2", "b2")]
                + [("eps1", _c_float), ("eps2", _c_float), ("p1", _c_float), ("p2", _c_float),
                   ("salt1", ctypes.c_uint), ("salt2", ctypes.c_uint), ("seed", _ptr)]
                + [(k, _ptr) for k in ("o0", "o1", "o2", "ost1", "ost2", "dg1", "db1", "dg2", "db2", "xp")]
                + [("grad_partials", _c_int)])


class ColsumJob(ctypes.Structure):
    """msr3d_colsum_job_t (include/msr3d_hip.h)."""
    _fields_ = [("part", _ptr), ("dst", _ptr), ("n", _c_int), ("reserved", _c_int)]


class WgradPiece(ctypes.Structure):
    """msr3d_wgrad_piece_t (include/msr3d_hip.h)."""
    _fields_ = [(k, _c_int) for k in ("kind", "prob", "ntile", "ktile", "s0", "s1", "second", "slot")]


class WgradProblem(ctypes.Structure):
    """msr3d_wgrad_problem_t (include/msr3d_hip.h)."""
    _fields_ = [("dy", _ptr), ("ldy", _c_int), ("n_out", _c_int), ("x", _ptr), ("ldx", _c_int), ("k_in", _c_int),
                ("M", _c_int), ("xcd_rot", _c_int), ("dW", _ptr), ("ldw", _c_int), ("db", _ptr)]


BLK = {"attn_fwd": 0, "ffn_fwd": 1, "ffn_bwd": 2, "attn_bwd": 3, "linear": 4, "linear_ksplit": 5}
GEMM_MULTI_MAX = 4
PRO = {"plain": 0, "add": 1, "ln": 2, "ln2": 3, "lnbwd": 4, "ln2bwd": 5}
EPI = {"bias": 0, "gelu": 1, "gelubwd": 2}

# name -> argtypes; every entry point returns int status and ends with the stream.
_SIGNATURES = {
    "msr3d_strip_gemm_f32": [ctypes.POINTER(StripGemm), _ptr],
    "msr3d_split_pack": [_c_int, _ptr, _ptr, _c_int, _ptr],
    "msr3d_rows_linear_split": [_c_int, _c_int, _c_int, _ptr, _c_int, _ptr, ctypes.c_uint, _ptr, _ptr, _c_int, _ptr],
    "msr3d_split_pack_begin": [_c_int, _ptr, _ptr, _c_int, _ptr, ctypes.c_longlong, _ptr, _ptr],
    "msr3d_wgrad_split_colsum": [_c_int, _ptr, _ptr, _c_int, _c_int, _ptr, _ptr],
    "msr3d_scene_block": [ctypes.POINTER(SceneBlock), _ptr],
    "msr3d_scene_rows": [ctypes.POINTER(SceneRows), _ptr],
    "msr3d_wgrad_split": [_c_int, _ptr, _ptr, _c_int, _ptr],
    "msr3d_quant_rows_fp8": [_c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr, _ptr],
    "msr3d_fp8_gemm_lowrank": [_c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _ptr, _c_int, _ptr, _ptr, _c_int, _ptr, _c_int,
                               _ptr, _c_int, _ptr],
    "msr3d_fp8_gemm_lowrank_acc": [_c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _ptr, _c_int, _ptr, _ptr, _c_int, _ptr, _c_int,
                               _ptr, _c_int, _ptr],
    "msr3d_wgrad_split_halves": [_c_int, _ptr, _ptr, _c_int, _ptr, ctypes.c_longlong, _ptr, _ptr],
    "msr3d_wgrad_split_mixed": [_c_int, _ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _ptr, ctypes.c_longlong, _ptr, _ptr],
    "msr3d_wgrad_stream": [_c_int, _ptr, _c_int, _ptr, _ptr, _c_int, _c_int, _ptr, _ptr, ctypes.c_longlong, _ptr, _ptr],
    "msr3d_wgrad_rows_split": [_c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr, _c_int, _c_int, _ptr,
                               ctypes.c_longlong, _ptr, _ptr],
    "msr3d_rows_gemm_split": [_c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _c_int, _ptr, _c_int, _ptr, _ptr,
                              _ptr],
    "msr3d_bn_train_stats": [ctypes.c_longlong, _c_int, _ptr, _c_int, _c_float, _c_float, _ptr, _ptr, _ptr, _ptr, _ptr,
                             _ptr],
    "msr3d_bf16_gemm_lowrank": [_c_int, _c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr, _c_int,
                                _ptr, _c_int, _c_int, _c_float, _ptr],
    "msr3d_bf16_gemm_lowrank_acc": [_c_int, _c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr, _c_int,
                                    _ptr, _c_int, _c_float, _ptr],
    "msr3d_bf16_gemm_batched": [_c_int] * 5 + [_ptr, _c_int, ctypes.c_longlong, ctypes.c_longlong, _ptr, _c_int,
                                ctypes.c_longlong, ctypes.c_longlong, _ptr, _c_int, ctypes.c_longlong,
                                ctypes.c_longlong, _c_int, _c_float, _ptr],
    "msr3d_rmsnorm_fwd": [_c_int, _c_int, _ptr, _ptr, _ptr, _c_float, _ptr, _ptr, _ptr, _ptr],
    "msr3d_rmsnorm_bwd": [_c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_rope_inplace": [_c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _c_int, _ptr],
    "msr3d_rope_inplace2": [_c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _c_int, _ptr],
    "msr3d_causal_softmax_fwd": [_c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr],
    "msr3d_causal_softmax_bwd": [_c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr],
    "msr3d_swiglu_fwd": [ctypes.c_longlong, _ptr, _ptr, _ptr, _ptr],
    "msr3d_swiglu_bwd": [ctypes.c_longlong, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_transpose_bf16": [_c_int, _c_int, _c_int, _c_int, _ptr, _c_int, ctypes.c_longlong, ctypes.c_longlong,
                             _ptr, _c_int, ctypes.c_longlong, ctypes.c_longlong, _ptr],
    "msr3d_bf16_gemm_skinny": [_c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr, _c_int, _c_int, _c_float, _ptr],
    "msr3d_bf16_gemm_skinny_quant": [_c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr, _c_int, _c_int, _c_float, _ptr,
                                     _c_int, _ptr, _ptr],
    "msr3d_colsum_partials": [_c_int, _ptr, _ptr],
    "msr3d_lora_grad": [_c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr, _c_int, _c_float, _c_int, _ptr,
                        ctypes.c_longlong, _ptr],
    "msr3d_lora_grad_pair": [_c_int, _c_int, _c_int, ctypes.POINTER(LoraGradJob), _c_float, _c_int, _ptr],
    "msr3d_lora_shadows": [_c_int, _ptr, _ptr],
    "msr3d_attn_fwd": [_c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _c_int, _ptr, _c_float, _ptr, _ptr, _ptr],
    "msr3d_attn_bwd": [_c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _ptr, _c_float, _ptr, _ptr,
                       _ptr, _ptr, _ptr, _ptr],
    "msr3d_sa_level_split": [_c_int, _c_int, _c_int, _c_int, _c_float] + [_ptr] * 13,
    "msr3d_sa_level2_rows": [_c_int, _c_int, _c_int, _c_float] + [_ptr] * 14 + [_c_int, _ptr],
    "msr3d_sa_level2_rows_ws_bytes": [_c_int],
    "msr3d_sa_level1_rows": [_c_int, _c_int, _c_int] + [_ptr] * 13 + [_c_int, _ptr],
    "msr3d_sa_level3_tiles": [_c_int] + [_ptr] * 12,
    "msr3d_sa_fps2_query_plan": [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_float, _c_int,
                                 _ptr, _ptr, _ptr, _c_float, _ptr, _ptr, _ptr, _ptr],
    "msr3d_sa_plan12": [_c_int, _c_int, _ptr, _ptr, _c_int, _c_int, _c_float] + [_ptr] * 8,
    "msr3d_sa_level1_rows_ws_bytes": [_c_int, _c_int],
    "msr3d_seq_ce_fwd": [_c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_seq_ce_bwd": [_c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_gemm_multi_f32": [_c_int, ctypes.POINTER(GemmProblem), _ptr],
    "msr3d_scene_prologue": [_c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_float, _ptr, _ptr,
                             _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_scene_prologue_agent": [_c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_float] + [_ptr] * 8 + [_ptr],
    "msr3d_step_begin": [_ptr, ctypes.c_longlong, _ptr, _ptr],
    "msr3d_pos_embed_fwd": [_c_int, _c_int] + [_ptr] * 6 + [_c_float] + [_ptr] * 4 + [_c_float] + [_ptr] * 5 + [_ptr],
    "msr3d_pos_embed_tokens_fwd": [_c_int, _c_int, _c_int] + [_ptr] * 6 + [_c_float] + [_ptr] * 4 + [_c_float] + [_ptr] * 10 + [_ptr],
    "msr3d_pos_embed_bwd": [_c_int] + [_ptr] * 17 + [_ptr],
    "msr3d_anchor_front_fwd": [_c_int, _c_int] + [_ptr] * 10 + [_c_float] + [_ptr] * 5 + [_ptr],
    "msr3d_anchor_front_bwd": [_c_int, _c_int] + [_ptr] * 14 + [_ptr],
    "msr3d_furthest_point_sampling": [_c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr],
    "msr3d_gather_points": [_c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr],
    "msr3d_gather_points_grad": [_c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr],
    "msr3d_ball_query": [_c_int, _c_int, _c_int, _c_float, _c_int, _ptr, _ptr, _ptr, _ptr],
    "msr3d_group_points": [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr],
    "msr3d_group_points_grad": [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr],
    "msr3d_three_nn": [_c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_three_interpolate": [_c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_three_interpolate_grad": [_c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_gemm_f32": [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr, _c_int,
                       _ptr, _ptr, _c_int, _c_float, _c_float, _ptr, ctypes.c_uint, _ptr, ctypes.c_size_t, _ptr],
    "msr3d_linear_wgrad_f32": [_c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, ctypes.c_size_t, _ptr],
    "msr3d_linear_wgrad_acc_f32": [_c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, ctypes.c_size_t, _ptr],
    "msr3d_linear_bwd_f32": [_c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _c_float, _ptr, _ptr, _ptr,
                             ctypes.c_size_t, _ptr],
    "msr3d_colsum_f32": [_c_int, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr],
    "msr3d_gelu_bwd_f32": [ctypes.c_longlong, _ptr, _ptr, _ptr, _c_float, _ptr, ctypes.c_uint, _ptr],
    "msr3d_spatial_attn_fwd": [_c_int] * 5 + [_ptr, _ptr, _ptr, _c_int, _ptr, _c_int, _ptr, _ptr, _ptr,
                                              _ptr, _c_int, _ptr],
    "msr3d_spatial_attn_bwd": [_c_int] * 5 + [_ptr, _ptr, _ptr, _c_int, _ptr, _c_int, _ptr, _ptr, _ptr,
                                              _ptr, _ptr, _ptr, _ptr, _c_int, _ptr, _c_int, _c_int, _ptr],
    "msr3d_group_rows": [_c_int] * 6 + [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_group_rows_grad": [_c_int] * 6 + [_ptr, _ptr, _ptr, _ptr],
    "msr3d_bn_relu_train_fwd": [ctypes.c_longlong, _c_int, _ptr, _ptr, _ptr, _c_float, _c_float, _ptr, _ptr, _ptr,
                                _ptr, _ptr, _ptr, _c_int, _ptr],
    "msr3d_bn_relu_train_bwd": [ctypes.c_longlong, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                _ptr, _ptr, _ptr, _ptr],
    "msr3d_bn_relu_maxpool_train_fwd": [ctypes.c_longlong, _c_int, _c_int, _ptr, _ptr, _ptr, _c_float, _c_float,
                                        _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _ptr],
    "msr3d_bn_relu_maxpool_train_bwd": [ctypes.c_longlong, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                        _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_pairwise_locs": [_c_int, _c_int, _ptr, _c_int, _c_float, _ptr, _ptr],
    "msr3d_agent_fourier": [_c_int, _c_int, _ptr, _c_int, _ptr, _ptr, _ptr, _c_int, _c_int, _ptr, _ptr],
    "msr3d_add_row_vectors": [_c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_dropout_add_ln_fwd": [_c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _c_float, _c_float, _ptr,
                                 ctypes.c_uint, _ptr, _ptr, _ptr, _ptr],
    "msr3d_dropout_add_ln_bwd": [_c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _c_float, _ptr, ctypes.c_uint,
                                 _ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _ptr],
    "msr3d_dropout_add_ln2_fwd": [_c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _c_float, _c_float, ctypes.c_uint,
                                  _ptr, _ptr, _c_float, _c_float, ctypes.c_uint, _ptr, _ptr, _ptr, _ptr,
                                  _ptr, _ptr, _ptr],
    "msr3d_dropout_add_ln2_bwd": [_c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _c_float, ctypes.c_uint, _ptr, _ptr,
                                  _ptr, _c_float, ctypes.c_uint, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                  _ptr, _ptr],
    "msr3d_bump_seed": [_ptr, _ptr],
    "msr3d_scene_scatter": [_c_int, _c_int, _c_int, _c_int, _ptr, ctypes.c_longlong, _ptr, _ptr, _c_int,
                            _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_project_scatter_bf16": [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr, ctypes.c_longlong, _ptr, _ptr,
                                   _ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_set_reserved_cus": [_c_int],
    "msr3d_dot_f32": [ctypes.c_longlong, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_adamw_flat": [ctypes.c_longlong, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_float, _c_float,
                         _c_float, _c_float, _c_float, _c_float, _c_int, _c_int, _c_int, _c_int, _ptr],
    "msr3d_adamw_flat_masked": [ctypes.c_longlong, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_float, _c_float,
                                _c_float, _c_float, _c_float, _c_float, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr],
    "msr3d_adamw_flat_scaled": [ctypes.c_longlong, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_float, _c_float,
                                _c_float, _c_float, _c_float, _c_float, _c_int, _c_int, _c_int, _c_int, _ptr, _c_float,
                                _ptr],
    "msr3d_sa_fps2": [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_sa_fps2_query": [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_float, _c_int, _ptr,
                            _ptr],
    "msr3d_sa_fps2_flags": [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_sa_fps2_query_flags": [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_float, _c_int,
                                  _ptr, _ptr, _ptr],
    "msr3d_sa_level": [_c_int, _c_int, _c_int, _c_int, _c_float, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                       _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_segment_scan": [_c_int, _ptr, _ptr, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "msr3d_preprocess_pcd": [_c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, ctypes.c_ulonglong,
                             _ptr, _ptr, _ptr, _ptr, _ptr],
}

_lib = None


def exported_symbols():
    """Every symbol include/msr3d_hip.h declares (checked by the CPU test-suite)."""
    return ["msr3d_abi_version", "msr3d_status_string", "msr3d_sqdist_contract", "msr3d_wgrad_form", "msr3d_attn_fwd_form"] + list(_SIGNATURES)


ABI_VERSION = 29        # MSR3D_ABI_VERSION of include/msr3d_hip.h these signatures were written for


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # the in-tree build is part of the product; build it rather than limp along
        from . import build as _build
        _build.build()
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise ImportError(
            f"msr3d_amd: cannot load {LIB_PATH} ({e}). Build it with `python -m msr3d_amd.build`; "
            "there is no CPU fallback.") from e
    lib.msr3d_abi_version.restype = _c_int
    if lib.msr3d_abi_version() != ABI_VERSION:
        raise ImportError(
            f"msr3d_amd: {LIB_PATH} has ABI version {lib.msr3d_abi_version()}, this package binds "
            f"version {ABI_VERSION} (include/msr3d_hip.h). Rebuild it with `python -m msr3d_amd.build`.")
    lib.msr3d_sqdist_contract.restype = _c_int
    lib.msr3d_wgrad_form.restype = _c_int
    lib.msr3d_wgrad_form.argtypes = [_c_int]
    lib.msr3d_attn_fwd_form.restype = _c_int
    lib.msr3d_attn_fwd_form.argtypes = [_c_int]
    lib.msr3d_status_string.restype = ctypes.c_char_p
    lib.msr3d_status_string.argtypes = [_c_int]
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_size_t if name.endswith("_ws_bytes") else _c_int
        if argtypes and argtypes[-1] is _ptr and not name.endswith("_ws_bytes"):
            setattr(lib, name, _Entry(name, fn))      # (a launch: can be bracketed by events, see set_timing_sink)
    _lib = lib
    return lib


_BLK_NAMES = {0: "attn_fwd", 1: "ffn_fwd", 2: "ffn_bwd", 3: "attn_bwd", 4: "linear", 5: "linear_ksplit"}
_PRO_NAMES = {0: "plain", 1: "add", 2: "ln", 3: "ln2", 4: "lnbwd", 5: "ln2bwd"}


class _Entry:
    """A launching entry point of the library.  Transparent (one attribute test per call) unless bench.py's kernel census
    is on (`set_timing_sink(sink, census=True)`): then EVERY launch is bracketed by HIP events on torch's current stream
    -- the stream the callers launch on -- and recorded under the entry's name; msr3d_scene_block / msr3d_scene_rows
    are keyed by the struct's kind / prologue as well (their kernels differ)."""
    __slots__ = ("name", "fn")

    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *a):
        if not _timing_census or _timing_sink is None:
            return self.fn(*a)
        key = self.name
        if key == "msr3d_scene_block":
            key += "[" + _BLK_NAMES.get(a[0]._obj.kind, "?") + "]"
        elif key == "msr3d_scene_rows":
            key += "[" + _PRO_NAMES.get(a[0]._obj.pro, "?") + "]"
        elif key in ("msr3d_sa_level_split", "msr3d_sa_level"):
            key += f"[{int(a[0])}]"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = self.fn(*a)
        e1.record()
        _timing_sink.setdefault(key, []).append((e0, e1))
        return rc


_lib_split2 = None
_SPLIT2_ENTRIES = ("msr3d_sa_level_split", "msr3d_sa_level1_rows", "msr3d_sa_level2_rows", "msr3d_sa_level1_rows_ws_bytes",
                   "msr3d_sa_level2_rows_ws_bytes", "msr3d_sa_plan12")


def load_split2():
    """libmsr3d_hip_split2.so: csrc/sa_split.hip compiled with two bf16 terms per operand and three products per product
    (msr3d_amd/build.py) -- the labelled variant MSR3D_SA_MMA=split2 / fused.set_sa_mma("split2").  Same entry names and
    signatures as the main library's set-abstraction entries; loaded only on request."""
    global _lib_split2
    if _lib_split2 is None:
        load()                                       # (builds both libraries if needed; the HIP runtime is torch's)
        from . import build as _build
        if not os.path.exists(_build.LIB_SPLIT2):
            _build.build()
        lib = ctypes.CDLL(_build.LIB_SPLIT2)
        ver = getattr(lib, "msr3d_abi_version", None)
        if ver is not None:
            ver.restype = _c_int
        if ver is None or ver() != ABI_VERSION:
            # built from an older header: the entry points' arguments may have moved
            raise ImportError(
                f"msr3d_amd: {_build.LIB_SPLIT2} has ABI version {ver() if ver is not None else 'none (pre-v27 build)'}, this "
                f"package binds version {ABI_VERSION} (include/msr3d_hip.h). Rebuild it with `python -m msr3d_amd.build`.")
        lib.msr3d_set_reserved_cus.argtypes = [_c_int]
        lib.msr3d_set_reserved_cus.restype = _c_int
        if _reserved_cus[0]:                         # (its kernels size their grids from their OWN copy of the setting)
            check(lib.msr3d_set_reserved_cus(_reserved_cus[0]), "msr3d_set_reserved_cus (split2)")
        for name in _SPLIT2_ENTRIES:
            fn = getattr(lib, name)
            fn.argtypes = _SIGNATURES[name]
            fn.restype = ctypes.c_size_t if name.endswith("_ws_bytes") else _c_int
        _lib_split2 = lib
    return _lib_split2


_lib_bf16 = None
_BF16_ENTRIES = ("msr3d_scene_block", "msr3d_wgrad_split", "msr3d_wgrad_split_colsum", "msr3d_wgrad_split_halves",
                 "msr3d_wgrad_split_mixed", "msr3d_wgrad_stream")


def load_bf16():
    """libmsr3d_hip_bf16.so: csrc/scene_block.hip + csrc/wgrad_split.hip compiled with MSR3D_TRAIN_PLANES=1 -- the
    LABELLED reduced variant MSR3D_TRAIN_MMA=bf16 / scene_blocks.set_train_mma("bf16") of the trainable part: one bf16
    MFMA product of bf16-rounded operands per product instead of six, fp32 accumulate.  Same entry names, signatures and
    buffer layouts as the main library's; loaded only on request."""
    global _lib_bf16
    if _lib_bf16 is None:
        load()
        from . import build as _build
        if not os.path.exists(_build.LIB_BF16):
            _build.build()
        lib = ctypes.CDLL(_build.LIB_BF16)
        lib.msr3d_abi_version.restype = _c_int
        if lib.msr3d_abi_version() != ABI_VERSION:
            raise ImportError(
                f"msr3d_amd: {_build.LIB_BF16} has ABI version {lib.msr3d_abi_version()}, this package binds version "
                f"{ABI_VERSION} (include/msr3d_hip.h). Rebuild it with `python -m msr3d_amd.build`.")
        lib.msr3d_wgrad_form.restype = _c_int
        lib.msr3d_wgrad_form.argtypes = [_c_int]
        lib.msr3d_attn_fwd_form.restype = _c_int
        lib.msr3d_attn_fwd_form.argtypes = [_c_int]
        for name in _BF16_ENTRIES:
            fn = getattr(lib, name)
            fn.argtypes = _SIGNATURES[name]
            fn.restype = _c_int
            setattr(lib, name, _Entry(name, fn))
        _lib_bf16 = lib
    return _lib_bf16


_reserved_cus = [0]


def set_reserved_cus(n):
    """msr3d_set_reserved_cus on EVERY loaded library: the split2 variant library has its own copy of the setting (and
    of usable_cus()), and takes the current value when it is loaded later."""
    n = int(n)
    check(load().msr3d_set_reserved_cus(n), "msr3d_set_reserved_cus")
    _reserved_cus[0] = n
    if _lib_split2 is not None:
        check(_lib_split2.msr3d_set_reserved_cus(n), "msr3d_set_reserved_cus (split2)")


def check(status, what):
    if status != 0:
        msg = load().msr3d_status_string(status).decode()
        raise RuntimeError(f"{what} failed: {msg} (status {status})")


def current_stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# ---------------------------------------------------------------------------------------
# Per-launch timing with HIP events on the launching stream (bench.py's roofline leg).
# Disabled unless bench.py installs a sink: {entry_point_name: [(start_event, end_event)]}.
# ---------------------------------------------------------------------------------------
_timing_sink = None
_timing_every = 1
_timing_calls = {}
_timing_census = False


def set_timing_sink(sink, every=1, census=False):
    """every = k: only every k-th launch of a name is bracketed by events.  An event pair costs the GPU ~6 us of idle
    time on either side of the launch (a barrier packet each): timing EVERY launch of the dominant kernel put 12 us --
    1 % -- of measurement overhead into each timed step."""
    global _timing_sink, _timing_every, _timing_census
    _timing_sink = sink
    _timing_every = max(1, int(every))
    _timing_census = bool(census) and sink is not None
    _timing_calls.clear()


class kernel_timer:
    __slots__ = ("rec", "e0")

    def __init__(self, name):
        self.rec = _timing_sink.get(name) if (_timing_sink is not None and not _timing_census) else None
        if self.rec is not None and _timing_every > 1:
            c = _timing_calls.get(name, 0)
            _timing_calls[name] = c + 1
            if c % _timing_every:
                self.rec = None

    def __enter__(self):
        if self.rec is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()          # torch's current stream == the stream we launch on
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.rec.append((self.e0, e1))
        return False
