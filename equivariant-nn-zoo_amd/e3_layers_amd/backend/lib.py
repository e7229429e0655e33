"""ctypes binding of ``csrc/libe3k.so`` (the C ABI declared in ``include/e3k.h``).

The library is built in-tree by ``make -C equivariant-nn-zoo_amd/csrc`` (driven by
``__graft_entry__.build()``).  There is **no fallback**: if the shared object is missing or a
call returns a non-zero status a ``RuntimeError`` is raised — the product path never routes
through a CPU or eager-PyTorch implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from .tuning import knob as _knob

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.normpath(os.path.join(_HERE, "..", "..", "csrc"))
LIB_PATH = _knob("E3K_LIB") or os.path.join(CSRC_DIR, "libe3k.so")  # E3K_LIB: debug builds only

TP_MAXQ = 8


class GemmProblem(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("A2", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
        ("row_index", C.c_void_p), ("group_dev", C.c_void_p),
        ("M1", C.c_int32), ("M2", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("V", C.c_int32), ("accumulate", C.c_int32),
        ("a_r1", C.c_int64), ("a_r2", C.c_int64), ("a_k", C.c_int64),
        ("a2_r1", C.c_int64),
        ("b_k", C.c_int64), ("b_n", C.c_int64),
        ("c_r1", C.c_int64), ("c_r2", C.c_int64), ("c_n", C.c_int64),
        ("alpha", C.c_float), ("act", C.c_int32), ("act_cst", C.c_float), ("chain", C.c_int32),
    ]


class GemmSegment(C.Structure):
    _fields_ = [
        ("templates", C.POINTER(GemmProblem)), ("n_templates", C.c_int32), ("n_keys", C.c_int32),
        ("a_base", C.c_void_p), ("a2_base", C.c_void_p), ("b_base", C.c_void_p), ("c_base", C.c_void_p),
        ("bias_base", C.c_void_p), ("M1", C.c_int64), ("perm", C.c_void_p), ("groups_dev", C.c_void_p),
        ("b_key_stride", C.c_int64),
    ]


class TpGroup(C.Structure):
    _fields_ = [
        ("l1", C.c_int32), ("x_off", C.c_int32), ("mul", C.c_int32), ("mask", C.c_uint32),
        ("y_off", C.c_int32 * 4),
        ("w_off", C.c_int32 * TP_MAXQ), ("out_off", C.c_int32 * TP_MAXQ), ("out_stride", C.c_int32 * TP_MAXQ),
        ("coeff", C.c_float * TP_MAXQ),
    ]


class KwInstr(C.Structure):
    _fields_ = [("w_off", C.c_int64), ("m_off", C.c_int64), ("u", C.c_int32), ("w_out", C.c_int32)]


class Block(C.Structure):
    _fields_ = [("off", C.c_int32), ("mul", C.c_int32), ("dim", C.c_int32), ("_pad", C.c_int32)]


class GateSeg(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("in_off", C.c_int32), ("gate_off", C.c_int32), ("out_off", C.c_int32),
        ("mul", C.c_int32), ("dim", C.c_int32), ("act", C.c_int32), ("cst", C.c_float),
    ]


class GemmSet(C.Structure):
    _fields_ = [("p", C.POINTER(GemmProblem)), ("n", C.c_int32), ("n_rounds", C.c_int32), ("round_start", C.c_int32 * 5),
                ("_pad", C.c_int32)]


class LayerDesc(C.Structure):
    _fields_ = ([("tp", C.c_void_p)]
                + [(k, GemmSet) for k in ("lin1_fwd", "lin1_dgrad", "lin1_dgrad_acc", "lin1_wgrad", "post_fwd", "post_dgrad",
                                          "post_wgrad", "sc_fwd", "sc_dgrad", "sc_wgrad", "last_fwd", "last_dgrad", "last_wgrad")]
                + [("gate", C.POINTER(GateSeg)), ("n_gate", C.c_int32), ("n_in_blocks", C.c_int32), ("in_blocks", C.POINTER(Block)),
                   ("kw", C.POINTER(KwInstr)), ("n_kw", C.c_int32), ("V", C.c_int32), ("ld_m", C.c_int64),
                   ("k0", C.c_int32), ("h", C.c_int32), ("n_hidden", C.c_int32), ("act", C.c_int32), ("cst", C.c_float),
                   ("alphas", C.c_float * 4)]
                + [(k, C.c_int32) for k in ("d_in", "d_x1", "d_mid", "d_conv", "d_out", "W", "post_in_covered", "sc_in_covered",
                                            "lin1_in_covered", "sc_out_covered", "post_out_covered", "tp_bwd_x_overwrites")])


class LayerRadial(C.Structure):
    _fields_ = [("R", C.c_int64), ("E", C.c_int64), ("use_table", C.c_int32), ("keep", C.c_int32), ("knots", C.c_int32),
                ("have_rows", C.c_int32), ("in_kernel", C.c_int32), ("packed", C.c_int32), ("radial", C.c_void_p), ("bin", C.c_void_p), ("bin_ptr", C.c_void_p), ("bin_perm", C.c_void_p),
                ("bin_coef", C.c_void_p), ("bin_seg", C.c_void_p), ("w_last", C.c_void_p), ("w_hidden", C.c_void_p * 4), ("h", C.c_void_p),
                ("z", C.c_void_p * 4), ("T", C.c_void_p), ("w", C.c_void_p), ("erec_dst", C.c_void_p), ("erec_src", C.c_void_p), ("P", C.c_void_p)]


class RadialStackItem(C.Structure):
    _fields_ = [("rad", LayerRadial), ("g_rows", C.c_void_p), ("gb_last", C.c_void_p), ("gb_hidden", C.c_void_p * 4),
                ("g_h", C.c_void_p), ("g_radial", C.c_void_p), ("g_slope", C.c_void_p), ("hp", C.c_void_p), ("g_hp", C.c_void_p)]


class SlopeCtx(C.Structure):
    _fields_ = [("knots", C.c_void_p), ("bessel_w", C.c_void_p), ("r_max", C.c_float), ("r_min", C.c_float), ("p", C.c_float),
                ("one_over_r", C.c_int32), ("cutoff_kind", C.c_int32), ("_pad", C.c_int32), ("acc", C.c_void_p), ("g_bessel", C.c_void_p)]


class KwMultiItem(C.Structure):
    _fields_ = [("args", C.c_void_p), ("W", C.c_void_p), ("M", C.c_void_p), ("g_W", C.c_void_p), ("accumulate_w", C.c_int32),
                ("_pad", C.c_int32)]


class KwStackItem(C.Structure):
    _fields_ = [("w_sc", C.c_void_p), ("m", C.c_void_p), ("gb_sc", C.c_void_p), ("acc_sc", C.c_int32), ("_pad", C.c_int32)]


class MlpNet(C.Structure):
    _fields_ = [("weights", C.c_void_p * 4), ("z", C.c_void_p * 4), ("out", C.c_void_p), ("g_out", C.c_void_p),
                ("g_weights", C.c_void_p * 4), ("g_x", C.c_void_p)]


class LayerFwdArgs(C.Structure):
    _fields_ = ([("N", C.c_int64), ("E", C.c_int64)]
                + [(k, C.c_int32) for k in ("in_cf", "out_cf", "keep", "fork", "has_w", "n_keys", "have_m", "_pad")]
                + [(k, C.c_void_p) for k in ("main", "side", "side2", "x", "node_attrs", "sh", "src", "dst_ptr", "dst_perm", "perm",
                                             "bounds", "reps", "w_lin1", "w_post", "w_sc")]
                + [("rad", LayerRadial), ("next", C.c_void_p), ("next_rad", C.POINTER(LayerRadial))]
                + [(k, C.c_void_p) for k in ("x_cf", "a_rep", "m", "conv", "x1", "mid", "y")])


class LayerBwdArgs(C.Structure):
    _fields_ = ([("N", C.c_int64), ("E", C.c_int64)]
                + [(k, C.c_int32) for k in ("in_cf", "out_cf", "fork", "n_keys", "need_x", "need_attrs", "need_radial", "acc_sc",
                                            "have_m", "fuse_xw")]
                + [(k, C.c_void_p) for k in ("main", "side", "side2", "side3", "x_cf", "sh", "x1", "mid", "conv", "a_rep", "m",
                                             "src", "dst", "dst_ptr", "dst_perm", "src_ptr", "src_perm", "perm", "bounds", "reps",
                                             "w_lin1", "w_post", "w_sc")]
                + [("rad", LayerRadial), ("gy", C.c_void_p)]
                + [(k, C.c_void_p) for k in ("gb_lin1", "gb_post", "gb_sc", "gb_last")]
                + [("gb_hidden", C.c_void_p * 4)]
                + [(k, C.c_void_p) for k in ("g_x", "g_attrs", "g_radial", "g_conv", "g_mid", "g_x1", "g_xcf", "g_w", "g_T",
                                             "table_ws", "g_h", "gm", "ga", "kw_ws")])


# name -> (restype, argtypes); every symbol include/e3k.h declares must appear here (tests check it)
_P, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SIGNATURES = {
    "e3k_strerror": (C.c_char_p, [C.c_int]),
    "e3k_version": (C.c_int, []),
    "e3k_tp_limits": (None, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "e3k_gemm": (C.c_int, [C.POINTER(GemmProblem), C.c_int, _P]),
    "e3k_gemm_wgrad": (C.c_int, [C.POINTER(GemmProblem), C.c_int, _P]),
    "e3k_gemm_rebased": (C.c_int, [C.POINTER(GemmProblem), C.c_int, _P, _P, _P, _P, _P, _I64, _I32, _P]),
    "e3k_gemm_grouped_rebased": (C.c_int, [C.POINTER(GemmProblem), C.c_int, _P, _P, _P, _I64, _P, _P, _I32, _I64, _I32, _P]),
    "e3k_gemm_grouped": (C.c_int, [C.POINTER(GemmProblem), C.c_int, _P, _P, _I32, _I64, _I32, _P]),
    "e3k_gemm_multi": (C.c_int, [C.POINTER(GemmSegment), _I32, _I32, _P]),
    "e3k_layer_create": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(C.c_void_p)]),
    "e3k_layer_destroy": (None, [_P]),
    "e3k_layer_fwd": (C.c_int, [_P, C.POINTER(LayerFwdArgs)]),
    "e3k_layer_bwd": (C.c_int, [_P, C.POINTER(LayerBwdArgs)]),
    "e3k_radial_stack_fwd": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(LayerRadial), _I32, _P]),
    "e3k_radial_stack_bwd": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(RadialStackItem), _I32, C.POINTER(SlopeCtx), _P]),
    "e3k_radial_slope_fwd": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(LayerRadial), _I32, C.POINTER(SlopeCtx), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p), _P]),
    "e3k_slope_tangent_fwd": (C.c_int, [C.POINTER(C.c_void_p), _I32, _I32, C.POINTER(C.c_float), _P, _I64, _P, _I32, _I32, _F, _F, _F,
                                        _I32, _I32, _I32, _F, C.POINTER(C.c_void_p), _P]),
    "e3k_slope_tangent_bwd_scratch": (C.c_int64, [_I32, _I32, _I32, _I32, _I64]),
    "e3k_slope_tangent_bwd": (C.c_int, [C.POINTER(C.c_void_p), _I32, _I32, C.POINTER(C.c_float), _P, _I64, _P, _I32, _I32, _F, _F, _F,
                                        _I32, _I32, _I32, _F, C.POINTER(C.c_void_p), _P, C.POINTER(C.c_void_p), _P, _P]),
    "e3k_mlp_hidden_fwd_multi": (C.c_int, [C.POINTER(MlpNet), _I32, _P, _I64, _I32, _I32, _I32, C.POINTER(C.c_float), _I32, _F, _P]),
    "e3k_mlp_hidden_bwd_multi": (C.c_int, [C.POINTER(MlpNet), _I32, _P, _I64, _I32, _I32, _I32, C.POINTER(C.c_float), _I32, _F, _P]),
    "e3k_kw_args_create": (C.c_int, [_P, _I32, _I32, _I64, C.POINTER(C.c_void_p)]),
    "e3k_kw_args_destroy": (None, [_P]),
    "e3k_keyed_weights_fwd_multi": (C.c_int, [C.POINTER(KwMultiItem), _I32, _P, _I32, _P]),
    "e3k_keyed_weights_bwd_multi_workspace": (C.c_int64, [C.POINTER(KwMultiItem), _I32, _I32]),
    "e3k_keyed_weights_bwd_multi": (C.c_int, [C.POINTER(KwMultiItem), _I32, _P, _I32, _P, _P, _P]),
    "e3k_kw_stack_fwd": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(KwStackItem), _I32, _P, _P, _I32, _P, _P]),
    "e3k_kw_stack_bwd_workspace": (C.c_int64, [C.POINTER(C.c_void_p), _I32, _I32]),
    "e3k_kw_stack_bwd": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(KwStackItem), _I32, _P, _P, _P, _I64, _I32, _P, _P, _P, _P]),
    "e3k_layer_profile": (C.c_int, [_P, _I32]),
    "e3k_layer_profile_mask": (C.c_int, [_P, _I32, C.c_uint32]),
    "e3k_layer_profile_read": (C.c_int, [_P, _I32, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _I32]),
    "e3k_colsum": (C.c_int, [_P, _I64, _I32, _I64, _P, _P]),
    "e3k_counts_to_ptr": (C.c_int, [_P, _I32, _P, _P]),
    "e3k_sq_error": (C.c_int, [_P, _P, _P, _I32, _I64, C.c_float, _P, _P, _P]),
    "e3k_onehot": (C.c_int, [_P, _I64, _I32, _P, _P, _P]),
    "e3k_flag_fetch_clear": (C.c_int, [_P, _P, _P]),
    "e3k_fctp_reduce_bwd": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _P, _I32, _P, _P]),
    "e3k_edge_vector_fwd": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P]),
    "e3k_edge_vector_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P]),
    "e3k_sph_harm_fwd": (C.c_int, [_P, _I64, C.POINTER(_I32), _I32, _I32, _I32, _P, _P]),
    "e3k_sph_harm_bwd": (C.c_int, [_P, _P, _I64, C.POINTER(_I32), _I32, _I32, _I32, _P, _P]),
    "e3k_sph_harm_bwd2": (C.c_int, [_P, _P, _P, _I64, C.POINTER(_I32), _I32, _I32, _I32, _P, _P, _P]),
    "e3k_radial_basis_fwd": (C.c_int, [_P, _I64, _P, _I32, _F, _F, _F, _I32, _I32, _P, _P]),
    "e3k_radial_basis_bwd": (C.c_int, [_P, _P, _I64, _P, _I32, _F, _F, _F, _I32, _I32, _P, _P, _P]),
    "e3k_radial_basis_bwd2": (C.c_int, [_P, _P, _P, _P, _I64, _P, _I32, _F, _F, _F, _I32, _I32, _P, _P, _P, _P]),
    "e3k_keyed_weights_fwd": (C.c_int, [_P, _P, C.POINTER(KwInstr), _I32, _I32, _I32, _I64, _P, _P]),
    "e3k_keyed_weights_bwd_workspace": (C.c_int64, [C.POINTER(KwInstr), _I32, _I32, _I32]),
    "e3k_keyed_weights_bwd": (C.c_int, [_P, _P, _P, C.POINTER(KwInstr), _I32, _I32, _I32, _I64, _P, _P, _I32, _P, _P]),
    "e3k_mlp_hidden_fwd": (C.c_int, [_P, _I64, _I32, _I32, _I32, C.POINTER(_P), C.POINTER(_F), _I32, _F, C.POINTER(_P), _P, _P]),
    "e3k_mlp_hidden_bwd": (C.c_int, [_P, _I64, _I32, _I32, _I32, C.POINTER(_P), C.POINTER(_F), _I32, _F, C.POINTER(_P), _P,
                                     C.POINTER(_P), _P, _P]),
    "e3k_radius_graph_count": (C.c_int, [_P, _P, _P, _I64, _F, _P, _P, _P, _P]),
    "e3k_radius_graph_fill": (C.c_int, [_P, _P, _P, _I64, _F, _P, _P, _P, _I64, _P, _P]),
    "e3k_tp_plan_create": (C.c_int, [C.POINTER(TpGroup), _I32, _I32, _I32, _I32, _I32, C.POINTER(_P)]),
    "e3k_tp_plan_destroy": (None, [_P]),
    "e3k_tp_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P]),
    "e3k_tp_bwd_w": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "e3k_tp_bwd_x_overwrites": (C.c_int, [_P]),
    "e3k_tp_table_supported": (C.c_int, [_P]),
    "e3k_tp_fwd_table": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P]),
    "e3k_tp_bwd_x_table": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P]),
    "e3k_tp_fwd_ptable": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _P, _P]),
    "e3k_tp_bwd_x_ptable": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _P, _P]),
    "e3k_tp_bwd_xw_ptable": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "e3k_tp_bwd_xw": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "e3k_tp_bwd_xe": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "e3k_tp_edge_partials_floats": (C.c_int64, [_P, _I64]),
    "e3k_tp_bwd_xw_dual": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P]),
    "e3k_edge_records": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I64, _P, _P]),
    "e3k_tp_table2_supported": (C.c_int, [_P]),
    "e3k_tp_second_order_streamed_supported": (C.c_int, [_P]),
    "e3k_tp_bwd_e_table": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P]),
    "e3k_tp_fwd_jvp_table": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P]),
    "e3k_tp_bwd_x_dual_table": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P]),
    "e3k_tp_bwd_w_dual": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P]),
    "e3k_tp_bwd_x": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P]),
    "e3k_csr_workspace_ints": (C.c_int64, [_I64, _I64]),
    "e3k_csr_build": (C.c_int, [_P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "e3k_group_rows": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P]),
    "e3k_rtable_bins_workspace_ints": (C.c_int64, [_I64, _I32]),
    "e3k_rtable_bins": (C.c_int, [_P, _I64, _F, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "e3k_rtable_bins_keyed": (C.c_int, [_P, _P, _I32, _I64, _F, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "e3k_rtable_interp_fwd": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "e3k_rtable_interp_fwd2": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _P]),
    "e3k_rtable_bwd_workspace_floats": (C.c_int64, [_I64, _I32, _I32]),
    "e3k_rtable_interp_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _I32, _P]),
    "e3k_rtable_pack": (C.c_int, [_P, _I32, _I32, _P, _P]),
    "e3k_rtable_pack_multi": (C.c_int, [_P, _I32, _P, _P, _I32, _P]),
    "e3k_rtable_interp_packed": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "e3k_rtable_guard": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _F, _F, _I32, _P]),
    "e3k_act_fwd": (C.c_int, [_P, _I64, _I32, _F, _P, _P]),
    "e3k_act_bwd": (C.c_int, [_P, _P, _I64, _I32, _F, _P, _P]),
    "e3k_act_bwd_from_output": (C.c_int, [_P, _P, _I64, _I32, _F, _P, _P]),
    "e3k_act_bwd2": (C.c_int, [_P, _P, _P, _I64, _I32, _F, _P, _P, _P]),
    "e3k_relayout": (C.c_int, [_P, _I64, _I32, C.POINTER(Block), _I32, _I32, _P, _P]),
    "e3k_gate_fwd": (C.c_int, [_P, _I64, _I32, _I32, C.POINTER(GateSeg), _I32, _I32, _P, _P]),
    "e3k_gate_bwd": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, C.POINTER(GateSeg), _I32, _I32, _P, _P]),
    "e3k_gate_bwd2": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, C.POINTER(GateSeg), _I32, _I32, _P, _P, _P]),
    "e3k_norm_act_fwd": (C.c_int, [_P, _I64, _I32, C.POINTER(Block), _I32, _I32, _F, _I32, _P, _P]),
    "e3k_norm_act_bwd": (C.c_int, [_P, _P, _I64, _I32, C.POINTER(Block), _I32, _I32, _F, _I32, _P, _P]),
    "e3k_norm_act_bwd2": (C.c_int, [_P, _P, _P, _I64, _I32, C.POINTER(Block), _I32, _I32, _F, _I32, _P, _P, _P]),
    "e3k_layernorm_fwd": (C.c_int, [_P, _I64, _I32, C.POINTER(Block), _I32, _P, _P, _P, _P]),
    "e3k_layernorm_bwd": (C.c_int, [_P, _P, _P, _I64, _I32, C.POINTER(Block), _I32, _P, _P, _P, _P]),
    "e3k_layernorm_bwd2": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, C.POINTER(Block), _I32, _P, _P, _P, _P, _P]),
    "e3k_adam_ema_step": (C.c_int, [_P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _I32, _F, _I32, _P, _P]),
    "e3k_segment_sum": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libe3k.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libe3k.so not found at {LIB_PATH}: build it with `make -C {CSRC_DIR}` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the HIP path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().e3k_strerror(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (e3k status {code})")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device (every launch asks: the raw accessor skips the
    Python Stream object that ``torch.cuda.current_stream()`` builds, ~8 us per call)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_cuda(*tensors: torch.Tensor) -> None:
    """Operands must live on the CURRENT device: launches go to ``stream_ptr()``, the current stream of the current
    device, and the C ABI never calls hipSetDevice (one process per GPU; use ``torch.cuda.set_device`` / ``device()``)."""
    cur = _cur_device() if _cur_device is not None else torch.cuda.current_device()
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "e3_layers_amd kernels run on the GPU only (got a CPU tensor); there is no CPU "
                "fallback in the product path — the float64 CPU restatement lives in oracle/ for tests."
            )
        if t.device.index != cur:
            raise RuntimeError(
                f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: the kernels launch on the current "
                "device's stream — wrap the call in `with torch.cuda.device(tensor.device):`")


def f32c(t: torch.Tensor) -> torch.Tensor:
    """Contiguous fp32 view/copy."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
