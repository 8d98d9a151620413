"""ctypes binding of libclica_hip.so (C ABI: include/clica.h).

The HIP library IS the product path: if it is missing or a call fails this module raises --
there is no eager/PyTorch fallback anywhere in cl_ica_amd.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLICA_LIB") or os.path.join(_HERE, "lib", "libclica_hip.so")     # CLICA_LIB: debug builds (tools/fmlp_trace.py)

c_f32p = C.c_void_p
c_i64 = C.c_int64
c_i32 = C.c_int32
c_size = C.c_size_t


class LpLossDesc(C.Structure):
    _fields_ = [("B", c_i64), ("B3", c_i64), ("n", c_i32), ("p", C.c_float), ("tau", C.c_float),
                ("alpha", C.c_float), ("compat", c_i32), ("pow", c_i32), ("no_eps", c_i32)]


class AdamDesc(C.Structure):
    """clica_adam_desc (include/clica.h): the optimizer applied by clica_mlp_wgrad_split_adam's reduction launch."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("count", c_i64),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("grad_scale", C.c_float),
                ("step_dev", C.c_void_p), ("t_offset", c_i32), ("split16_state", C.c_void_p), ("n_layers", c_i32)]


class DyParts(C.Structure):
    """clica_lp_dy_parts (include/clica.h): the pair sweep's partials + what the loss's reduction launch would have done with them."""
    _fields_ = [("part", C.c_void_p), ("nsplit", c_i32), ("nsplit_alt", c_i32), ("np", c_i32), ("n", c_i32), ("rows", c_i64),
                ("guard_words", C.c_void_p), ("guard_limit", C.c_float), ("blocksums", C.c_void_p), ("nblocks", c_i32),
                ("inv_count", C.c_float), ("means", C.c_void_p), ("tick", C.c_void_p)]


class ChainTail(C.Structure):
    """clica_chain_tail (include/clica.h): what the backward chain needs to leave the n-wide layers' weight-gradient slabs."""
    _fields_ = [("a_last", C.c_void_p), ("lda", c_i64), ("x", C.c_void_p), ("ldx", c_i64), ("n_layers", c_i32),
                ("N", C.POINTER(c_i32)), ("K", C.POINTER(c_i32)), ("wgrad_workspace", C.c_void_p), ("wgrad_workspace_bytes", c_size),
                ("dy_parts", C.POINTER(DyParts))]


class DotLossDesc(C.Structure):
    _fields_ = [("B", c_i64), ("B3", c_i64), ("n", c_i32), ("tau", C.c_float), ("alpha", C.c_float),
                ("normalize", c_i32)]


class SamplerDesc(C.Structure):
    _fields_ = [("space", c_i32), ("dist", c_i32), ("n", c_i32), ("box_min", C.c_float),
                ("box_max", C.c_float), ("scale", C.c_float), ("shape_p", C.c_float),
                ("seed", C.c_uint64), ("stream_id", C.c_uint32)]


_LOSS_FWD = [c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, C.c_void_p, c_size, C.c_void_p]
_LOSS_BWD = [c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_f32p,
             c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_i32, C.c_void_p, c_size, C.c_void_p]

# name -> argtypes (restype is int for all but clica_last_error); mirrors include/clica.h
SIGNATURES: Dict[str, list] = {
    "clica_version": [],
    "clica_lp_loss_workspace_bytes": [C.POINTER(LpLossDesc), C.POINTER(c_size), C.POINTER(c_size)],
    "clica_lp_loss_fwd": [C.POINTER(LpLossDesc)] + _LOSS_FWD,
    "clica_lp_loss_bwd": [C.POINTER(LpLossDesc)] + _LOSS_BWD,
    "clica_lp_loss_bwd_sym": [C.POINTER(LpLossDesc), c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                              c_f32p, c_i64, c_f32p, c_i64, C.c_void_p, c_size, C.c_void_p],
    "clica_lp_loss_train_workspace_bytes": [C.POINTER(LpLossDesc), C.POINTER(c_size)],
    "clica_lp_loss_train_path": [C.POINTER(LpLossDesc), C.POINTER(c_i32)],
    "clica_lp_loss_set_matrix_cores": [c_i32],
    "clica_lp_loss_train_spread": [C.POINTER(LpLossDesc), C.c_void_p, c_size, C.POINTER(C.c_float), C.c_void_p],
    "clica_lp_loss_train_guard": [C.POINTER(LpLossDesc), C.c_void_p, c_size, C.POINTER(C.c_float), C.c_void_p],
    "clica_lp_loss_set_spread_limit": [C.c_float],
    "clica_lp_loss_fwd_train": [C.POINTER(LpLossDesc), c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_f32p, c_f32p,
                                c_f32p, c_i64, c_f32p, c_i64, C.c_void_p, c_size, C.c_void_p],
    "clica_lp_loss_bwd_sym_train": [C.POINTER(LpLossDesc), c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_i64, c_f32p, C.c_void_p,
                                    C.c_void_p, c_size, C.c_void_p],
    "clica_lp_loss_bwd_sym_train_parts": [C.POINTER(LpLossDesc), c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                          C.c_void_p, c_size, C.POINTER(DyParts), C.c_void_p],
    "clica_dot_loss_workspace_bytes": [C.POINTER(DotLossDesc), C.POINTER(c_size), C.POINTER(c_size)],
    "clica_dot_loss_fwd": [C.POINTER(DotLossDesc)] + _LOSS_FWD,
    "clica_dot_loss_bwd": [C.POINTER(DotLossDesc)] + _LOSS_BWD,
    "clica_linear_fwd": [c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_f32p, c_i64, c_i64, c_i64, c_i64, c_i32, C.c_float, C.c_void_p],
    "clica_linear_dgrad": [c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, C.c_float, c_f32p, c_i64, c_i64, c_i64, c_i64, C.c_void_p],
    "clica_linear_wgrad_workspace_bytes": [c_i64, c_i64, c_i64, C.POINTER(c_size)],
    "clica_linear_plan": [c_i32, c_i64, c_i64, c_i64, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i32)],
    "clica_linear_wgrad": [c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_i64, c_i64, c_i32, C.c_void_p, c_size, C.c_void_p],
    "clica_mlp_fwd": [c_f32p, c_i64, c_i64, c_i32, C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p),
                      C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(c_i32), C.POINTER(c_i32), c_f32p, C.POINTER(C.c_void_p), C.c_float, C.c_void_p],
    "clica_mlp_fwd_mixed": [c_f32p, c_i64, c_i64, c_f32p, c_i32, C.c_float, c_f32p, c_i64, c_i32,
                            C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p),
                            C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(c_i32), C.POINTER(c_i32), c_f32p, C.POINTER(C.c_void_p),
                            C.c_float, C.c_void_p],
    "clica_mlp_signmask_bytes": [c_i64, C.POINTER(c_size)],
    "clica_mlp_pack_split_bytes": [c_i32, C.POINTER(c_i32), C.POINTER(c_i32), c_i32, C.POINTER(c_size)],
    "clica_mlp_pack_split": [c_i32, C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(c_i32), C.POINTER(c_i32), c_i32, C.c_void_p, C.c_void_p],
    "clica_mlp_pack_split_both": [c_i32, C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(c_i32), C.POINTER(c_i32), C.c_void_p, C.c_void_p,
                                  C.c_void_p],
    "clica_mlp_planes_bytes": [c_i64, c_i32, c_i32, C.POINTER(c_size)],
    "clica_mlp_fwd_split": [c_f32p, c_i64, c_i64, c_f32p, c_i32, C.c_float, c_f32p, c_i64, c_i32, C.POINTER(C.c_void_p),
                            C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(c_i32), C.POINTER(c_i32), C.c_void_p, C.POINTER(C.c_void_p),
                            C.POINTER(C.c_void_p), C.c_float, C.c_void_p],
    "clica_mlp_dgrad_split": [c_f32p, c_i64, c_i64, c_i32, C.POINTER(c_i32), C.POINTER(c_i32), C.c_void_p, C.POINTER(C.c_void_p),
                              C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p), C.c_float, C.c_void_p],
    "clica_mlp_wgrad_split_kind": [c_i32, c_i32, C.POINTER(c_i32)],
    "clica_mlp_planes_from_f32": [c_f32p, c_i64, c_i64, c_i32, c_i32, C.c_void_p, C.c_void_p],
    "clica_mlp_planes_from_f32_t": [c_f32p, c_i64, c_i64, c_i32, C.c_void_p, C.c_void_p],
    "clica_linear_split_fwd": [C.c_void_p, C.c_void_p, c_f32p, c_i64, c_i32, c_i32, c_i32, C.c_float, C.c_void_p, C.c_void_p, c_i32,
                               c_f32p, c_i64, C.c_void_p],
    "clica_linear_split_dgrad": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, c_i64, c_i32, c_i32, C.c_void_p, C.c_void_p,
                                 c_f32p, c_i64, C.c_void_p],
    "clica_mlp_wgrad_split_workspace_bytes": [c_i64, c_i32, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_size)],
    "clica_mlp_wgrad_split": [c_i64, c_i32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(c_i64),
                              C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p),
                              C.POINTER(c_i32), C.POINTER(c_i32), c_i32, C.c_void_p, c_size, C.c_void_p],
    # f16x2 arithmetic (round 5)
    "clica_split16_state_bytes": [C.POINTER(c_size)],
    "clica_split16_state_init": [C.c_void_p, C.c_void_p],
    "clica_split16_update": [C.c_void_p, c_i32, C.c_void_p],
    "clica_split16_read": [C.c_void_p, C.POINTER(c_i32), C.POINTER(c_i32), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p],
    "clica_split16_clear_flags": [C.c_void_p, C.c_void_p],
    "clica_split16_guard": [C.c_void_p, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i32), C.c_void_p],
    "clica_split16_set_dp_poison": [C.c_void_p, C.c_void_p, C.c_void_p],
    "clica_split16_poison_export": [C.c_void_p, C.c_void_p, C.c_void_p],
    "clica_mlp_planes16_bytes": [c_i64, c_i32, c_i32, C.POINTER(c_size)],
    "clica_mlp_pack_split16_bytes": [c_i32, C.POINTER(c_i32), C.POINTER(c_i32), c_i32, C.POINTER(c_size)],
    "clica_mlp_pack_split16_both": [c_i32, C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(c_i32), C.POINTER(c_i32), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p],
    "clica_mlp_fwd_split16": [c_f32p, c_i64, c_i64, c_f32p, c_i32, C.c_float, c_f32p, c_i64, c_i32, C.POINTER(C.c_void_p),
                              C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(c_i32), C.POINTER(c_i32), C.c_void_p, C.POINTER(C.c_void_p),
                              C.POINTER(C.c_void_p), C.c_float, C.c_void_p, C.c_void_p],
    "clica_mlp_dgrad_split16": [c_f32p, c_i64, c_i64, c_i32, C.POINTER(c_i32), C.POINTER(c_i32), C.c_void_p, C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p), C.c_float, C.c_void_p, C.c_void_p],
    "clica_mlp_wgrad_split16": [c_i64, c_i32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(c_i64),
                                C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p),
                                C.POINTER(c_i32), C.POINTER(c_i32), c_i32, C.c_void_p, C.POINTER(c_i32), C.POINTER(c_i32), C.c_void_p, c_size,
                                C.c_void_p],
    "clica_mlp_wgrad_split16_tail": [c_i64, c_i32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(c_i64),
                                     C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p),
                                     C.POINTER(c_i32), C.POINTER(c_i32), c_i32, C.c_void_p, C.POINTER(c_i32), C.POINTER(c_i32), c_i32, C.c_void_p, c_size,
                                     C.c_void_p],
    "clica_mlp_wgrad_split_adam": [c_i64, c_i32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(c_i64),
                                   C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p),
                                   C.POINTER(c_i32), C.POINTER(c_i32), C.c_void_p, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(AdamDesc),
                                   c_i32, C.c_void_p, c_size, C.c_void_p],
    "clica_mlp_chain_tail_supported": [c_i32, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i32)],
    "clica_mlp_dgrad_split_tail": [c_f32p, c_i64, c_i64, c_i32, C.POINTER(c_i32), C.POINTER(c_i32), C.c_void_p, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p), C.c_float, C.c_void_p,
                                   C.POINTER(ChainTail), C.c_void_p],
    "clica_mlp_planes16_from_f32": [c_f32p, c_i64, c_i64, c_i32, c_i32, C.c_void_p, C.c_void_p, c_i32, c_i32, C.c_void_p],
    "clica_mlp_planes16_from_f32_t": [c_f32p, c_i64, c_i64, c_i32, C.c_void_p, C.c_void_p, c_i32, c_i32, C.c_void_p],
    "clica_linear_split_fwd16": [C.c_void_p, C.c_void_p, c_f32p, c_i64, c_i32, c_i32, c_i32, C.c_float, C.c_void_p, C.c_void_p, c_i32,
                                 c_f32p, c_i64, C.c_void_p, c_i32, C.c_void_p],
    "clica_linear_split_dgrad16": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, c_i64, c_i32, c_i32, C.c_void_p, C.c_void_p,
                                   c_f32p, c_i64, C.c_void_p, c_i32, C.c_void_p],
    "clica_mlp_pack_bytes": [c_i32, C.POINTER(c_i32), C.POINTER(c_i32), c_i32, C.POINTER(c_size)],
    "clica_mlp_pack": [c_i32, C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(c_i32), C.POINTER(c_i32), c_i32, c_f32p, C.c_void_p],
    "clica_mlp_pack_both": [c_i32, C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(c_i32), C.POINTER(c_i32), c_f32p, c_f32p, C.c_void_p],
    "clica_mlp_dgrad": [c_f32p, c_i64, c_i64, c_i32, C.POINTER(c_i32), C.POINTER(c_i32), c_f32p, C.POINTER(C.c_void_p), C.POINTER(c_i64),
                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(c_i64), C.c_float, C.c_void_p],
    "clica_mlp_wgrad_workspace_bytes": [c_i64, c_i32, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_size)],
    "clica_mlp_wgrad": [c_i64, c_i32, C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p), C.POINTER(c_i64),
                        C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(C.c_void_p), C.POINTER(c_i32), C.POINTER(c_i32), c_i32,
                        C.c_void_p, c_size, C.c_void_p],
    "clica_rescale_fwd": [c_f32p, c_i64, c_f32p, c_f32p, c_i64, c_f32p, c_i64, c_i32, C.c_void_p],
    "clica_rescale_bwd": [c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_i32, C.c_void_p],
    "clica_softclip_fwd": [c_f32p, c_i64, c_f32p, c_f32p, c_i64, c_i64, c_i32, C.c_void_p],
    "clica_softclip_bwd": [c_f32p, c_i64, c_f32p, c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_i32, C.c_void_p],
    "clica_leaky_relu_fwd": [c_f32p, c_i64, c_f32p, c_i64, c_i64, c_i32, C.c_float, C.c_void_p],
    "clica_leaky_relu_bwd": [c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_i64, c_i32, C.c_float, C.c_void_p],
    "clica_conv_im2col_k4s2": [c_f32p, c_i64, c_i32, c_i32, c_i32, c_f32p, C.c_void_p],
    "clica_conv_k4s2_fwd_patches": [c_f32p, c_f32p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32p, C.c_void_p, C.c_void_p],
    "clica_conv_k4s2_fwd": [c_f32p, c_f32p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32p, C.c_void_p, C.c_void_p],
    "clica_conv_k4s2_dgrad": [c_f32p, c_f32p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32p, c_i32, c_i32, C.c_void_p, C.c_void_p],
    "clica_conv_k4s2_wgrad_workspace_bytes": [c_i64, c_i32, c_i32, C.POINTER(c_size)],
    "clica_conv_k4s2_wgrad": [c_f32p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32p, c_f32p, c_i32, C.c_void_p, c_size, C.c_void_p],
    "clica_conv_k4s2_wgrad_patches_workspace_bytes": [c_i64, c_i32, c_i32, C.POINTER(c_size)],
    "clica_conv_k4s2_wgrad_patches": [c_f32p, c_f32p, c_i64, c_i32, c_i32, c_f32p, c_f32p, c_i32, C.c_void_p, c_size, C.c_void_p],
    "clica_conv_k4s2_dgrad_input": [c_f32p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32p, C.c_void_p],
    "clica_conv_gather": [c_i32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i32, C.c_void_p],
    "clica_conv_k4s2_fwd_image": [c_f32p, c_f32p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p],
    "clica_conv_k4s2_wgrad_image": [c_f32p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_f32p, c_f32p, c_i32, C.c_void_p, c_size, C.c_void_p],
    "clica_conv16_pack": [c_i32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_f32p, C.c_void_p],
    "clica_conv16_amax": [c_f32p, c_i64, C.c_void_p, C.c_void_p],
    "clica_conv16_zero_slots": [C.c_void_p, c_i32, C.c_void_p],
    "clica_conv_k4s2_fwd_patches_amax": [c_f32p, c_f32p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p],
    "clica_conv16_first_fwd": [c_f32p, C.c_void_p, c_f32p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "clica_conv16_k4s2_fwd": [c_f32p, C.c_void_p, c_f32p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "clica_conv16_k4s2_dgrad": [c_f32p, C.c_void_p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32p, c_i32, c_i32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "clica_conv16_k4s2_wgrad_workspace_bytes": [c_i64, c_i32, c_i32, C.POINTER(c_size)],
    "clica_conv16_k4s2_wgrad": [c_f32p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32p, c_f32p, c_i32, C.c_void_p, C.c_void_p, C.c_void_p, c_size, C.c_void_p],
    "clica_mixing_fwd_act": [c_f32p, c_i64, c_f32p, c_i32, c_i32, C.c_float, c_f32p, c_i64, c_i64, c_i32, C.c_void_p],
    "clica_mixing_fwd": [c_f32p, c_i64, c_f32p, c_i32, C.c_float, c_f32p, c_i64, c_i64, c_i32, C.c_void_p],
    "clica_adam_step": [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p],
    "clica_adam_step_at": [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, c_i32,
                           C.c_void_p],
    "clica_adam_step_s16": [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, c_i32,
                            C.c_void_p, c_i32, C.c_void_p],
    "clica_adam_step_tick": [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                             C.c_void_p, C.c_void_p],
    "clica_adam_step_s16_tick": [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                 C.c_void_p, C.c_void_p, c_i32, C.c_void_p],
    "clica_tick": [C.c_void_p, C.c_void_p],
    "clica_stamp": [C.c_void_p, c_i32, c_i32, C.c_void_p],
    "clica_publish_host": [C.c_void_p, c_i32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "clica_clock_probe": [C.c_void_p, c_i32, c_i32, C.c_void_p],
    "clica_set_tuning": [C.c_char_p, c_i32],
    "clica_abort_capture": [C.c_void_p],
    "clica_kitti_gather_pairs": [C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, c_i64, c_f32p, c_f32p, c_i32, c_f32p, C.c_void_p],
    "clica_moments_workspace_bytes": [c_i64, c_i32, C.POINTER(c_size)],
    "clica_moments": [c_f32p, c_i64, c_i32, c_f32p, c_i64, c_i32, c_i64, C.c_void_p, C.c_void_p, c_size, C.c_void_p],
    "clica_nn_search_workspace_bytes": [c_i64, c_i64, c_i32, c_i32, C.POINTER(c_size)],
    "clica_nn_search": [c_f32p, c_i64, c_i64, c_f32p, c_i64, c_i64, c_i32, c_i32, C.c_void_p, c_f32p, C.c_void_p, c_size, C.c_void_p],
    "clica_sample": [C.POINTER(SamplerDesc), c_f32p, c_i64, c_f32p, c_i64, c_i64, C.c_void_p, C.c_void_p],
    "clica_sample_scaled": [C.POINTER(SamplerDesc), c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_i64, C.c_void_p, C.c_void_p],
    "clica_sample_pair": [C.POINTER(SamplerDesc), C.POINTER(SamplerDesc), c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_i64,
                          C.c_void_p, C.c_void_p],
    "clica_mlp_pack_split16_both_sample": [c_i32, C.POINTER(C.c_void_p), C.POINTER(c_i64), C.POINTER(c_i32), C.POINTER(c_i32), C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.POINTER(SamplerDesc), C.POINTER(SamplerDesc), c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64,
                                           c_i64, C.c_void_p, C.c_void_p],
}

_lib: Optional[C.CDLL] = None


class ClicaError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library or raise.  Never falls back to anything."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ClicaError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cl_ica_amd/csrc -j8`).  cl_ica_amd has no CPU/eager fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.clica_last_error.restype = C.c_char_p
    lib.clica_last_error.argtypes = []
    missing = []
    for name, args in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.argtypes = args
        fn.restype = C.c_int
    if missing:
        raise ClicaError(f"{LIB_PATH} does not export {missing}; rebuild it (stale library?)")
    _lib = lib
    return lib


_DEBUG_SYNC = os.environ.get("CLICA_DEBUG_SYNC", "0") != "0"      # name every library call on stderr and drain the device behind it


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().clica_last_error().decode("utf-8", "replace")
        raise ClicaError(f"{what} failed (rc={rc}): {msg}")
    if _DEBUG_SYNC and not torch.cuda.is_current_stream_capturing():
        import sys
        sys.stderr.write(f"[clica] {what}\n"); sys.stderr.flush()
        torch.cuda.synchronize()


def require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise ClicaError(f"{name} must live on the GPU (got device {t.device}); cl_ica_amd runs only "
                         "through its HIP kernels")
    if t.dtype != torch.float32:
        raise ClicaError(f"{name} must be float32 (got {t.dtype})")


def stream_ptr() -> int:
    # raw handle of the current stream of the current device: what torch.cuda.current_stream().cuda_stream returns, without building
    # the Python Stream object (11 us per call, eleven calls per drop-in training step)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def rowmajor(t: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """Return (tensor usable by the kernels, leading dimension).  Strided row views
    (mu[::2], z[:, :k]) are passed through without a copy."""
    assert t.dim() == 2
    if t.shape[1] == 1 or t.stride(1) == 1:
        ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
        if ld >= t.shape[1]:
            return t, ld
    t = t.contiguous()
    return t, t.shape[1]


# zero-initialised scratch per (device, stream, tag): the loss forward keeps a ticket word in it
_WS: Dict[Tuple[int, int, str], torch.Tensor] = {}


def workspace(tag: str, nbytes: int, device: torch.device) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream_ptr(), tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.zeros(max(nbytes, 1024), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf
