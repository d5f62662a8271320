"""ctypes binding of libdcahip.so (include/dcahip.h) -- the only door to the HIP kernels.

There is NO fallback: if the shared library is missing, or a kernel is asked to run on anything
but device memory of an AMD GPU, this module raises.  PyTorch is used for what it is good at
here -- owning device buffers and streams; tensors cross the boundary as raw device pointers.
"""
import ctypes
import os

import torch

from . import build as _build

_c = ctypes
_f32p, _i32p, _i64p, _f64p, _vp = _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p

NLL_HAS_PI = 1
NLL_CONST_DISP = 2
NLL_POISSON = 4
NLL_MSE = 8

REG_MAX_SEGS = 16


class SmallLayer(_c.Structure):
    """dcahip_small_layer (include/dcahip.h)."""
    _fields_ = [('W', _c.c_void_p), ('ldw', _c.c_long), ('bias', _c.c_void_p), ('K', _c.c_int), ('H', _c.c_int),
                ('beta', _c.c_void_p), ('moving_mean', _c.c_void_p), ('moving_var', _c.c_void_p),
                ('Z', _c.c_void_p), ('ldz', _c.c_long), ('xhat', _c.c_void_p), ('ldx', _c.c_long),
                ('Hout', _c.c_void_p), ('ldh', _c.c_long), ('inv_std', _c.c_void_p)]


class StackBwdLayer(_c.Structure):
    """dcahip_stack_bwd_layer (include/dcahip.h)."""
    _fields_ = [('W', _c.c_void_p), ('ldw', _c.c_long), ('K', _c.c_int), ('H', _c.c_int),
                ('Hact', _c.c_void_p), ('ldh', _c.c_long), ('xhat', _c.c_void_p), ('ldx', _c.c_long),
                ('inv_std', _c.c_void_p), ('Hprev', _c.c_void_p), ('ldp', _c.c_long),
                ('gW', _c.c_void_p), ('ldg', _c.c_long), ('dbeta', _c.c_void_p), ('dH', _c.c_void_p), ('lddh', _c.c_long)]


class RegDesc(_c.Structure):
    """dcahip_reg_desc (include/dcahip.h)."""
    _fields_ = [('nseg', _c.c_int), ('start', _c.c_long * REG_MAX_SEGS), ('end', _c.c_long * REG_MAX_SEGS),
                ('l1', _c.c_float * REG_MAX_SEGS), ('l2', _c.c_float * REG_MAX_SEGS)]


ACT_CODES = {'linear': 0, 'relu': 1, 'tanh': 2, 'sigmoid': 3, 'elu': 4, 'selu': 5, 'softplus': 6,
             'softsign': 7, 'LeakyReLU': 8}

OPT_KINDS = {'sgd': 0, 'rmsprop': 1, 'adagrad': 2, 'adadelta': 3, 'adam': 4, 'adamax': 5}

_SIGNATURES = {
    'dcahip_version': (_c.c_int, []),
    'dcahip_zinb_max_partials': (_c.c_int, []),
    'dcahip_zinb_nll': (_c.c_int, [_f32p, _f32p, _f32p, _c.c_long, _f32p, _f32p, _c.c_long, _f32p,
                                   _i32p, _i64p, _c.c_int, _c.c_int, _c.c_float, _c.c_float,
                                   _c.c_int, _f32p, _f32p, _f32p, _c.c_long, _f64p,
                                   _c.POINTER(_c.c_int), _vp]),
    'dcahip_zinb_nll_planes': (_c.c_int, [_f32p, _f32p, _f32p, _c.c_long, _f32p, _f32p, _c.c_long, _f32p,
                                          _i32p, _i64p, _c.c_int, _c.c_int, _c.c_float, _c.c_float, _c.c_int,
                                          _c.c_void_p, _c.c_long, _c.c_long, _c.c_long, _c.c_long, _c.c_long,
                                          _f32p, _c.c_long, _f64p, _c.POINTER(_c.c_int), _vp]),
    'dcahip_loss_finalize': (_c.c_int, [_f64p, _c.c_int, _c.c_double, _f32p, _vp]),
    'dcahip_step_end': (_c.c_int, [_f32p, _c.c_double, _f32p, _c.c_int, _f64p, _i64p, _c.c_int, _vp]),
    'dcahip_zinb_heads_infer': (_c.c_int, [_f32p, _f32p, _f32p, _c.c_long, _f32p, _c.c_int, _c.c_int,
                                           _f32p, _f32p, _f32p, _c.c_long, _c.c_int, _vp]),
    'dcahip_heads_fused_workspace_bytes': (_c.c_long, [_c.c_int, _c.c_int, _c.c_int, _c.c_long, _c.c_int]),
    'dcahip_heads_fused': (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_long, _f32p,
                                      _f32p, _c.c_long, _f32p, _i32p, _i64p, _c.c_int, _c.c_int,
                                      _c.c_int, _c.c_float, _c.c_float, _c.c_int, _f32p, _c.c_long,
                                      _f32p, _f32p, _c.c_long, _f64p, _c.POINTER(_c.c_int), _vp,
                                      _c.c_long, _vp]),
    'dcahip_heads_tile_order_len': (_c.c_int, [_c.c_int]),
    'dcahip_x3_product_32x32': (_c.c_int, [_f32p, _f32p, _f32p, _c.c_int, _vp]),
    'dcahip_heads_fused_ordered': (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_long, _f32p,
                                              _f32p, _c.c_long, _f32p, _i32p, _i64p, _c.c_int, _c.c_int,
                                              _c.c_int, _c.c_float, _c.c_float, _c.c_int, _f32p, _c.c_long,
                                              _f32p, _f32p, _c.c_long, _f64p, _c.POINTER(_c.c_int), _vp,
                                              _c.c_long, _i32p, _vp]),
    'dcahip_transpose': (_c.c_int, [_f32p, _c.c_long, _c.c_int, _c.c_int, _f32p, _c.c_long, _vp]),
    'dcahip_hidden_small_chain': (_c.c_int, [_c.POINTER(SmallLayer), _c.c_int, _f32p, _c.c_long, _c.c_int, _c.c_int,
                                             _c.c_float, _c.c_float, _c.c_int, _vp]),
    'dcahip_transpose_rows': (_c.c_int, [_f32p, _c.c_long, _i32p, _i64p, _c.c_int, _c.c_int, _f32p, _c.c_long, _vp]),
    'dcahip_heads_fused_loss': (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_long, _f32p,
                                           _f32p, _c.c_long, _f32p, _i32p, _i64p, _c.c_int, _c.c_int,
                                           _c.c_int, _c.c_float, _c.c_float, _c.c_int, _f32p, _c.c_long,
                                           _f32p, _f32p, _c.c_long, _f64p, _c.POINTER(_c.c_int), _vp,
                                           _c.c_long, _i32p, _f32p, _vp]),
    'dcahip_sgemm': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _f32p, _c.c_long,
                                _f32p, _c.c_long, _f32p, _c.c_long, _f32p, _i32p, _i64p, _c.c_int,
                                _c.c_int, _vp, _c.c_long, _vp]),
    'dcahip_sgemm_workspace_bytes': (_c.c_long, [_c.c_int] * 7),
    'dcahip_split_planes': (_c.c_int, [_f32p, _c.c_long, _c.c_void_p, _c.c_void_p, _c.c_long, _c.c_int, _c.c_void_p,
                                       _c.c_long, _c.c_long, _c.c_void_p]),
    'dcahip_gemm_p3': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_long, _c.c_long,
                                  _c.c_void_p, _c.c_long, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_void_p, _c.c_void_p,
                                  _c.c_int, _c.c_int, _c.c_void_p, _c.c_long, _c.c_void_p]),
    'dcahip_gemm_p3_workspace_bytes': (_c.c_long, [_c.c_int] * 5),
    'dcahip_col_moments_chunks': (_c.c_int, [_c.c_int]),
    'dcahip_col_moments': (_c.c_int, [_f32p, _c.c_long, _c.c_int, _c.c_int, _f32p, _vp]),
    'dcahip_moments_combine': (_c.c_int, [_f32p, _f32p, _c.c_int, _c.c_int, _f32p, _vp]),
    'dcahip_bn_relu_apply': (_c.c_int, [_f32p, _c.c_long, _c.c_int, _c.c_int, _f32p, _f32p, _c.c_int,
                                        _f32p, _f32p, _f32p, _c.c_float, _c.c_float, _c.c_int,
                                        _f32p, _c.c_long, _f32p, _c.c_long, _f32p, _vp]),
    'dcahip_bn_bwd_sums': (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_long, _c.c_int,
                                      _c.c_int, _f32p, _c.c_int, _vp]),
    'dcahip_bn_bwd_apply': (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_long, _f32p,
                                       _f32p, _c.c_int, _c.c_float, _c.c_int, _c.c_int, _f32p,
                                       _c.c_long, _f32p, _c.c_int, _vp]),
    'dcahip_bn_fused_max_rows': (_c.c_int, []),
    'dcahip_dense_small_max_k': (_c.c_int, []),
    'dcahip_dense_bn_small': (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                         _f32p, _f32p, _f32p, _c.c_float, _c.c_float, _c.c_int, _f32p, _c.c_long, _f32p,
                                         _c.c_long, _f32p, _c.c_long, _f32p, _vp]),
    'dcahip_dense_bn_bwd_small': (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_long, _f32p, _f32p, _c.c_long,
                                             _f32p, _c.c_long, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_int,
                                             _f32p, _c.c_long, _f32p, _f32p, _c.c_long, _vp]),
    'dcahip_bn_relu_train_small': (_c.c_int, [_f32p, _c.c_long, _c.c_int, _c.c_int, _f32p, _f32p, _f32p, _c.c_float,
                                              _c.c_float, _c.c_int, _f32p, _c.c_long, _f32p, _c.c_long, _f32p, _vp]),
    'dcahip_bn_bwd_small': (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_float,
                                       _c.c_int, _c.c_int, _f32p, _c.c_long, _f32p, _c.c_int, _vp]),
    'dcahip_rmsprop_clip_end': (_c.c_int, [_f32p, _f32p, _f32p, _c.c_long, _f32p, _c.c_float, _c.c_float, _c.c_float,
                                           _f32p, _c.c_double, _f32p, _c.c_int, _f64p, _i64p, _c.c_int, _vp]),
    'dcahip_relu_bwd': (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _c.c_int, _c.c_int, _f32p,
                                   _c.c_long, _c.c_int, _vp]),
    'dcahip_relu_fwd': (_c.c_int, [_f32p, _c.c_long, _c.c_int, _c.c_int, _f32p, _c.c_long, _c.c_int, _vp]),
    'dcahip_colsum_chain': (_c.c_int, [_f32p, _c.c_long, _c.c_int, _c.c_int, _f32p, _f32p, _vp]),
    'dcahip_optimizer_step': (_c.c_int, [_c.c_int, _f32p, _f32p, _f32p, _f32p, _c.c_long, _f32p, _i64p,
                                         _c.c_float, _vp]),
    'dcahip_counter_add': (_c.c_int, [_i64p, _c.c_int, _vp]),
    'dcahip_prelu_workspace_doubles': (_c.c_int, [_c.c_int]),
    'dcahip_prelu_fwd': (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_int, _c.c_int, _f32p, _c.c_long, _vp]),
    'dcahip_prelu_bwd': (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_int, _c.c_int, _f32p, _f64p, _vp]),
    'dcahip_elempi_workspace_doubles': (_c.c_int, [_c.c_int]),
    'dcahip_elempi_fwd': (_c.c_int, [_f32p, _c.c_long, _f32p, _f32p, _c.c_int, _c.c_int, _f32p, _c.c_long, _vp]),
    'dcahip_elempi_bwd': (_c.c_int, [_f32p, _c.c_long, _f32p, _f32p, _c.c_long, _f32p, _c.c_int, _c.c_int, _f32p, _f32p,
                                     _f64p, _vp]),
    'dcahip_bcast_cols': (_c.c_int, [_f32p, _c.c_long, _c.c_int, _c.c_int, _f32p, _c.c_long, _vp]),
    'dcahip_row_sums_strided': (_c.c_int, [_f32p, _c.c_long, _c.c_int, _c.c_int, _f32p, _c.c_long, _vp]),
    'dcahip_nadam_step': (_c.c_int, [_f32p, _f32p, _f32p, _f32p, _c.c_long, _f32p, _i64p, _f32p, _c.c_float, _vp]),
    'dcahip_dropout_apply': (_c.c_int, [_f32p, _c.c_long, _i32p, _i64p, _c.c_int, _c.c_int, _c.c_float,
                                        _c.c_ulonglong, _i64p, _c.c_int, _c.c_long, _f32p, _c.c_long, _vp]),
    'dcahip_l1l2_workspace_doubles': (_c.c_int, []),
    'dcahip_l1l2_apply': (_c.c_int, [_c.POINTER(RegDesc), _f32p, _f32p, _f32p, _f64p, _vp]),
    'dcahip_prep_chunks': (_c.c_int, [_c.c_int]),
    'dcahip_prep_row_sums': (_c.c_int, [_f32p, _c.c_long, _c.c_int, _c.c_int, _f32p, _vp]),
    'dcahip_prep_col_pass': (_c.c_int, [_f32p, _c.c_long, _c.c_int, _c.c_int, _f32p, _c.c_int, _f32p,
                                        _c.c_long, _f64p, _vp]),
    'dcahip_prep_col_finish': (_c.c_int, [_f64p, _c.c_int, _c.c_int, _c.c_double, _f32p, _f32p, _f32p, _vp]),
    'dcahip_prep_scale': (_c.c_int, [_f32p, _c.c_long, _c.c_int, _c.c_int, _f32p, _f32p, _vp]),
    'dcahip_rmsprop_clip': (_c.c_int, [_f32p, _f32p, _f32p, _c.c_long, _f32p, _c.c_float,
                                       _c.c_float, _c.c_float, _vp]),
    'dcahip_hidden_stack_max_rows': (_c.c_int, []),
    'dcahip_hidden_stack_workspace_bytes': (_c.c_long, [_c.c_int, _c.c_int]),
    'dcahip_hidden_stack_fwd': (_c.c_int, [_c.POINTER(SmallLayer), _c.c_int, _c.c_int, _c.c_float, _c.c_float, _c.c_int,
                                           _c.c_int, _c.c_int, _c.c_int, _vp, _c.c_long, _vp]),
    'dcahip_hidden_stack_bwd': (_c.c_int, [_c.POINTER(StackBwdLayer), _c.c_int, _c.c_int, _c.c_float, _c.c_int,
                                           _f32p, _c.c_long, _c.c_int, _c.c_int, _c.c_int, _vp, _c.c_long, _vp]),
    'dcahip_hidden_stack_step_blocks': (_c.c_int, [_c.c_int]),
    'dcahip_hidden_stack_fwd_sync': (_c.c_int, [_c.POINTER(SmallLayer), _c.c_int, _c.c_int, _c.c_float, _c.c_float, _c.c_int,
                                                _c.c_int, _f32p, _f32p, _c.c_int, _f32p, _vp, _c.c_long, _vp]),
    'dcahip_hidden_stack_bwd_sync': (_c.c_int, [_c.POINTER(StackBwdLayer), _c.c_int, _c.c_int, _c.c_float, _c.c_int,
                                                _f32p, _c.c_long, _c.c_int, _f32p, _f32p, _vp, _c.c_long, _vp]),
    'dcahip_counts_compact_ld': (_c.c_long, [_c.c_int]),
    'dcahip_counts_compact': (_c.c_int, [_f32p, _c.c_long, _c.c_int, _c.c_int, _vp, _c.c_long, _i32p, _vp]),
    'dcahip_enc0_sparse_supported': (_c.c_int, [_c.c_int]),
    'dcahip_enc0_dw_sparse_workspace_bytes': (_c.c_long, [_c.c_int, _c.c_int, _c.c_int]),
    'dcahip_enc0_lut': (_c.c_int, [_f32p, _c.c_int, _c.c_int, _vp, _vp]),
    'dcahip_enc0_dw_sparse': (_c.c_int, [_vp, _c.c_long, _i32p, _i32p, _f32p, _f32p, _c.c_int, _f32p, _f32p, _f32p, _i32p, _i64p,
                                         _c.c_long, _c.c_int, _c.c_int, _c.c_int, _f32p, _c.c_long, _f32p, _c.c_long,
                                         _vp, _c.c_long, _c.c_int, _vp]),
    'dcahip_peer_slot_bytes': (_c.c_long, [_c.c_int, _c.c_int]),
    'dcahip_peer_flag_bytes': (_c.c_long, [_c.c_int]),
    'dcahip_peer_exchange': (_c.c_int, [_f32p, _c.c_int, _vp, _vp, _c.c_int, _c.c_int, _c.c_int, _vp, _f32p, _c.c_int, _i32p,
                                        _c.c_long, _vp]),
    'dcahip_enc0_lut_entries': (_c.c_int, []),
    'dcahip_enc0_fwd_lut_workspace_bytes': (_c.c_long, [_c.c_int, _c.c_int, _c.c_int]),
    'dcahip_enc0_fwd_lut': (_c.c_int, [_vp, _c.c_long, _i32p, _i32p, _f32p, _f32p, _c.c_int, _vp, _f32p, _f32p, _i32p, _i64p,
                                       _c.c_long, _c.c_int, _c.c_int, _c.c_int, _f32p, _c.c_long, _f32p, _f32p, _c.c_long,
                                       _vp, _c.c_long, _vp]),
    'dcahip_absmax_exp': (_c.c_int, [_f32p, _c.c_long, _c.c_long, _c.c_int, _i32p, _vp, _vp]),
    'dcahip_split_planes_h2': (_c.c_int, [_f32p, _c.c_long, _i32p, _i64p, _c.c_long, _c.c_int, _vp, _c.c_long, _c.c_long, _i32p, _vp]),
    'dcahip_gemm_h2_supported': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int]),
    'dcahip_gemm_h2_workspace_bytes': (_c.c_long, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'dcahip_gemm_h2': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _vp, _c.c_long, _c.c_long, _i32p, _c.c_int,
                                  _vp, _c.c_long, _c.c_long, _i32p, _c.c_int, _c.c_float, _f32p, _c.c_long, _f32p, _c.c_int, _c.c_int,
                                  _vp, _c.c_long, _vp]),
    'dcahip_zinb_nll_planes_h2': (_c.c_int, [_f32p, _f32p, _f32p, _c.c_long, _f32p, _f32p, _c.c_long, _f32p,
                                             _i32p, _i64p, _c.c_int, _c.c_int, _c.c_float, _c.c_float, _c.c_int, _c.c_int,
                                             _vp, _c.c_long, _c.c_long, _c.c_long, _c.c_long, _c.c_long, _f32p, _c.c_long, _f64p,
                                             _c.POINTER(_c.c_int), _vp]),
    'dcahip_heads_fused_compact': (_c.c_int, [_f32p, _c.c_long, _f32p, _c.c_long, _f32p, _c.c_long, _f32p,
                                              _f32p, _c.c_long, _vp, _c.c_long, _i32p, _i32p, _f32p, _f32p, _i32p, _i64p,
                                              _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_float, _c.c_int, _f32p,
                                              _c.c_long, _f32p, _f32p, _c.c_long, _f64p, _c.POINTER(_c.c_int), _vp,
                                              _c.c_long, _i32p, _f32p, _c.c_int, _vp]),
}

# Entry points that exist only in EXPERIMENT builds of the sources (-DDCA_EXP_DW_SMALL, -DDCA_EXP_ENC0_SPARSE_FWD: kernels
# that were measured and lost, include/dcahip.h conventions): bound when the loaded library has them (tools/ point build.LIB
# at such a build), absent from the product library (tests/test_cabi.py).
_EXPERIMENT_SIGNATURES = {
    'dcahip_enc0_dw_small_max_rows': (_c.c_int, []),
    'dcahip_enc0_dw_small': (_c.c_int, [_vp, _c.c_long, _i32p, _i32p, _f32p, _f32p, _c.c_int, _f32p, _f32p, _i32p, _i64p,
                                        _c.c_long, _c.c_int, _c.c_int, _c.c_int, _f32p, _c.c_long, _f32p, _c.c_long, _vp]),
    'dcahip_enc0_fwd_sparse_workspace_bytes': (_c.c_long, [_c.c_int]),
    'dcahip_enc0_fwd_sparse': (_c.c_int, [_vp, _c.c_long, _i32p, _i32p, _f32p, _f32p, _c.c_int, _f32p, _f32p, _i32p, _i64p,
                                          _c.c_long, _c.c_int, _c.c_int, _c.c_int, _f32p, _c.c_long, _f32p, _f32p, _c.c_long,
                                          _vp, _c.c_long, _vp]),
}

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def lib():
    """Loads libdcahip.so once (building it with hipcc first if it is absent or stale)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if _build.needs_build():
        try:
            _build.build_hip(verbose=False)
        except Exception as e:  # noqa: BLE001
            if not os.path.exists(path):
                raise HipExtensionMissing(
                    'dca_amd: libdcahip.so is missing and could not be built (%s). The HIP '
                    'extension is mandatory; there is no CPU fallback.' % e) from e
    try:
        L = ctypes.CDLL(path)
    except OSError as e:
        raise HipExtensionMissing('dca_amd: cannot load %s: %s' % (path, e)) from e
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError => header / library mismatch: loud
        fn.restype, fn.argtypes = res, args
    for name, (res, args) in _EXPERIMENT_SIGNATURES.items():
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
    assert L.dcahip_version() == 1
    _lib = L
    return L


def exported_symbols():
    return sorted(_SIGNATURES)


def has(name):
    """Whether the loaded library exports `name` (experiment entry points: only in -D builds)."""
    return hasattr(lib(), name)


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError('dca_amd: no AMD GPU visible (torch.cuda.is_available() is False); the '
                           'training path runs on MI355X only -- there is no CPU fallback.')


def ptr(t):
    """Device pointer of a tensor (None -> NULL). Refuses host memory."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('dca_amd.hip: host tensor passed to a HIP kernel')
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """Raw handle of torch's current stream on the current device (every kernel is launched on it).  The
    private fast getter saves ~8 us per launch over torch.cuda.current_stream() (two dozen launches per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        raise RuntimeError('dca_amd.hip: %s failed with code %d' % (what, rc))
