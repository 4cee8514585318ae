"""ctypes binding of ``libspectrum_hip.so`` (declared in ``include/spectrum_hip.h``).

The library is the only compute backend of this package: there is no CPU or
eager fallback.  If the shared object is missing or does not export a declared
symbol, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPECTRUM_HIP_LIB selects another build of the same library (the diagnostic trace build)
LIB_PATH = os.environ.get("SPECTRUM_HIP_LIB") or os.path.join(_HERE, "libspectrum_hip.so")

ACT = {None: 0, "none": 0, "linear": 0, "relu": 1, "elu": 2, "lrelu": 3, "tanh": 4, "sigmoid": 5}
F32, BF16 = 0, 1


class ConvGeom(C.Structure):
    """Mirror of ``sv_conv_geom``."""

    _fields_ = [(n, C.c_int32) for n in
                ("N", "H", "W", "Cin", "KH", "KW", "SH", "SW", "pad_t", "pad_l", "Ho", "Wo", "Cout", "up_h", "up_w")]

    def key(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)


class NetLayer(C.Structure):
    """Mirror of ``sv_net_layer``."""

    _fields_ = [("Cin", C.c_int32), ("Cout", C.c_int32), ("K", C.c_int32), ("up", C.c_int32), ("pool", C.c_int32),
                ("act", C.c_int32), ("alpha", C.c_float), ("reserved", C.c_int32), ("w_off", C.c_int64),
                ("b_off", C.c_int64), ("y", C.c_void_p), ("idx", C.c_void_p)]


class WgradJob(C.Structure):
    """Mirror of ``sv_wgrad_job``."""

    _fields_ = [("g", ConvGeom), ("workspace", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p),
                ("accumulate", C.c_int32), ("blocks", C.c_int32)]


class GradSlabs(C.Structure):
    """Mirror of ``sv_grad_slabs``."""

    _fields_ = [("slabs", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("slab_floats", C.c_int32),
                ("n_split", C.c_int32), ("n_dw", C.c_int32), ("n_db", C.c_int32)]


class BatchRows(C.Structure):
    """Mirror of ``sv_batch_rows``."""

    _fields_ = [("x_rows", C.c_void_p), ("n_rows", C.c_int64), ("row_index", C.c_void_p), ("n_index", C.c_int64),
                ("counter", C.c_void_p), ("counter_base", C.c_void_p), ("batch_stride", C.c_int64),
                ("x_batch", C.c_void_p)]


class WgradBatchJob(C.Structure):
    """Mirror of ``sv_wgrad_batch_job``."""

    _fields_ = [("g", ConvGeom), ("x", C.c_void_p), ("dy", C.c_void_p), ("y", C.c_void_p), ("idx", C.c_void_p),
                ("dw", C.c_void_p), ("db", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_uint64),
                ("act", C.c_int32), ("alpha", C.c_float), ("accumulate", C.c_int32), ("blocks", C.c_int32)]


class DgradJob(C.Structure):
    """Mirror of ``sv_dgrad_job``."""

    _fields_ = [("g", ConvGeom), ("w", C.c_void_p), ("dy", C.c_void_p), ("y", C.c_void_p), ("idx", C.c_void_p),
                ("dx", C.c_void_p), ("act", C.c_int32), ("alpha", C.c_float)]


_p, _i, _f, _i64, _u64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_uint64, C.c_size_t
_g = C.POINTER(ConvGeom)

# name -> (restype, argtypes); one entry per declaration in include/spectrum_hip.h
SIGNATURES = {
    "sv_last_error": (C.c_char_p, []),
    "sv_version": (_i, []),
    "sv_set_option": (_i, [C.c_char_p, _i]),
    "sv_get_option": (_i, [C.c_char_p, C.POINTER(C.c_int)]),
    "sv_conv2d_fwd": (_i, [_g, _p, _p, _p, _p, _i, _f, _i, _p]),
    "sv_conv2d_dgrad": (_i, [_g, _p, _p, _p, _i, _p]),
    "sv_conv2d_wgrad_workspace_bytes": (_sz, [_g]),
    "sv_conv2d_wgrad": (_i, [_g, _p, _p, _p, _i, _p, _sz, _i, _p]),
    "sv_static_scratch_bytes": (_sz, []),
    "sv_conv2d_wgrad_static": (_i, [_g, _p, _p, _p, _i, _i, _p]),
    "sv_conv2d_fused_supported": (_i, [_g, _i]),
    "sv_conv2d_s1_mfma_bwd": (_i, [_g]),
    "sv_conv2d_pool_fwd": (_i, [_g, _p, _p, _p, _p, _p, _i, _f, _i, _p]),
    "sv_conv2d_bwd": (_i, [_g, _p, _p, _p, _p, _p, _p, _p, _p, _i, _f, _p, _sz, _i, _p]),
    "sv_convnet_supported": (_i, [C.POINTER(NetLayer), _i, _i, _i, _i]),
    "sv_convnet_fwd": (_i, [C.POINTER(NetLayer), _i, _i, _i, _i, _p, _p, _i, _p]),
    "sv_convnet_fwd_mse": (_i, [C.POINTER(NetLayer), _i, _i, _i, _i, _p, _p, _p, _p, _f, _p, C.POINTER(C.c_int), _i, _p]),
    "sv_convnet_rows_supported": (_i, [C.POINTER(NetLayer), _i, _i, _i, _i]),
    "sv_convnet_fwd_mse_rows": (_i, [C.POINTER(NetLayer), _i, _i, _i, _i, C.POINTER(BatchRows), _p, _p, _p, _f, _p,
                                     C.POINTER(C.c_int), _i, _p]),
    "sv_conv2d_bwd_partial": (_i, [_g, _p, _p, _p, _p, _i, _f, _i, _p, _sz, _i, _p]),
    "sv_conv2d_wgrad_finish": (_i, [C.POINTER(WgradJob), _i, _i, _p]),
    "sv_conv2d_wgrad_batch": (_i, [C.POINTER(WgradBatchJob), _i, _i, _i, _p]),
    "sv_convnet_fwd_mse_dgrad_supported": (_i, [C.POINTER(NetLayer), _i, _i, _i, _i, C.POINTER(DgradJob), _i]),
    "sv_convnet_fwd_mse_dgrad": (_i, [C.POINTER(NetLayer), _i, _i, _i, _i, _p, C.POINTER(BatchRows), _p, _p, _p, _f, _p,
                                 C.POINTER(C.c_int), C.POINTER(DgradJob), _i, _i, _p]),
    "sv_conv2d_dgrad_chain_supported": (_i, [C.POINTER(DgradJob), _i]),
    "sv_conv2d_dgrad_chain": (_i, [C.POINTER(DgradJob), _i, _i, _p]),
    "sv_conv2d_wgrad_batch_slabs_supported": (_i, [C.POINTER(WgradBatchJob), _i]),
    "sv_conv2d_wgrad_batch_slabs": (_i, [C.POINTER(WgradBatchJob), _i, C.POINTER(GradSlabs), _i, _p]),
    "sv_conv2d_transpose_fwd": (_i, [_g, _p, _p, _p, _p, _i, _f, _i, _p]),
    "sv_conv2d_transpose_dgrad": (_i, [_g, _p, _p, _p, _i, _p]),
    "sv_conv2d_transpose_wgrad": (_i, [_g, _p, _p, _p, _i, _p, _sz, _i, _p]),
    "sv_conv2d_scratch_bytes": (_sz, [_g, _i]),
    "sv_conv2d_fwd_ws": (_i, [_g, _p, _p, _p, _p, _i, _f, _p, _sz, _i, _p]),
    "sv_conv2d_dgrad_ws": (_i, [_g, _p, _p, _p, _p, _sz, _i, _p]),
    "sv_conv2d_transpose_fwd_ws": (_i, [_g, _p, _p, _p, _p, _i, _f, _p, _sz, _i, _p]),
    "sv_conv2d_transpose_dgrad_ws": (_i, [_g, _p, _p, _p, _p, _sz, _i, _p]),
    "sv_act_fwd": (_i, [_p, _p, _i64, _i, _f, _i, _p]),
    "sv_colreduce_workspace_bytes": (_sz, [_i64, _i]),
    "sv_act_bwd": (_i, [_p, _p, _p, _p, _i, _i64, _i, _i, _f, _p, _sz, _i, _p]),
    "sv_bias_grad": (_i, [_p, _p, _i, _i64, _i, _p, _sz, _i, _p]),
    "sv_maxpool2d_fwd": (_i, [_p, _p, _p] + [_i] * 11 + [_p]),
    "sv_maxpool2d_bwd": (_i, [_p, _p, _p] + [_i] * 11 + [_p]),
    "sv_upsample2d_fwd": (_i, [_p, _p] + [_i] * 7 + [_p]),
    "sv_upsample2d_bwd": (_i, [_p, _p] + [_i] * 7 + [_p]),
    "sv_batchnorm_fwd_train": (_i, [_p] * 8 + [_i64, _i, _f, _f, _i, _p, _sz, _i, _p]),
    "sv_batchnorm_fwd_infer": (_i, [_p] * 6 + [_i64, _i, _f, _i, _p]),
    "sv_batchnorm_bwd": (_i, [_p] * 8 + [_i, _i64, _i, _p, _sz, _i, _p]),
    "sv_batchnorm_workspace_bytes": (_sz, [_i64, _i, _i]),
    "sv_batchnorm_act_fwd_train": (_i, [_p] * 8 + [_i64, _i, _f, _f, _i, _i, _i, _f, _p, _sz, _i, _p]),
    "sv_batchnorm_act_fwd_infer": (_i, [_p] * 6 + [_i64, _i, _f, _i, _f, _i, _p]),
    "sv_batchnorm_sync_moments": (_i, [_p, _p, _i64, _i, _i, _p, _sz, _i, _p]),
    "sv_batchnorm_sync_fwd_apply": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _f, _i, _p]),
    "sv_batchnorm_sync_bwd_sums": (_i, [_p, _p, _p, _i, _f, _p, _p, _p, _p, _p, _i, _i64, _i, _i, _p, _sz, _i, _p]),
    "sv_batchnorm_sync_bwd_apply": (_i, [_p, _p, _p, _i, _f, _p, _p, _p, _p, _i64, _p, _i, _f, _p, _i, _i64, _i, _i, _p,
                                         _sz, _i, _p]),
    "sv_batchnorm_act_bwd": (_i, [_p, _p, _p, _i, _f, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _i, _i64, _i, _i, _p, _sz,
                                  _i, _p]),
    "sv_concat_channels_fwd": (_i, [_p, _p, _p, _i64, _i, _i, _i, _p]),
    "sv_concat_channels_bwd": (_i, [_p, _p, _p, _i64, _i, _i, _i, _p]),
    "sv_dense_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "sv_dense_dgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "sv_dense_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "sv_dense_wgrad_workspace_bytes": (_sz, [_i, _i, _i]),
    "sv_dense_wgrad_ws": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _sz, _i, _p]),
    "sv_dropout_fwd": (_i, [_p, _p, _i64, _f, _u64, _u64, _i, _p]),
    "sv_dropout_bwd": (_i, [_p, _p, _i64, _f, _u64, _u64, _i, _p]),
    "sv_reduce_sum_h_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "sv_reduce_sum_h_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "sv_reparam_kl_fwd": (_i, [_p] * 7 + [_i, _i, _u64, _u64, _p, _p]),
    "sv_reparam_kl_bwd": (_i, [_p] * 6 + [_f, _i, _i, _p]),
    "sv_loss_workspace_bytes": (_sz, []),
    "sv_loss_mse": (_i, [_p, _p, _p, _i, _f, _p, _i, _p, _f, _i64, _i, _p, _p, _f, _i, _p]),
    "sv_loss_mse_map": (_i, [_p, _p, _p, _i, _f, _p, _i64, _i, _i, _p]),
    "sv_sample_mse": (_i, [_p, _p, _p, _i64, _i64, _i, _p]),
    "sv_roc_auc_counts": (_i, [_p, _i64, _p, _i64, _p, _p]),
    "sv_loss_mae": (_i, [_p, _p, _p, _i, _p, _f, _i64, _p, _i, _p]),
    "sv_loss_bce": (_i, [_p, _f, _i, _p, _i, _p, _f, _i64, _p, _i, _p]),
    "sv_adam_keras_step": (_i, [_p, _p, _p, _p, _i64, _p, _f, _f, _f, _f, _i, _f, _i, _f, _i, _p, _p, _i, _f, _p, _p, _f,
                                _p]),
    "sv_adam_keras_step_slabs": (_i, [_p, _p, _p, _p, _i64, _p, _f, _f, _f, _f, _i, _f, _i, _f, _i, _p, _p, _i, _f, _p, _p,
                                      _f, C.POINTER(GradSlabs), _i, _p]),
    "sv_deferred_reduce_begin": (_i, []),
    "sv_deferred_reduce_flush": (_i, [_i, _p]),
    "sv_fold_scalar": (_i, [_p, _i, _f, _p, _i, _p, _f, _p]),
    "sv_scalar_program": (_i, [_i, _p, _p, _p, _p, _p, _p]),
    "sv_adam_keras_prepare": (_i, [_p, _f, _f, _f, _i, _f, _i, _p]),
    "sv_adam_keras_apply": (_i, [_p, _p, _p, _p, _i64, _p, _f, _f, _f, _f, _i, _p]),
    "sv_gather_rows": (_i, [_p, _i, _p, _p, _i64, _i64, _p]),
    "sv_conv2d_masked_supported": (_i, [_g]),
    "sv_conv2d_fwd_masked": (_i, [_g, _p, _p, _p, _p, _p, _i, _f, _p, _sz, _i, _p]),
    "sv_conv2d_wgrad_masked": (_i, [_g, _p, _p, _p, _p, _i, _p, _sz, _i, _p]),
    "sv_mask_input": (_i, [_p, _p, _p, _i64, _p]),
    "sv_random_block_mask": (_i, [_p, _i, _i, _i, _i, C.c_double, _u64, _u64, _p]),
    "sv_mask_input_dt": (_i, [_p, _p, _p, _i64, _i, _p]),
    "sv_axpy": (_i, [_p, _p, _f, _i64, _i, _p]),
    "sv_fill_f32": (_i, [_p, _f, _i64, _p]),
    "sv_process_regions": (_i, [_p, _p, _p, _i64, _i, _i, _i, _i, _i64, _p, _p, _p, _p]),
    "sv_sort_min_diff": (_i, [_p, _i64, _i, _i, _i, _i, _p, _p]),
    "sv_reorder_rows": (_i, [_p, _p, _p, _i64, _i, _i, _p]),
    "sv_axpy_f32": (_i, [_p, _p, _f, _i64, _p]),
    "sv_cast_f32_bf16": (_i, [_p, _p, _i64, _p]),
    "sv_cast_bf16_f32": (_i, [_p, _p, _i64, _p]),
}


class SpectrumHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback backend.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover - build/packaging error
            raise ImportError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error() -> str:
    return lib.sv_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = ""):
    if rc != 0:
        raise SpectrumHipError(f"{what or 'libspectrum_hip'} failed (code {rc}): {last_error()}")
