"""ctypes binding of libwdf_hip.so (C ABI: include/wdf_hip.h).

PyTorch is used here only as plumbing: it owns device memory (tensor.data_ptr()) and the
HIP stream the kernels are enqueued on.  There is NO fallback: if the HIP library is
missing or a call fails, this raises -- the product never computes on the CPU.
"""
import ctypes as C
import os

import numpy as np

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# WDF_HIP_LIB: another build of the same library (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("WDF_HIP_LIB") or os.path.join(_HERE, "libwdf_hip.so")

WDF_X_TIME_MAJOR = 1 << 0
WDF_PREC_F64 = 1 << 1
WDF_GENERAL_ROOT = 1 << 3
WDF_MLP_LANE_PER_SEQUENCE = 1 << 4
WDF_ONE_SEQUENCE_PER_LANE = 1 << 5
WDF_R_PER_SEQUENCE = 1 << 6

# Set True to run the one-lane-per-sequence MLP kernels (csrc/wdf_mlp.h) instead of the default
# 16-lane row per sequence (csrc/wdf_mlp_row.h): parity tests and A/B timing.
MLP_LANE_PER_SEQUENCE = False

# Set True to run the one-pass step with one sequence per lane even for even batches (WDF_ONE_SEQUENCE_PER_LANE)
ONE_SEQUENCE_PER_LANE = os.environ.get("WDF_ONE_SEQUENCE_PER_LANE", "") not in ("", "0")     # (env: A/B runs of bench.py)

# Set True to make every clipper call take the general per-step root evaluation (WDF_GENERAL_ROOT):
# parity tests compare it with the default path, tools time one against the other on the same box.
GENERAL_ROOT = False


ABI_VERSION = 6                # include/wdf_hip.h WDF_HIP_ABI_VERSION


def _root_flag():
    return WDF_GENERAL_ROOT if GENERAL_ROOT else 0


class WdfHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libwdf_hip.so (once).  Raises WdfHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WdfHipError(
            f"{LIB_PATH} not found: build it with `make -C differentiable-wdfs_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, fp, i64, ci, cf = C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float
    L.wdf_abi_version.restype = ci
    if L.wdf_abi_version() != ABI_VERSION:
        raise WdfHipError(f"{LIB_PATH} has ABI version {L.wdf_abi_version()}, this host layer needs {ABI_VERSION}: "
                          "rebuild it (make -C differentiable-wdfs_amd/csrc)")
    L.wdf_last_error.restype = C.c_char_p
    L.wdf_device_info.restype = ci
    L.wdf_device_info.argtypes = [ci, C.c_char_p, ci]
    L.wdf_clipper_fwd.restype = ci
    L.wdf_clipper_fwd.argtypes = [fp, fp, fp, cf, ci, ci, fp, fp, fp, fp, i64, i64, ci, vp]
    L.wdf_clipper_bwd.restype = ci
    L.wdf_clipper_bwd.argtypes = [fp, fp, fp, cf, ci, ci, fp, fp, vp, fp, fp, fp, ci, i64, i64, ci, vp]
    L.wdf_clipper_bwd_ws_bytes.restype = C.c_size_t
    L.wdf_clipper_bwd_ws_bytes.argtypes = [i64]
    L.wdf_clipper_tp_chunks.restype = ci
    L.wdf_clipper_tp_chunks.argtypes = [i64, ci]
    L.wdf_clipper_tp_warm_unit.restype = ci
    L.wdf_clipper_tp_warm_unit.argtypes = []
    L.wdf_clipper_fwd_tp_ws_bytes.restype = C.c_size_t
    L.wdf_clipper_fwd_tp_ws_bytes.argtypes = [i64, ci]
    L.wdf_clipper_fwd_tp.restype = ci
    L.wdf_clipper_fwd_tp.argtypes = [fp, fp, fp, cf, ci, ci, fp, fp, fp, fp, i64, i64, ci, ci, cf, vp, vp, ci, vp]
    L.wdf_clipper_fwd_tp_state_bytes.restype = C.c_size_t
    L.wdf_clipper_fwd_tp_state_bytes.argtypes = [i64, ci, ci]
    L.wdf_clipper_fwd_tp_state_reset.restype = ci
    L.wdf_clipper_fwd_tp_state_reset.argtypes = [vp, i64, ci, vp]
    L.wdf_clipper_fwd_tp_warm.restype = ci
    L.wdf_clipper_fwd_tp_warm.argtypes = [fp, fp, fp, cf, ci, ci, fp, fp, fp, fp, i64, i64, ci, ci, cf, vp, vp, vp, ci, ci, vp]
    L.wdf_clipper_bwd_tp_ws_bytes.restype = C.c_size_t
    L.wdf_clipper_bwd_tp_ws_bytes.argtypes = [i64, ci]
    L.wdf_clipper_bwd_tp_ws_init.restype = ci
    L.wdf_clipper_bwd_tp_ws_init.argtypes = [vp, i64, ci, vp]
    L.wdf_clipper_bwd_tp.restype = ci
    L.wdf_clipper_bwd_tp.argtypes = [fp, fp, fp, cf, ci, ci, fp, fp, vp, fp, fp, ci, i64, i64, ci, ci, vp]
    L.wdf_clipper_bwd_mse_tp.restype = ci
    L.wdf_clipper_bwd_mse_tp.argtypes = [fp, fp, fp, cf, ci, ci, fp, fp, fp, cf, vp, fp, fp, fp, ci, i64, i64, ci,
                                         ci, vp]
    L.wdf_clipper_bwd_mse_tp_adam.restype = ci
    L.wdf_clipper_bwd_mse_tp_adam.argtypes = [fp, fp, fp, cf, ci, ci, fp, fp, fp, cf, vp, fp, fp, i64, i64, ci, ci,
                                              fp, fp, vp, fp, cf, cf, cf, fp, fp, vp]
    L.wdf_clipper_step_mse_tp_ws_bytes.restype = C.c_size_t
    L.wdf_clipper_step_mse_tp_ws_bytes.argtypes = [i64, ci]
    L.wdf_clipper_step_mse_tp_ws_init.restype = ci
    L.wdf_clipper_step_mse_tp_ws_init.argtypes = [vp, i64, ci, vp]
    L.wdf_clipper_step_mse_tp.restype = ci
    L.wdf_clipper_step_mse_tp.argtypes = [fp, fp, fp, cf, ci, ci, fp, cf, i64, fp, fp, fp, i64, i64, ci, ci, cf, vp, vp, vp,
                                          ci, fp, fp, ci, fp, fp, vp, fp, cf, cf, cf, fp, fp, ci, vp]
    L.wdf_clipper_step_esr_tp.restype = ci
    L.wdf_clipper_step_esr_tp.argtypes = [fp, fp, fp, cf, ci, ci, fp, C.c_double, C.c_double, i64, fp, fp, fp, i64, i64, ci, ci, cf,
                                          vp, vp, vp, ci, fp, fp, fp, fp, fp, vp, fp, cf, cf, cf, fp, fp, ci, vp]
    L.wdf_esr_finish.restype = ci
    L.wdf_esr_finish.argtypes = [fp, C.c_double, C.c_double, fp, fp, vp]
    L.wdf_loss_sums_ws_bytes.restype = i64
    L.wdf_loss_sums.restype = ci
    L.wdf_loss_sums.argtypes = [fp, fp, i64, i64, i64, vp, vp, vp]
    L.wdf_esr_coef.restype = ci
    L.wdf_esr_coef.argtypes = [vp, C.c_double, C.c_double, fp, fp, vp]
    L.wdf_loss_esr_grad.restype = ci
    L.wdf_loss_esr_grad.argtypes = [fp, fp, fp, i64, i64, i64, fp, vp]
    L.wdf_clipper_bwd_esr_tp.restype = ci
    L.wdf_clipper_bwd_esr_tp.argtypes = [fp, fp, fp, cf, ci, ci, fp, fp, fp, fp, i64, vp, fp, fp, fp, ci, i64, i64, ci,
                                         ci, vp]
    L.wdf_clipper_asym_fwd.restype = ci
    L.wdf_clipper_asym_fwd.argtypes = [fp, fp, cf, ci, C.c_double, ci, fp, fp, fp, fp, vp, i64, i64, vp]
    L.wdf_clipper_asym_fwd_tp_ws_bytes.restype = C.c_size_t
    L.wdf_clipper_asym_fwd_tp_ws_bytes.argtypes = [i64, ci]
    L.wdf_clipper_asym_fwd_tp.restype = ci
    L.wdf_clipper_asym_fwd_tp.argtypes = [fp, fp, cf, ci, C.c_double, ci, fp, fp, fp, fp, i64, i64, ci, ci, cf, vp, vp, vp]
    L.wdf_clipper_asym_bwd_ws_bytes.restype = C.c_size_t
    L.wdf_clipper_asym_bwd_ws_bytes.argtypes = [i64]
    L.wdf_clipper_asym_bwd.restype = ci
    L.wdf_clipper_asym_bwd.argtypes = [fp, fp, cf, C.c_double, ci, fp, fp, vp, fp, i64, i64, vp]
    L.wdf_clipper_asym_bwd_tp_ws_bytes.restype = C.c_size_t
    L.wdf_clipper_asym_bwd_tp_ws_bytes.argtypes = [i64, ci]
    L.wdf_clipper_asym_bwd_tp.restype = ci
    L.wdf_clipper_asym_bwd_tp.argtypes = [fp, fp, cf, ci, fp, fp, fp, fp, vp, fp, fp, i64, i64, ci, vp]
    L.wdf_ss_dyn_row_len.restype = ci
    L.wdf_ss_dyn_row_len.argtypes = [ci, ci]
    L.wdf_ss_dyn_fwd.restype = ci
    L.wdf_ss_dyn_fwd.argtypes = [fp, fp, ci, ci, ci, ci, fp, fp, ci, ci, ci, ci, fp, fp, fp, fp, i64, i64, vp]
    L.wdf_ss_dyn_bwd_ws_bytes.restype = C.c_size_t
    L.wdf_ss_dyn_bwd_ws_bytes.argtypes = [i64]
    L.wdf_ss_dyn_bwd.restype = ci
    L.wdf_ss_dyn_bwd.argtypes = [fp, fp, ci, ci, ci, ci, fp, fp, ci, ci, ci, ci, fp, fp, fp, vp, fp, fp, fp, fp, i64, i64, vp]
    L.wdf_ss_dyn_fwd_tp_ws_bytes.restype = C.c_size_t
    L.wdf_ss_dyn_fwd_tp_ws_bytes.argtypes = [ci, i64, ci]
    L.wdf_ss_dyn_fwd_tp.restype = ci
    L.wdf_ss_dyn_fwd_tp.argtypes = [fp, fp, ci, ci, ci, ci, fp, fp, ci, ci, ci, ci, fp, fp, fp, fp, i64, i64, ci, ci, cf, fp, vp, vp, vp]
    L.wdf_ss_dyn_bwd_tp_ws_bytes.restype = C.c_size_t
    L.wdf_ss_dyn_bwd_tp_ws_bytes.argtypes = [ci, i64, i64, ci]
    L.wdf_ss_dyn_bwd_tp.restype = ci
    L.wdf_ss_dyn_bwd_tp.argtypes = [fp, fp, ci, ci, ci, ci, fp, fp, ci, ci, ci, ci, fp, fp, fp, vp, fp, fp, fp, fp, i64, i64, ci, vp]
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    L.wdf_ss_dyn_rows.restype = ci
    L.wdf_ss_dyn_rows.argtypes = [ip, ci, dp, ci, ip, ci, vp, ci, ci, fp, fp, i64, i64, vp]
    L.wdf_ss_dyn_rows_bwd_ws_bytes.restype = C.c_size_t
    L.wdf_ss_dyn_rows_bwd_ws_bytes.argtypes = [ci, i64, i64]
    L.wdf_ss_dyn_rows_bwd.restype = ci
    L.wdf_ss_dyn_rows_bwd.argtypes = [ip, ci, dp, ci, ip, ci, vp, ci, ci, fp, fp, vp, vp, i64, i64, vp]
    L.wdf_clipper_mlp_wgrad_matrix_core_chunks.restype = ci
    L.wdf_clipper_mlp_wgrad_matrix_core_chunks.argtypes = [i64, i64]
    L.wdf_asym_root.restype = ci
    L.wdf_asym_root.argtypes = [fp, fp, cf, ci, C.c_double, ci, vp, i64, vp]
    L.wdf_mlp_weight_count.restype = ci
    L.wdf_mlp_weight_count.argtypes = [ci, ci]
    L.wdf_clipper_mlp_fwd.restype = ci
    L.wdf_clipper_mlp_fwd.argtypes = [fp, fp, fp, fp, ci, ci, cf, fp, fp, fp, fp, i64, i64, ci, vp]
    L.wdf_clipper_mlp_bwd_w_ws_bytes.restype = i64
    L.wdf_clipper_mlp_bwd_w_ws_bytes.argtypes = [ci, ci, i64]
    L.wdf_clipper_mlp_bwd_w.restype = ci
    L.wdf_clipper_mlp_bwd_w.argtypes = [fp, fp, fp, fp, ci, ci, cf, fp, fp, vp, fp, fp, i64, i64, ci, vp]
    L.wdf_clipper_mlp_tp_chunks.restype = ci
    L.wdf_clipper_mlp_tp_chunks.argtypes = [i64, ci]
    L.wdf_clipper_mlp_fwd_tp_ws_bytes.restype = C.c_size_t
    L.wdf_clipper_mlp_fwd_tp_ws_bytes.argtypes = [i64, ci]
    L.wdf_clipper_mlp_fwd_tp.restype = ci
    L.wdf_clipper_mlp_fwd_tp.argtypes = [fp, fp, fp, fp, ci, ci, cf, fp, fp, fp, fp, i64, i64, ci, ci, vp, cf, vp, vp, vp]
    L.wdf_clipper_mlp_bwd_w_tp_ws_bytes.restype = i64
    L.wdf_clipper_mlp_bwd_w_tp_ws_bytes.argtypes = [ci, ci, i64, i64, ci]
    L.wdf_clipper_mlp_bwd_w_tp.restype = ci
    L.wdf_clipper_mlp_bwd_w_tp.argtypes = [fp, fp, fp, fp, ci, ci, cf, fp, fp, vp, fp, fp, i64, i64, ci, vp]
    L.wdf_clipper_mlp_fwd_tp_kappa.restype = ci
    L.wdf_clipper_mlp_fwd_tp_kappa.argtypes = [fp, fp, fp, fp, ci, ci, cf, fp, fp, fp, fp, fp, fp, i64, i64, ci, ci, vp, cf, vp, vp, vp]
    L.wdf_clipper_mlp_tp_starts.restype = ci
    L.wdf_clipper_mlp_tp_starts.argtypes = [i64, ci, ci, C.POINTER(C.c_int64)]
    L.wdf_clipper_mlp_bwd_w_tp_kappa.restype = ci
    L.wdf_clipper_mlp_bwd_w_tp_kappa.argtypes = [fp, fp, fp, fp, ci, ci, cf, fp, fp, fp, vp, fp, fp, i64, i64, ci, vp]
    L.wdf_clipper_mlp_bwd_ws_bytes.restype = C.c_size_t
    L.wdf_clipper_mlp_bwd_ws_bytes.argtypes = [i64]
    L.wdf_clipper_mlp_bwd.restype = ci
    L.wdf_clipper_mlp_bwd.argtypes = [fp, fp, fp, fp, ci, ci, cf, fp, fp, fp, fp, fp, vp, fp, i64, i64, ci, vp]
    L.wdf_mlp_eval.restype = ci
    L.wdf_mlp_eval.argtypes = [fp, fp, fp, ci, ci, fp, i64, vp]
    L.wdf_mlp_fit_epoch.restype = ci
    L.wdf_mlp_fit_epoch.argtypes = [fp, fp, fp, i64, ci, fp, fp, fp, vp, cf, cf, cf, cf, cf, cf, vp, ci, ci, vp]
    L.wdf_clipper_mlp_wgrad_ws_bytes.restype = i64
    L.wdf_clipper_mlp_wgrad_ws_bytes.argtypes = [ci, ci, i64]
    L.wdf_clipper_mlp_wgrad.restype = ci
    L.wdf_clipper_mlp_wgrad.argtypes = [fp, fp, fp, fp, fp, ci, ci, cf, vp, fp, i64, vp]
    i32p = C.POINTER(C.c_int32)
    L.wdf_clipper_mlp_step_state_bytes.restype = C.c_size_t
    L.wdf_clipper_mlp_step_state_bytes.argtypes = [ci, ci, i64, i64, ci, ci]
    L.wdf_clipper_mlp_step_plan.restype = ci
    L.wdf_clipper_mlp_step_plan.argtypes = [vp, ci, ci, i64, i64, ci, ci, i32p, ci, ci, ci, ci, ci, cf, vp]
    L.wdf_clipper_mlp_step_read.restype = ci
    L.wdf_clipper_mlp_step_read.argtypes = [vp, ci, ci, i64, i64, ci, ci, i32p, i32p, i32p, i32p, C.POINTER(C.c_float), vp]
    L.wdf_clipper_mlp_step_set.restype = ci
    L.wdf_clipper_mlp_step_set.argtypes = [vp, ci, C.c_int32, vp]
    L.wdf_clipper_mlp_step_set_wcol.restype = ci
    L.wdf_clipper_mlp_step_set_wcol.argtypes = [vp, ci, ci, i64, i64, ci, ci, i32p, vp]
    L.wdf_clipper_mlp_step_prepare.restype = ci
    L.wdf_clipper_mlp_step_prepare.argtypes = [fp, fp, cf, i64, i64, fp, fp, vp]
    L.wdf_clipper_mlp_step.restype = ci
    L.wdf_clipper_mlp_step.argtypes = [fp, fp, fp, fp, fp, ci, ci, ci, cf, fp, i64, C.c_double, C.c_double, fp, fp, fp, vp, i64, i64,
                                       ci, ci, ci, vp, fp, fp, fp, fp, fp, vp, fp, cf, cf, cf, vp]
    L.wdf_ss_probe.restype = ci
    L.wdf_ss_probe.argtypes = [vp, ci, vp, fp, ci, vp, ci, fp, vp, vp, vp]
    L.wdf_ss_probe_adam.restype = ci
    L.wdf_ss_probe_adam.argtypes = [vp, ci, vp, ci, vp, fp, ci, vp, ci, fp, vp, vp, vp]
    L.wdf_ss_lin_step_ws_bytes.restype = C.c_size_t
    L.wdf_ss_lin_step_ws_bytes.argtypes = [ci, ci, i64, i64, ci]
    L.wdf_ss_lin_step_mse.restype = ci
    L.wdf_ss_lin_step_mse.argtypes = [fp, fp, vp, ci, ci, ci, fp, cf, fp, vp, fp, fp, fp, i64, i64, ci, fp, fp, vp]
    L.wdf_ss_nl_step_ws_bytes.restype = C.c_size_t
    L.wdf_ss_nl_step_ws_bytes.argtypes = [ci, ci, i64, i64, ci]
    L.wdf_ss_nl_step_chunk_len.restype = ci
    L.wdf_ss_nl_step_chunk_len.argtypes = [i64, ci]
    L.wdf_ss_nl_step_plan.restype = ci
    L.wdf_ss_nl_step_plan.argtypes = [vp, ci, ci, i64, i64, ci, ci, ci, ci, ci, cf, vp]
    L.wdf_ss_nl_step_set.restype = ci
    L.wdf_ss_nl_step_set.argtypes = [vp, ci, C.c_double, vp]
    L.wdf_ss_nl_step_read.restype = ci
    L.wdf_ss_nl_step_read.argtypes = [vp, vp, vp]
    L.wdf_ss_nl_step_mse.restype = ci
    L.wdf_ss_nl_step_mse.argtypes = [fp, fp, fp, vp, ci, ci, ci, ci, ci, fp, cf, fp, vp, fp, fp, i64, i64, ci, vp]
    L.wdf_ss_ncoef.restype = ci
    L.wdf_ss_ncoef.argtypes = [ci, ci]
    L.wdf_ss_fwd.restype = ci
    L.wdf_ss_fwd.argtypes = [fp, fp, fp, ci, ci, ci, ci, ci, fp, fp, fp, fp, i64, i64, ci, vp]
    L.wdf_ss_fwd_lin_tp_ws_bytes.restype = C.c_size_t
    L.wdf_ss_fwd_lin_tp_ws_bytes.argtypes = [ci, i64, ci]
    L.wdf_ss_fwd_lin_tp.restype = ci
    L.wdf_ss_fwd_lin_tp.argtypes = [fp, fp, ci, ci, fp, fp, fp, fp, i64, i64, ci, vp, vp]
    L.wdf_ss_bwd.restype = ci
    L.wdf_ss_bwd.argtypes = [fp, fp, fp, ci, ci, ci, ci, ci, fp, fp, vp, fp, fp, fp, i64, i64, ci, vp]
    L.wdf_ss_tp_chunks.restype = ci
    L.wdf_ss_tp_chunks.argtypes = [i64, ci]
    L.wdf_ss_fwd_tp_ws_bytes.restype = C.c_size_t
    L.wdf_ss_fwd_tp_ws_bytes.argtypes = [ci, i64, ci]
    L.wdf_ss_fwd_tp.restype = ci
    L.wdf_ss_fwd_tp.argtypes = [fp, fp, fp, ci, ci, ci, ci, fp, fp, fp, fp, i64, i64, ci, ci, cf, fp, vp, vp, vp]
    L.wdf_ss_tp_starts.restype = ci
    L.wdf_ss_tp_starts.argtypes = [i64, ci, ci, C.POINTER(C.c_int64)]
    L.wdf_ss_bwd_tp_ws_bytes.restype = C.c_size_t
    L.wdf_ss_bwd_tp_ws_bytes.argtypes = [ci, ci, i64, ci]
    L.wdf_ss_bwd_tp.restype = ci
    L.wdf_ss_bwd_tp.argtypes = [fp, fp, fp, ci, ci, ci, ci, ci, fp, fp, vp, fp, fp, fp, i64, i64, ci, vp]
    L.wdf_ss_bwd_ws_bytes.restype = C.c_size_t
    L.wdf_ss_bwd_ws_bytes.argtypes = [ci, ci, i64]
    L.wdf_omega_f32.restype = ci
    L.wdf_omega_f32.argtypes = [fp, fp, vp, i64, vp]
    L.wdf_omega_f64.restype = ci
    L.wdf_omega_f64.argtypes = [vp, vp, vp, i64, vp]
    L.wdf_diode_pair_f32.restype = ci
    L.wdf_diode_pair_f32.argtypes = [fp, fp, cf, cf, ci, ci, fp, i64, vp]
    L.wdf_adam_step.restype = ci
    L.wdf_adam_step.argtypes = [fp, fp, fp, fp, vp, fp, cf, cf, cf, fp, fp, ci, vp]
    L.wdf_adam_step_multi.restype = ci
    L.wdf_adam_step_multi.argtypes = [vp, ci, vp]
    L.wdf_event_create.restype = vp
    L.wdf_event_record.restype = ci
    L.wdf_event_record.argtypes = [vp, vp]
    L.wdf_event_elapsed_ms.restype = ci
    L.wdf_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
    L.wdf_event_destroy.restype = None
    L.wdf_event_destroy.argtypes = [vp]
    L.wdf_clock_stamp.restype = ci
    L.wdf_clock_stamp.argtypes = [vp, vp]
    L.wdf_event_bracket_next.restype = None
    L.wdf_event_bracket_next.argtypes = [vp, vp]
    _lib = L
    return L


EXPORTED_SYMBOLS = (
    "wdf_abi_version", "wdf_last_error", "wdf_device_info",
    "wdf_clipper_fwd", "wdf_clipper_bwd", "wdf_clipper_bwd_ws_bytes",
    "wdf_clipper_tp_chunks", "wdf_clipper_tp_warm_unit", "wdf_clipper_fwd_tp_ws_bytes", "wdf_clipper_fwd_tp",
    "wdf_clipper_fwd_tp_state_bytes", "wdf_clipper_fwd_tp_state_reset", "wdf_clipper_fwd_tp_warm",
    "wdf_clipper_bwd_tp_ws_bytes", "wdf_clipper_bwd_tp_ws_init", "wdf_clipper_bwd_tp", "wdf_clipper_bwd_mse_tp",
    "wdf_clipper_bwd_mse_tp_adam", "wdf_clipper_step_mse_tp_ws_bytes", "wdf_clipper_step_mse_tp_ws_init",
    "wdf_clipper_step_mse_tp", "wdf_clipper_step_esr_tp", "wdf_esr_finish", "wdf_loss_sums_ws_bytes", "wdf_loss_sums", "wdf_esr_coef", "wdf_loss_esr_grad", "wdf_clipper_bwd_esr_tp",
    "wdf_clipper_asym_fwd", "wdf_clipper_asym_fwd_tp_ws_bytes", "wdf_clipper_asym_fwd_tp", "wdf_clipper_asym_bwd_ws_bytes", "wdf_clipper_asym_bwd",
    "wdf_clipper_asym_bwd_tp_ws_bytes", "wdf_clipper_asym_bwd_tp", "wdf_asym_root",
    "wdf_ss_dyn_row_len", "wdf_ss_dyn_fwd", "wdf_ss_dyn_bwd_ws_bytes", "wdf_ss_dyn_bwd", "wdf_clipper_mlp_wgrad_matrix_core_chunks",
    "wdf_ss_dyn_fwd_tp_ws_bytes", "wdf_ss_dyn_fwd_tp", "wdf_ss_dyn_bwd_tp_ws_bytes", "wdf_ss_dyn_bwd_tp",
    "wdf_ss_dyn_rows", "wdf_ss_dyn_rows_bwd_ws_bytes", "wdf_ss_dyn_rows_bwd",
    "wdf_mlp_weight_count", "wdf_clipper_mlp_fwd", "wdf_clipper_mlp_bwd", "wdf_clipper_mlp_bwd_ws_bytes",
    "wdf_clipper_mlp_bwd_w_ws_bytes", "wdf_clipper_mlp_bwd_w",
    "wdf_clipper_mlp_tp_chunks", "wdf_clipper_mlp_fwd_tp_ws_bytes", "wdf_clipper_mlp_fwd_tp",
    "wdf_clipper_mlp_bwd_w_tp_ws_bytes", "wdf_clipper_mlp_bwd_w_tp",
    "wdf_clipper_mlp_fwd_tp_kappa", "wdf_clipper_mlp_bwd_w_tp_kappa", "wdf_clipper_mlp_tp_starts",
    "wdf_clipper_mlp_wgrad_ws_bytes", "wdf_clipper_mlp_wgrad", "wdf_mlp_eval", "wdf_mlp_fit_epoch",
    "wdf_clipper_mlp_step_state_bytes", "wdf_clipper_mlp_step_plan", "wdf_clipper_mlp_step_read", "wdf_clipper_mlp_step_set",
    "wdf_clipper_mlp_step_set_wcol", "wdf_clipper_mlp_step_prepare", "wdf_clipper_mlp_step",
    "wdf_ss_probe", "wdf_ss_probe_adam", "wdf_ss_lin_step_ws_bytes", "wdf_ss_lin_step_mse",
    "wdf_ss_nl_step_ws_bytes", "wdf_ss_nl_step_chunk_len", "wdf_ss_nl_step_plan", "wdf_ss_nl_step_set", "wdf_ss_nl_step_read",
    "wdf_ss_nl_step_mse",
    "wdf_ss_ncoef", "wdf_ss_fwd", "wdf_ss_bwd", "wdf_ss_bwd_ws_bytes", "wdf_ss_fwd_lin_tp_ws_bytes", "wdf_ss_fwd_lin_tp",
    "wdf_ss_tp_chunks", "wdf_ss_tp_starts", "wdf_ss_fwd_tp_ws_bytes", "wdf_ss_fwd_tp", "wdf_ss_bwd_tp_ws_bytes", "wdf_ss_bwd_tp",
    "wdf_omega_f32", "wdf_omega_f64", "wdf_diode_pair_f32", "wdf_adam_step", "wdf_adam_step_multi",
    "wdf_event_create", "wdf_event_record", "wdf_event_elapsed_ms", "wdf_event_destroy", "wdf_event_bracket_next",
    "wdf_clock_stamp",
)


def _check(rc, what):
    if rc != 0:
        raise WdfHipError(f"{what} failed (rc={rc}): {lib().wdf_last_error().decode()}")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    # torch's current stream on the current device, as the raw hipStream_t (no Stream object: this runs per launch)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _f32_dev(t, name):
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise WdfHipError(f"{name}: expected a contiguous float32 tensor on the GPU, got "
                          f"{type(t).__name__} {getattr(t, 'dtype', None)} {getattr(t, 'device', None)}")
    return t


def require_gpu():
    if not torch.cuda.is_available():
        raise WdfHipError("no HIP device visible: the WDF engine runs on MI355X only (no CPU fallback)")


def omega64(x, want_iters=False):
    """Wright omega in fp64 on the device (toms917's iteration structure); x: float64 device tensor."""
    require_gpu()
    if not (x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()):
        raise WdfHipError("omega64: expected a contiguous float64 device tensor")
    w = torch.empty_like(x)
    it = torch.empty(x.shape, dtype=torch.int32, device=x.device) if want_iters else None
    _check(lib().wdf_omega_f64(_ptr(x), _ptr(w), _ptr(it), x.numel(), _stream()), "wdf_omega_f64")
    return (w, it) if want_iters else w


def clipper_fwd(x, theta, fs, r=None, n_up=1, n_down=1, want_stash=True, z0=None, want_zT=False,
                time_major=False, fp64=False):
    """x [B,T] (or [T,B] if time_major) -> y [T,B], zstash [T,B] | None, zT [B] | None.
    fp64: tree and root arithmetic in double (WDF_PREC_F64; sequential, the accuracy reference on the device)."""
    require_gpu()
    x = _f32_dev(x, "x")
    r = _f32_dev(r, "r")
    theta = _f32_dev(theta, "theta")
    z0 = _f32_dev(z0, "z0")
    if theta.numel() != 4:
        raise WdfHipError("theta must hold {Is, nVt, R, C}")
    B, T = (x.shape[1], x.shape[0]) if time_major else x.shape
    if r is not None and r.shape != x.shape:
        raise WdfHipError("r must have the shape of x")
    y = torch.empty((T, B), dtype=torch.float32, device=x.device)
    zs = torch.empty((T, B), dtype=torch.float32, device=x.device) if want_stash else None
    zT = torch.empty((B,), dtype=torch.float32, device=x.device) if want_zT else None
    flags = (WDF_X_TIME_MAJOR if time_major else 0) | _root_flag() | (WDF_PREC_F64 if fp64 else 0)
    rc = lib().wdf_clipper_fwd(_ptr(x), _ptr(r), _ptr(theta), float(fs), int(n_up), int(n_down),
                               _ptr(y), _ptr(zs), _ptr(z0), _ptr(zT), B, T, flags, _stream())
    _check(rc, "wdf_clipper_fwd")
    return y, zs, zT


def clipper_bwd(x, theta, fs, zstash, gy, r=None, n_up=1, n_down=1, want_gz0=False, time_major=False,
                gtheta=None, accumulate=False, ws=None, gzT=None, fp64=False):
    """dL/d{Is, nVt, R, C} as a float32[4] device tensor (and dL/dz0 [B] if requested).
    gzT: optional dL/dzT [B] (a loss that also reads the forward's final state).
    fp64: the adjoint in double (WDF_PREC_F64; sequential, the accuracy reference on the device)."""
    require_gpu()
    x = _f32_dev(x, "x")
    r = _f32_dev(r, "r")
    theta = _f32_dev(theta, "theta")
    zstash = _f32_dev(zstash, "zstash")
    gy = _f32_dev(gy, "gy")
    B, T = (x.shape[1], x.shape[0]) if time_major else x.shape
    if tuple(gy.shape) != (T, B) or tuple(zstash.shape) != (T, B):
        raise WdfHipError(f"gy / zstash must be [T,B] = [{T},{B}]")
    if ws is None:
        ws = torch.empty((lib().wdf_clipper_bwd_ws_bytes(B),), dtype=torch.uint8, device=x.device)
    if gtheta is None:
        gtheta = torch.empty((4,), dtype=torch.float32, device=x.device)
        accumulate = False
    gz0 = torch.empty((B,), dtype=torch.float32, device=x.device) if want_gz0 else None
    flags = (WDF_X_TIME_MAJOR if time_major else 0) | _root_flag() | (WDF_PREC_F64 if fp64 else 0)
    rc = lib().wdf_clipper_bwd(_ptr(x), _ptr(r), _ptr(theta), float(fs), int(n_up), int(n_down),
                               _ptr(zstash), _ptr(gy), _ptr(ws), _ptr(gtheta), _ptr(gz0), _ptr(_f32_dev(gzT, "gzT")),
                               1 if accumulate else 0, B, T, flags, _stream())
    _check(rc, "wdf_clipper_bwd")
    return gtheta, gz0


class TpWarmState:
    """Persistent warm-start state of the time-parallel forward for ONE resident input batch
    (include/wdf_hip.h, wdf_clipper_fwd_tp_warm): control block + snapshot ring on the device."""

    def __init__(self, B, T, n_chunks, max_warm_tiles, device, min_warm_tiles=0):
        """max_warm_tiles / min_warm_tiles: in warm-start units of warm_unit() steps (16)."""
        L = lib()
        self.K = L.wdf_clipper_tp_chunks(int(T), int(n_chunks))
        chunk_len = -(-(-(-int(T) // max(1, int(n_chunks)))) // 32) * 32        # as the library rounds it
        # snapshots reach back at most three quarters of a chunk: the one-pass step may shorten the younger chunks by a quarter
        # (skewed spans, csrc/wdf_clipper_fused.h chunk_span) and they must still hold every snapshot
        self.unit = warm_unit()
        self.max_warm_tiles = max(1, min(int(max_warm_tiles), 32, (3 * chunk_len // 4) // self.unit))
        if os.environ.get("WDF_MAX_WARM_TILES"):                # (probing: a shallower snapshot ring, e.g. under WDF_FUSED_SKEW_STEPS)
            self.max_warm_tiles = max(1, min(self.max_warm_tiles, int(os.environ["WDF_MAX_WARM_TILES"])))
        self.B, self.T, self.n_chunks = int(B), int(T), int(n_chunks)
        self.min_warm_tiles = max(0, min(int(min_warm_tiles), self.max_warm_tiles))
        self.buf = torch.empty((L.wdf_clipper_fwd_tp_state_bytes(self.B, self.K, self.max_warm_tiles),),
                               dtype=torch.uint8, device=device)
        self.reset()

    def reset(self):
        _check(lib().wdf_clipper_fwd_tp_state_reset(_ptr(self.buf), self.B, self.min_warm_tiles, _stream()),
               "wdf_clipper_fwd_tp_state_reset")

    def info(self):
        """Host view of the control block (synchronises): snapshot sets available, warm-up tiles the next
        call will run, tiles the last call ran (-1: cold), its largest boundary miss, calls so far."""
        c = self.buf[:96].cpu()
        i, f = c.view(torch.int32), c.view(torch.float32)
        return {"valid": int(i[0]), "next_warm_tiles": int(i[2]), "last_warm_tiles": int(i[3]),
                "last_miss": float(f[12]), "n_calls": int(i[13]), "warm_unit_steps": self.unit, "cold_hold": int(i[20])}


def warm_unit():
    """Steps per warm-start unit ("warm tile") of the time-parallel clipper kernels."""
    return int(lib().wdf_clipper_tp_warm_unit())


def clipper_fwd_tp(x, theta, fs, n_chunks, warmup, tol=1e-6, r=None, n_up=1, n_down=1, want_stash=True, z0=None,
                   want_zT=False, ws=None, status=None, time_major=False, state=None):
    """Time-parallel forward.  Returns y [T,B], zstash | None, zT | None, status (device int32[4]:
    n_bad, max-miss float bits, chunk re-runs, ticket -- read it with tp_status()).
    state: a TpWarmState kept with this input batch -> warm-started chunks (wdf_clipper_fwd_tp_warm)."""
    require_gpu()
    x = _f32_dev(x, "x")
    r = _f32_dev(r, "r")
    theta = _f32_dev(theta, "theta")
    z0 = _f32_dev(z0, "z0")
    B, T = (x.shape[1], x.shape[0]) if time_major else x.shape
    if r is not None and r.shape != x.shape:
        raise WdfHipError("r must have the shape of x")
    y = torch.empty((T, B), dtype=torch.float32, device=x.device)
    zs = torch.empty((T, B), dtype=torch.float32, device=x.device) if want_stash else None
    zT = torch.empty((B,), dtype=torch.float32, device=x.device) if want_zT else None
    if ws is None:
        ws = torch.empty((lib().wdf_clipper_fwd_tp_ws_bytes(B, int(n_chunks)),), dtype=torch.uint8, device=x.device)
    if status is None:
        status = torch.empty((4,), dtype=torch.int32, device=x.device)
    flags = (WDF_X_TIME_MAJOR if time_major else 0) | _root_flag()
    if state is not None:
        if (state.B, state.T, state.n_chunks) != (B, T, int(n_chunks)) or state.buf.device != x.device:
            raise WdfHipError("warm-start state was made for another batch shape / chunking / device")
        rc = lib().wdf_clipper_fwd_tp_warm(_ptr(x), _ptr(r), _ptr(theta), float(fs), int(n_up), int(n_down), _ptr(y),
                                           _ptr(zs), _ptr(z0), _ptr(zT), B, T, int(n_chunks), int(warmup), float(tol),
                                           _ptr(ws), _ptr(status), _ptr(state.buf), state.max_warm_tiles, flags, _stream())
        _check(rc, "wdf_clipper_fwd_tp_warm")
        return y, zs, zT, status
    rc = lib().wdf_clipper_fwd_tp(_ptr(x), _ptr(r), _ptr(theta), float(fs), int(n_up), int(n_down), _ptr(y),
                                  _ptr(zs), _ptr(z0), _ptr(zT), B, T, int(n_chunks), int(warmup), float(tol),
                                  _ptr(ws), _ptr(status), flags, _stream())
    _check(rc, "wdf_clipper_fwd_tp")
    return y, zs, zT, status


def clipper_bwd_mse_tp_adam(x, theta, fs, zstash, zT, target, gscale, n_chunks, opt, r=None, n_up=1, n_down=1,
                            gtheta=None, sse=None, ws=None, time_major=False):
    """The MSE-fused reverse sweep with the Adam update of theta (in place, `opt` = binding.Adam(4, ...))
    folded into its last kernel; -> gtheta[4], sse[1]."""
    require_gpu()
    x, r, theta = _f32_dev(x, "x"), _f32_dev(r, "r"), _f32_dev(theta, "theta")
    zstash, zT, target = _f32_dev(zstash, "zstash"), _f32_dev(zT, "zT"), _f32_dev(target, "target")
    B, T = (x.shape[1], x.shape[0]) if time_major else x.shape
    if opt.n != 4 or theta.numel() != 4:
        raise WdfHipError("clipper_bwd_mse_tp_adam: theta and the optimizer hold {Is, nVt, R, C}")
    if ws is None:
        ws = bwd_tp_workspace(B, int(n_chunks), x.device)
    if gtheta is None:
        gtheta = torch.empty((4,), dtype=torch.float32, device=x.device)
    if sse is None:
        sse = torch.empty((1,), dtype=torch.float32, device=x.device)
    rc = lib().wdf_clipper_bwd_mse_tp_adam(_ptr(x), _ptr(r), _ptr(theta), float(fs), int(n_up), int(n_down), _ptr(zstash),
                                           _ptr(zT), _ptr(target), float(gscale), _ptr(ws), _ptr(gtheta), _ptr(sse), B, T,
                                           int(n_chunks), (WDF_X_TIME_MAJOR if time_major else 0) | _root_flag(),
                                           _ptr(opt.m), _ptr(opt.v), _ptr(opt.step), _ptr(opt.lr), opt.b1, opt.b2, opt.eps,
                                           _ptr(opt.lo), _ptr(opt.hi), _stream())
    _check(rc, "wdf_clipper_bwd_mse_tp_adam")
    return gtheta, sse


_R_SEQ_CACHE = {}       # (data_ptr, shape, strides, version) -> bool
R_PER_SEQUENCE = os.environ.get("WDF_R_PER_SEQUENCE", "1") not in ("", "0")     # (0: always the per-sample evaluation, for A/B runs and tests)


def r_is_per_sequence(r, time_major):
    """Is the resistance channel constant along every sequence (the reference's recordings: one pot value per file,
    dataimport.py:96 -- batch_data cuts the sequences out of it)?  One comparison pass per tensor (storage and version), cached:
    the one-pass step then evaluates calc_impedance once per chunk instead of every step (WDF_R_PER_SEQUENCE)."""
    if r is None or not R_PER_SEQUENCE:
        return False
    key = (r.data_ptr(), tuple(r.shape), tuple(r.stride()), r._version)
    hit = _R_SEQ_CACHE.get(key)
    if hit is None:
        if len(_R_SEQ_CACHE) > 64:
            _R_SEQ_CACHE.clear()
        with torch.no_grad():
            first = r[0:1, :] if time_major else r[:, 0:1]
            hit = _R_SEQ_CACHE[key] = bool((r == first).all())
    return hit


def step_mse_workspace(B, n_chunks, device):
    """Workspace of the one-pass training step with its ticket words cleared: allocate once, reuse."""
    ws = torch.empty((lib().wdf_clipper_step_mse_tp_ws_bytes(int(B), int(n_chunks)),), dtype=torch.uint8, device=device)
    _check(lib().wdf_clipper_step_mse_tp_ws_init(_ptr(ws), int(B), int(n_chunks), _stream()), "wdf_clipper_step_mse_tp_ws_init")
    return ws


def clipper_step_mse_tp(x, theta, fs, target, gscale, n_chunks, warmup, tol=1e-6, r=None, n_up=1, n_down=1, skip=0, y=None,
                        z0=None, want_zT=False, ws=None, status=None, state=None, gtheta=None, sse=None, accumulate=False,
                        opt=None, time_major=False):
    """The whole MSE training step in one pass over the data (include/wdf_hip.h, wdf_clipper_step_mse_tp):
    forward, loss and d(gscale/2 sum (y - target)^2)/d{Is, nVt, R, C}, x and target read once, y written once, no
    state stash.  n_chunks: time chunks asked for (rounded like wdf_clipper_tp_chunks does); state: a TpWarmState
    made for (B, T, n_chunks); opt: a binding.Adam(4, ...) to update theta in the same launch.
    -> y [T,B], zT [B] | None, gtheta[4], sse[1], status (device int32[4], read with tp_status())."""
    require_gpu()
    x, r, theta = _f32_dev(x, "x"), _f32_dev(r, "r"), _f32_dev(theta, "theta")
    target, z0 = _f32_dev(target, "target"), _f32_dev(z0, "z0")
    B, T = (x.shape[1], x.shape[0]) if time_major else x.shape
    if theta.numel() != 4:
        raise WdfHipError("theta must hold {Is, nVt, R, C}")
    if tuple(target.shape) != (T, B):
        raise WdfHipError(f"target must be [T,B] = [{T},{B}]")
    if r is not None and r.shape != x.shape:
        raise WdfHipError("r must have the shape of x")
    K = lib().wdf_clipper_tp_chunks(T, int(n_chunks))
    if y is None:
        y = torch.empty((T, B), dtype=torch.float32, device=x.device)
    zT = torch.empty((B,), dtype=torch.float32, device=x.device) if want_zT else None
    if ws is None:
        ws = step_mse_workspace(B, K, x.device)
    if status is None:
        status = torch.empty((4,), dtype=torch.int32, device=x.device)
    if gtheta is None:
        gtheta = torch.empty((4,), dtype=torch.float32, device=x.device)
        accumulate = False
    if sse is None:
        sse = torch.empty((1,), dtype=torch.float32, device=x.device)
    if state is not None and ((state.B, state.T, state.K) != (B, T, K) or state.buf.device != x.device):
        raise WdfHipError("warm-start state was made for another batch shape / chunking / device")
    if opt is not None and opt.n != 4:
        raise WdfHipError("clipper_step_mse_tp: the optimizer holds {Is, nVt, R, C}")
    o = opt
    rc = lib().wdf_clipper_step_mse_tp(
        _ptr(x), _ptr(r), _ptr(theta), float(fs), int(n_up), int(n_down), _ptr(target), float(gscale), int(skip), _ptr(y),
        _ptr(z0), _ptr(zT), B, T, K, int(warmup), float(tol), _ptr(ws), _ptr(status),
        None if state is None else _ptr(state.buf), 0 if state is None else state.max_warm_tiles, _ptr(gtheta), _ptr(sse),
        1 if accumulate else 0, *((None,) * 4 if o is None else (_ptr(o.m), _ptr(o.v), _ptr(o.step), _ptr(o.lr))),
        0.0 if o is None else o.b1, 0.0 if o is None else o.b2, 0.0 if o is None else o.eps,
        None if o is None else _ptr(o.lo), None if o is None else _ptr(o.hi),
        (WDF_X_TIME_MAJOR if time_major else 0) | _root_flag() | (WDF_ONE_SEQUENCE_PER_LANE if ONE_SEQUENCE_PER_LANE else 0) |
        (WDF_R_PER_SEQUENCE if r_is_per_sequence(r, time_major) else 0),
        _stream())
    _check(rc, "wdf_clipper_step_mse_tp")
    return y, zT, gtheta, sse, status


def clipper_step_esr_tp(x, theta, fs, target, n_global, eps_energy, skip, n_chunks, warmup, tol=1e-6, r=None, n_up=1, n_down=1,
                        y=None, z0=None, want_zT=False, ws=None, status=None, state=None, sums10=None, gtheta=None, loss3=None,
                        finish=True, opt=None, time_major=False):
    """The one-pass training step for the scripts' loss MSE + ESR past `skip` (include/wdf_hip.h, wdf_clipper_step_esr_tp).
    finish=True: the kernel finishes the step as a single rank -> gtheta[4], loss3 = {mse, esr, mse + esr} (and the Adam
    update of theta with `opt`).  finish=False: only sums10 = {S, E, gP[4], gQ[4]} of this batch comes back -- all-reduce it
    and call esr_finish.  -> y [T,B], zT | None, sums10, gtheta | None, loss3 | None, status."""
    require_gpu()
    x, r, theta = _f32_dev(x, "x"), _f32_dev(r, "r"), _f32_dev(theta, "theta")
    target, z0 = _f32_dev(target, "target"), _f32_dev(z0, "z0")
    B, T = (x.shape[1], x.shape[0]) if time_major else x.shape
    if theta.numel() != 4:
        raise WdfHipError("theta must hold {Is, nVt, R, C}")
    if tuple(target.shape) != (T, B):
        raise WdfHipError(f"target must be [T,B] = [{T},{B}]")
    if r is not None and r.shape != x.shape:
        raise WdfHipError("r must have the shape of x")
    K = lib().wdf_clipper_tp_chunks(T, int(n_chunks))
    dev = x.device
    if y is None:
        y = torch.empty((T, B), dtype=torch.float32, device=dev)
    zT = torch.empty((B,), dtype=torch.float32, device=dev) if want_zT else None
    if ws is None:
        ws = step_mse_workspace(B, K, dev)
    if status is None:
        status = torch.empty((4,), dtype=torch.int32, device=dev)
    if sums10 is None:
        sums10 = torch.empty((10,), dtype=torch.float32, device=dev)
    if finish:
        gtheta = torch.empty((4,), dtype=torch.float32, device=dev) if gtheta is None else gtheta
        loss3 = torch.empty((3,), dtype=torch.float32, device=dev) if loss3 is None else loss3
    else:
        gtheta = loss3 = opt = None
    if state is not None and ((state.B, state.T, state.K) != (B, T, K) or state.buf.device != dev):
        raise WdfHipError("warm-start state was made for another batch shape / chunking / device")
    o = opt
    if o is not None and o.n != 4:
        raise WdfHipError("clipper_step_esr_tp: the optimizer holds {Is, nVt, R, C}")
    rc = lib().wdf_clipper_step_esr_tp(
        _ptr(x), _ptr(r), _ptr(theta), float(fs), int(n_up), int(n_down), _ptr(target), float(n_global), float(eps_energy),
        int(skip), _ptr(y), _ptr(z0), _ptr(zT), B, T, K, int(warmup), float(tol), _ptr(ws), _ptr(status),
        None if state is None else _ptr(state.buf), 0 if state is None else state.max_warm_tiles, _ptr(sums10), _ptr(gtheta),
        _ptr(loss3), *((None,) * 4 if o is None else (_ptr(o.m), _ptr(o.v), _ptr(o.step), _ptr(o.lr))),
        0.0 if o is None else o.b1, 0.0 if o is None else o.b2, 0.0 if o is None else o.eps,
        None if o is None else _ptr(o.lo), None if o is None else _ptr(o.hi),
        (WDF_X_TIME_MAJOR if time_major else 0) | _root_flag() | (WDF_ONE_SEQUENCE_PER_LANE if ONE_SEQUENCE_PER_LANE else 0) |
        (WDF_R_PER_SEQUENCE if r_is_per_sequence(r, time_major) else 0),
        _stream())
    _check(rc, "wdf_clipper_step_esr_tp")
    return y, zT, sums10, gtheta, loss3, status


def esr_finish(sums10, n_global, eps_energy, gtheta=None, loss3=None):
    """(gtheta[4], loss3 = {mse, esr, mse + esr}) from the all-reduced sums10 of clipper_step_esr_tp(finish=False)."""
    require_gpu()
    sums10 = _f32_dev(sums10, "sums10")
    if gtheta is None:
        gtheta = torch.empty((4,), dtype=torch.float32, device=sums10.device)
    if loss3 is None:
        loss3 = torch.empty((3,), dtype=torch.float32, device=sums10.device)
    _check(lib().wdf_esr_finish(_ptr(sums10), float(n_global), float(eps_energy), _ptr(gtheta), _ptr(loss3), _stream()), "wdf_esr_finish")
    return gtheta, loss3


def loss_sums(y, target, skip, sums=None, ws=None):
    """S = sum (y - target)^2 and E = sum y^2 over rows skip.. of [T,B] arrays -> float64[2] on the device."""
    require_gpu()
    y, target = _f32_dev(y, "y"), _f32_dev(target, "target")
    T, B = y.shape
    if tuple(target.shape) != (T, B):
        raise WdfHipError("target must have y's shape [T,B]")
    if ws is None:
        ws = torch.empty((lib().wdf_loss_sums_ws_bytes(),), dtype=torch.uint8, device=y.device)
    if sums is None:
        sums = torch.empty((2,), dtype=torch.float64, device=y.device)
    _check(lib().wdf_loss_sums(_ptr(y), _ptr(target), B, T, int(skip), _ptr(ws), _ptr(sums), _stream()), "wdf_loss_sums")
    return sums


def esr_coef(sums, n, eps, gcoef=None, loss=None):
    """(gcoef[2] = {ga, gb}, loss[3] = {mse, esr, mse + esr}) from the global sums; see include/wdf_hip.h."""
    require_gpu()
    if sums.dtype != torch.float64 or not sums.is_cuda or sums.numel() != 2:
        raise WdfHipError("sums must be a float64[2] device tensor")
    if gcoef is None:
        gcoef = torch.empty((2,), dtype=torch.float32, device=sums.device)
    if loss is None:
        loss = torch.empty((3,), dtype=torch.float32, device=sums.device)
    _check(lib().wdf_esr_coef(_ptr(sums), float(n), float(eps), _ptr(gcoef), _ptr(loss), _stream()), "wdf_esr_coef")
    return gcoef, loss


def loss_esr_grad(y, target, gcoef, skip, gy=None):
    """dL/dy [T,B] of MSE + ESR past `skip` from the device coefficients of esr_coef()."""
    require_gpu()
    y, target, gcoef = _f32_dev(y, "y"), _f32_dev(target, "target"), _f32_dev(gcoef, "gcoef")
    T, B = y.shape
    if tuple(target.shape) != (T, B) or gcoef.numel() != 2:
        raise WdfHipError("target must have y's shape [T,B]; gcoef = {ga, gb}")
    if gy is None:
        gy = torch.empty_like(y)
    _check(lib().wdf_loss_esr_grad(_ptr(y), _ptr(target), _ptr(gcoef), B, T, int(skip), _ptr(gy), _stream()), "wdf_loss_esr_grad")
    return gy


def clipper_bwd_esr_tp(x, theta, fs, zstash, zT, target, gcoef, skip, n_chunks, r=None, n_up=1, n_down=1, gtheta=None,
                       sse=None, ws=None, time_major=False):
    """Reverse sweep with dL/dy = ga (y - target) + gb y past `skip` (MSE + ESR); -> gtheta[4], sse[1]."""
    require_gpu()
    x, r, theta = _f32_dev(x, "x"), _f32_dev(r, "r"), _f32_dev(theta, "theta")
    zstash, zT, target, gcoef = _f32_dev(zstash, "zstash"), _f32_dev(zT, "zT"), _f32_dev(target, "target"), _f32_dev(gcoef, "gcoef")
    B, T = (x.shape[1], x.shape[0]) if time_major else x.shape
    if tuple(target.shape) != (T, B) or tuple(zstash.shape) != (T, B):
        raise WdfHipError(f"target / zstash must be [T,B] = [{T},{B}]")
    if ws is None:
        ws = bwd_tp_workspace(B, int(n_chunks), x.device)
    if gtheta is None:
        gtheta = torch.empty((4,), dtype=torch.float32, device=x.device)
    if sse is None:
        sse = torch.empty((1,), dtype=torch.float32, device=x.device)
    rc = lib().wdf_clipper_bwd_esr_tp(_ptr(x), _ptr(r), _ptr(theta), float(fs), int(n_up), int(n_down), _ptr(zstash),
                                      _ptr(zT), _ptr(target), _ptr(gcoef), int(skip), _ptr(ws), _ptr(gtheta), _ptr(sse),
                                      None, 0, B, T, int(n_chunks), (WDF_X_TIME_MAJOR if time_major else 0) | _root_flag(),
                                      _stream())
    _check(rc, "wdf_clipper_bwd_esr_tp")
    return gtheta, sse


def bwd_tp_workspace(B, n_chunks, device):
    """A reverse-sweep workspace with its ticket words cleared (wdf_clipper_bwd_tp_ws_init): allocate once, reuse."""
    ws = torch.empty((lib().wdf_clipper_bwd_tp_ws_bytes(int(B), int(n_chunks)),), dtype=torch.uint8, device=device)
    _check(lib().wdf_clipper_bwd_tp_ws_init(_ptr(ws), int(B), int(n_chunks), _stream()), "wdf_clipper_bwd_tp_ws_init")
    return ws


def tp_status(status):
    """Host view of a time-parallel status word (synchronises)."""
    s = status.cpu()
    return {"n_bad": int(s[0]), "max_miss": float(s[1:2].view(torch.float32)[0]), "fallback_ran": bool(int(s[2])),
            "repaired_tiles": int(s[2])}


def clipper_bwd_tp(x, theta, fs, zstash, gy, n_chunks, r=None, n_up=1, n_down=1, want_gz0=False, gtheta=None,
                   accumulate=False, ws=None, time_major=False):
    require_gpu()
    x = _f32_dev(x, "x")
    r = _f32_dev(r, "r")
    theta = _f32_dev(theta, "theta")
    zstash = _f32_dev(zstash, "zstash")
    gy = _f32_dev(gy, "gy")
    B, T = (x.shape[1], x.shape[0]) if time_major else x.shape
    if tuple(gy.shape) != (T, B) or tuple(zstash.shape) != (T, B):
        raise WdfHipError(f"gy / zstash must be [T,B] = [{T},{B}]")
    if ws is None:
        ws = bwd_tp_workspace(B, int(n_chunks), x.device)
    if gtheta is None:
        gtheta = torch.empty((4,), dtype=torch.float32, device=x.device)
        accumulate = False
    gz0 = torch.empty((B,), dtype=torch.float32, device=x.device) if want_gz0 else None
    rc = lib().wdf_clipper_bwd_tp(_ptr(x), _ptr(r), _ptr(theta), float(fs), int(n_up), int(n_down), _ptr(zstash),
                                  _ptr(gy), _ptr(ws), _ptr(gtheta), _ptr(gz0), 1 if accumulate else 0, B, T,
                                  int(n_chunks),
                                  (WDF_X_TIME_MAJOR if time_major else 0) | _root_flag(), _stream())
    _check(rc, "wdf_clipper_bwd_tp")
    return gtheta, gz0


def clipper_bwd_mse_tp(x, theta, fs, zstash, zT, target, gscale, n_chunks, r=None, n_up=1, n_down=1, gtheta=None,
                       sse=None, accumulate=False, ws=None, time_major=False):
    """MSE-fused reverse sweep: y is rebuilt from the state stash (zstash [T,B], zT [B]) and
    dL/dy = gscale (y - target) is formed in the kernel.
    Returns (gtheta float32[4], sse float32[1] = sum (y - target)^2 over this batch)."""
    require_gpu()
    x = _f32_dev(x, "x")
    r = _f32_dev(r, "r")
    theta = _f32_dev(theta, "theta")
    zstash = _f32_dev(zstash, "zstash")
    zT = _f32_dev(zT, "zT")
    target = _f32_dev(target, "target")
    B, T = (x.shape[1], x.shape[0]) if time_major else x.shape
    for name, t in (("target", target), ("zstash", zstash)):
        if tuple(t.shape) != (T, B):
            raise WdfHipError(f"{name} must be [T,B] = [{T},{B}]")
    if zT.numel() != B:
        raise WdfHipError("zT must hold the B final states of the forward")
    if ws is None:
        ws = bwd_tp_workspace(B, int(n_chunks), x.device)
    if gtheta is None:
        gtheta = torch.empty((4,), dtype=torch.float32, device=x.device)
        accumulate = False
    if sse is None:
        sse = torch.empty((1,), dtype=torch.float32, device=x.device)
    rc = lib().wdf_clipper_bwd_mse_tp(_ptr(x), _ptr(r), _ptr(theta), float(fs), int(n_up), int(n_down),
                                      _ptr(zstash), _ptr(zT), _ptr(target), float(gscale), _ptr(ws), _ptr(gtheta),
                                      _ptr(sse), None, 1 if accumulate else 0, B, T, int(n_chunks),
                                      (WDF_X_TIME_MAJOR if time_major else 0) | _root_flag(),
                                      _stream())
    _check(rc, "wdf_clipper_bwd_mse_tp")
    return gtheta, sse


def clipper_mlp_fwd(x, theta2, w, hidden, n_tanh, fs, r=None, want_stash=True, z0=None, want_zT=False):
    """MLP-root clipper forward.  x [B,T], theta2 = {R, C}, w flat weights.  -> y [T,B], zstash, zT."""
    require_gpu()
    x = _f32_dev(x, "x")
    r = _f32_dev(r, "r")
    theta2 = _f32_dev(theta2, "theta2")
    w = _f32_dev(w, "w")
    z0 = _f32_dev(z0, "z0")
    n = lib().wdf_mlp_weight_count(int(hidden), int(n_tanh))
    if w.numel() != n:
        raise WdfHipError(f"a {n_tanh - 1}x{hidden} network has {n} weights, got {w.numel()}")
    B, T = x.shape
    y = torch.empty((T, B), dtype=torch.float32, device=x.device)
    zs = torch.empty((T, B), dtype=torch.float32, device=x.device) if want_stash else None
    zT = torch.empty((B,), dtype=torch.float32, device=x.device) if want_zT else None
    rc = lib().wdf_clipper_mlp_fwd(_ptr(x), _ptr(r), _ptr(theta2), _ptr(w), int(hidden), int(n_tanh), float(fs),
                                   _ptr(y), _ptr(zs), _ptr(z0), _ptr(zT), B, T, WDF_MLP_LANE_PER_SEQUENCE if MLP_LANE_PER_SEQUENCE else 0, _stream())
    _check(rc, "wdf_clipper_mlp_fwd")
    return y, zs, zT


def clipper_mlp_bwd(x, theta2, w, hidden, n_tanh, fs, zstash, gy, r=None):
    """-> gtheta2 [2], gb [T,B], ain [T,B], lrin [T,B] | None (see include/wdf_hip.h)."""
    require_gpu()
    x = _f32_dev(x, "x")
    r = _f32_dev(r, "r")
    theta2 = _f32_dev(theta2, "theta2")
    w = _f32_dev(w, "w")
    zstash = _f32_dev(zstash, "zstash")
    gy = _f32_dev(gy, "gy")
    B, T = x.shape
    gb = torch.empty((T, B), dtype=torch.float32, device=x.device)
    ain = torch.empty((T, B), dtype=torch.float32, device=x.device)
    lrin = torch.empty((T, B), dtype=torch.float32, device=x.device) if r is not None else None
    ws = torch.empty((lib().wdf_clipper_mlp_bwd_ws_bytes(B),), dtype=torch.uint8, device=x.device)
    gth = torch.empty((2,), dtype=torch.float32, device=x.device)
    rc = lib().wdf_clipper_mlp_bwd(_ptr(x), _ptr(r), _ptr(theta2), _ptr(w), int(hidden), int(n_tanh), float(fs),
                                   _ptr(zstash), _ptr(gy), _ptr(gb), _ptr(ain), _ptr(lrin), _ptr(ws), _ptr(gth),
                                   B, T, WDF_MLP_LANE_PER_SEQUENCE if MLP_LANE_PER_SEQUENCE else 0, _stream())
    _check(rc, "wdf_clipper_mlp_bwd")
    return gth, gb, ain, lrin


def clipper_mlp_bwd_w(x, theta2, w, hidden, n_tanh, fs, zstash, gy, r=None):
    """-> gtheta2 [2], gw [wdf_mlp_weight_count]: the reverse sweep with the weight gradient folded in."""
    require_gpu()
    x, r, theta2, w = _f32_dev(x, "x"), _f32_dev(r, "r"), _f32_dev(theta2, "theta2"), _f32_dev(w, "w")
    zstash, gy = _f32_dev(zstash, "zstash"), _f32_dev(gy, "gy")
    B, T = x.shape
    nbytes = lib().wdf_clipper_mlp_bwd_w_ws_bytes(int(hidden), int(n_tanh), B)
    if nbytes <= 0:
        raise WdfHipError(f"unsupported MLP root: width {hidden}, {n_tanh} tanh layers")
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    gth = torch.empty((2,), dtype=torch.float32, device=x.device)
    gw = torch.empty((w.numel(),), dtype=torch.float32, device=x.device)
    rc = lib().wdf_clipper_mlp_bwd_w(_ptr(x), _ptr(r), _ptr(theta2), _ptr(w), int(hidden), int(n_tanh), float(fs),
                                     _ptr(zstash), _ptr(gy), _ptr(ws), _ptr(gth), _ptr(gw), B, T, 0, _stream())
    _check(rc, "wdf_clipper_mlp_bwd_w")
    return gth, gw


def clipper_mlp_fwd_tp(x, theta2, w, hidden, n_tanh, fs, n_chunks, warmup, r=None, warmup_per_wave=None, tol=1e-6,
                       want_stash=True, z0=None, want_zT=False, ws=None, status=None, want_kappa=False, zinit=None):
    """Time-parallel MLP-root forward (csrc/wdf_mlp_tp.h).  warmup_per_wave: int32[ceil(B/4)] device tensor
    (multiples of 16) or None.  -> y [T,B], zstash | None, zT | None, status (int32[4]: n_bad, max-miss bits,
    gated waves, 0; read with mlp_tp_status()).  want_kappa: the forward also stores the adjoint recurrence's
    coefficient kappa [T,B] (for clipper_mlp_bwd_w_tp(kappa=...)); returned as a fifth value.  zinit [chunks,B]
    (with want_kappa): the states the chunks start their warm-up from (mlp_tp_starts() names the samples)."""
    require_gpu()
    x, r, theta2, w, z0 = _f32_dev(x, "x"), _f32_dev(r, "r"), _f32_dev(theta2, "theta2"), _f32_dev(w, "w"), _f32_dev(z0, "z0")
    if w.numel() != lib().wdf_mlp_weight_count(int(hidden), int(n_tanh)):
        raise WdfHipError(f"a {n_tanh - 1}x{hidden} network has {lib().wdf_mlp_weight_count(int(hidden), int(n_tanh))} weights, got {w.numel()}")
    B, T = x.shape
    if warmup_per_wave is not None and not (warmup_per_wave.is_cuda and warmup_per_wave.dtype == torch.int32
                                            and warmup_per_wave.numel() == (B + 3) // 4 and warmup_per_wave.is_contiguous()):
        raise WdfHipError("warmup_per_wave: expected a contiguous int32 device tensor with ceil(B/4) entries")
    y = torch.empty((T, B), dtype=torch.float32, device=x.device)
    zs = torch.empty((T, B), dtype=torch.float32, device=x.device) if want_stash else None
    zT = torch.empty((B,), dtype=torch.float32, device=x.device) if want_zT else None
    if ws is None:
        ws = torch.empty((lib().wdf_clipper_mlp_fwd_tp_ws_bytes(B, int(n_chunks)),), dtype=torch.uint8, device=x.device)
    if status is None:
        status = torch.empty((4,), dtype=torch.int32, device=x.device)
    if want_kappa:
        if zs is None:
            raise WdfHipError("want_kappa needs the stash (want_stash=True)")
        kap = torch.empty((T, B), dtype=torch.float32, device=x.device)
        if zinit is not None:
            zinit = _f32_dev(zinit, "zinit")
            if tuple(zinit.shape) != (lib().wdf_clipper_mlp_tp_chunks(T, int(n_chunks)), B) or warmup_per_wave is not None:
                raise WdfHipError("zinit: expected [chunks, B] and no warmup_per_wave")
        rc = lib().wdf_clipper_mlp_fwd_tp_kappa(_ptr(x), _ptr(r), _ptr(theta2), _ptr(w), int(hidden), int(n_tanh), float(fs),
                                                _ptr(y), _ptr(zs), _ptr(kap), _ptr(z0), _ptr(zT), _ptr(zinit), B, T, int(n_chunks),
                                                int(warmup), _ptr(warmup_per_wave), float(tol), _ptr(ws), _ptr(status),
                                                _stream())
        _check(rc, "wdf_clipper_mlp_fwd_tp_kappa")
        return y, zs, zT, status, kap
    rc = lib().wdf_clipper_mlp_fwd_tp(_ptr(x), _ptr(r), _ptr(theta2), _ptr(w), int(hidden), int(n_tanh), float(fs), _ptr(y),
                                      _ptr(zs), _ptr(z0), _ptr(zT), B, T, int(n_chunks), int(warmup), _ptr(warmup_per_wave),
                                      float(tol), _ptr(ws), _ptr(status), _stream())
    _check(rc, "wdf_clipper_mlp_fwd_tp")
    return y, zs, zT, status


def mlp_tp_starts(T, n_chunks, warmup):
    """The sample every chunk's wave begins at (warm-up included) for one warm-up value -> list of ints."""
    K = lib().wdf_clipper_mlp_tp_chunks(int(T), int(n_chunks))
    out = (C.c_int64 * K)()
    _check(lib().wdf_clipper_mlp_tp_starts(int(T), int(n_chunks), int(warmup), out), "wdf_clipper_mlp_tp_starts")
    return list(out)


def mlp_tp_status(status):
    s = status.cpu()
    # gated_waves: 4-sequence waves with a miss at the first pass (re-run chunk-locally from their predecessors' end
    # states); sequential_waves: those that still missed and went through the sequential kernel
    return {"n_bad": int(s[0]), "max_miss": float(s[1:2].view(torch.float32)[0]), "gated_waves": int(s[2]),
            "sequential_waves": int(s[3])}


def clipper_mlp_bwd_w_tp(x, theta2, w, hidden, n_tanh, fs, zstash, gy, n_chunks, r=None, ws=None, kappa=None):
    """Exact reverse sweep parallel over all steps (kappa / adjoint scan / weight-gradient kernels) -> gtheta2 [2], gw.
    kappa [T,B]: the coefficient a want_kappa forward left -- the kappa pass is skipped."""
    require_gpu()
    x, r, theta2, w = _f32_dev(x, "x"), _f32_dev(r, "r"), _f32_dev(theta2, "theta2"), _f32_dev(w, "w")
    zstash, gy = _f32_dev(zstash, "zstash"), _f32_dev(gy, "gy")
    B, T = x.shape
    nbytes = lib().wdf_clipper_mlp_bwd_w_tp_ws_bytes(int(hidden), int(n_tanh), B, T, int(n_chunks))
    if nbytes <= 0:
        raise WdfHipError(f"unsupported MLP root: width {hidden}, {n_tanh} tanh layers")
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    gth = torch.empty((2,), dtype=torch.float32, device=x.device)
    gw = torch.empty((w.numel(),), dtype=torch.float32, device=x.device)
    if kappa is not None:
        kappa = _f32_dev(kappa, "kappa")
        if tuple(kappa.shape) != (T, B):
            raise WdfHipError(f"kappa: expected [{T}, {B}], got {tuple(kappa.shape)}")
        rc = lib().wdf_clipper_mlp_bwd_w_tp_kappa(_ptr(x), _ptr(r), _ptr(theta2), _ptr(w), int(hidden), int(n_tanh), float(fs),
                                                  _ptr(zstash), _ptr(kappa), _ptr(gy), _ptr(ws), _ptr(gth), _ptr(gw), B, T,
                                                  int(n_chunks), _stream())
        _check(rc, "wdf_clipper_mlp_bwd_w_tp_kappa")
        return gth, gw
    rc = lib().wdf_clipper_mlp_bwd_w_tp(_ptr(x), _ptr(r), _ptr(theta2), _ptr(w), int(hidden), int(n_tanh), float(fs),
                                        _ptr(zstash), _ptr(gy), _ptr(ws), _ptr(gth), _ptr(gw), B, T, int(n_chunks), _stream())
    _check(rc, "wdf_clipper_mlp_bwd_w_tp")
    return gth, gw


def mlp_eval(ain, lrin, w, hidden, n_tanh):
    """out[n] = MLP(ain[n], lrin[n]) (flat float32 device tensors)."""
    require_gpu()
    ain = _f32_dev(ain, "ain")
    lrin = _f32_dev(lrin, "lrin")
    w = _f32_dev(w, "w")
    if lrin.numel() != ain.numel():
        raise WdfHipError("mlp_eval: ain and lrin must have the same number of samples")
    out = torch.empty((ain.numel(),), dtype=torch.float32, device=ain.device)
    rc = lib().wdf_mlp_eval(_ptr(ain), _ptr(lrin), _ptr(w), int(hidden), int(n_tanh), _ptr(out), ain.numel(), _stream())
    _check(rc, "wdf_mlp_eval")
    return out


def mlp_fit_epoch(xa, xl, ys, batch, w, opt, hidden, n_tanh, esr_n, eps_energy, loss_sum):
    """One epoch of mini-batch Adam on the table (xa, xl, ys) in the given order; w and the Adam
    state `opt` (binding.Adam with a scalar learning rate) are updated in place; loss_sum: device
    float64[1] receiving the sum of batch losses."""
    require_gpu()
    xa, xl, ys, w = _f32_dev(xa, "xa"), _f32_dev(xl, "xl"), _f32_dev(ys, "ys"), _f32_dev(w, "w")
    if not (xa.numel() == xl.numel() == ys.numel()):
        raise WdfHipError("mlp_fit_epoch: xa, xl, ys must have the same length")
    if loss_sum.dtype != torch.float64 or not loss_sum.is_cuda:
        raise WdfHipError("loss_sum must be a float64 device tensor")
    rc = lib().wdf_mlp_fit_epoch(_ptr(xa), _ptr(xl), _ptr(ys), xa.numel(), int(batch), _ptr(w), _ptr(opt.m), _ptr(opt.v),
                                 _ptr(opt.step), float(opt.lr_scalar), opt.b1, opt.b2, opt.eps, float(esr_n),
                                 float(eps_energy), _ptr(loss_sum), int(hidden), int(n_tanh), _stream())
    _check(rc, "wdf_mlp_fit_epoch")


def clipper_mlp_wgrad(ain, lrin, gb, theta2, w, hidden, n_tanh, fs):
    """-> gw [wdf_mlp_weight_count]: dL/dw from what clipper_mlp_bwd wrote (see include/wdf_hip.h)."""
    require_gpu()
    ain = _f32_dev(ain, "ain")
    lrin = _f32_dev(lrin, "lrin")
    gb = _f32_dev(gb, "gb")
    theta2 = _f32_dev(theta2, "theta2")
    w = _f32_dev(w, "w")
    S = ain.numel()
    if gb.numel() != S or (lrin is not None and lrin.numel() != S):
        raise WdfHipError("clipper_mlp_wgrad: ain, lrin and gb must have the same number of samples")
    nbytes = lib().wdf_clipper_mlp_wgrad_ws_bytes(int(hidden), int(n_tanh), S)
    if nbytes <= 0:
        raise WdfHipError(f"unsupported MLP root: width {hidden}, {n_tanh} tanh layers")
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=ain.device)
    gw = torch.empty((w.numel(),), dtype=torch.float32, device=ain.device)
    rc = lib().wdf_clipper_mlp_wgrad(_ptr(ain), _ptr(lrin), _ptr(gb), _ptr(theta2), _ptr(w), int(hidden), int(n_tanh),
                                     float(fs), _ptr(ws), _ptr(gw), S, _stream())
    _check(rc, "wdf_clipper_mlp_wgrad")
    return gw


ASYM_OMEGA_F32, ASYM_NEWTON_F64 = 0, 1


def clipper_asym_fwd(x, theta6, fs, mode, tol=1e-12, max_iter=50, z0=None, want_zT=False, want_iters=False, want_stash=False):
    """Two-different-diode clipper forward.  Returns y [T,B], zT | None, iters (int64 per wave) | None
    (and the state stash [T,B] as a fourth value when want_stash)."""
    require_gpu()
    x = _f32_dev(x, "x")
    theta6 = _f32_dev(theta6, "theta6")
    z0 = _f32_dev(z0, "z0")
    if theta6.numel() != 6:
        raise WdfHipError("theta6 must hold {Is_up, nVt_up, Is_down, nVt_down, R, C}")
    B, T = x.shape
    y = torch.empty((T, B), dtype=torch.float32, device=x.device)
    zT = torch.empty((B,), dtype=torch.float32, device=x.device) if want_zT else None
    it = torch.zeros(((B + 63) // 64,), dtype=torch.int64, device=x.device) if want_iters else None
    zs = torch.empty((T, B), dtype=torch.float32, device=x.device) if want_stash else None
    rc = lib().wdf_clipper_asym_fwd(_ptr(x), _ptr(theta6), float(fs), int(mode), float(tol), int(max_iter), _ptr(y),
                                    _ptr(zs), _ptr(z0), _ptr(zT), _ptr(it), B, T, _stream())
    _check(rc, "wdf_clipper_asym_fwd")
    return (y, zT, it, zs) if want_stash else (y, zT, it)


def clipper_asym_fwd_tp(x, theta6, fs, mode, n_chunks, warmup, tol=1e-12, max_iter=50, verify_tol=1e-6, z0=None, want_zT=False,
                        want_stash=False):
    """Time-parallel two-different-diode clipper forward (wdf_clipper_asym_fwd_tp).  n_chunks is rounded to a count that
    tiles T in 8-step units.  -> y [T,B], zT | None, zstash | None, status (int32[4]; read with mlp_tp_status())."""
    require_gpu()
    x, theta6, z0 = _f32_dev(x, "x"), _f32_dev(theta6, "theta6"), _f32_dev(z0, "z0")
    if theta6.numel() != 6:
        raise WdfHipError("theta6 must hold {Is_up, nVt_up, Is_down, nVt_down, R, C}")
    B, T = x.shape
    Lc = -(-(-(-T // max(1, int(n_chunks)))) // 8) * 8
    K = -(-T // Lc)
    y = torch.empty((T, B), dtype=torch.float32, device=x.device)
    zs = torch.empty((T, B), dtype=torch.float32, device=x.device) if want_stash else None
    zT = torch.empty((B,), dtype=torch.float32, device=x.device) if want_zT else None
    ws = torch.empty((lib().wdf_clipper_asym_fwd_tp_ws_bytes(B, K),), dtype=torch.uint8, device=x.device)
    status = torch.empty((4,), dtype=torch.int32, device=x.device)
    rc = lib().wdf_clipper_asym_fwd_tp(_ptr(x), _ptr(theta6), float(fs), int(mode), float(tol), int(max_iter), _ptr(y), _ptr(zs),
                                       _ptr(z0), _ptr(zT), B, T, K, int(warmup), float(verify_tol), _ptr(ws), _ptr(status), _stream())
    _check(rc, "wdf_clipper_asym_fwd_tp")
    return y, zT, zs, status


def clipper_asym_bwd(x, theta6, fs, zstash, gy, tol=1e-12, max_iter=50):
    """dL/d{Is_up, nVt_up, Is_down, nVt_down, R, C} of the Newton-mode loop for dL/dy = gy [T,B]."""
    require_gpu()
    x, theta6, zstash, gy = _f32_dev(x, "x"), _f32_dev(theta6, "theta6"), _f32_dev(zstash, "zstash"), _f32_dev(gy, "gy")
    B, T = x.shape
    if tuple(gy.shape) != (T, B) or tuple(zstash.shape) != (T, B):
        raise WdfHipError(f"gy / zstash must be [T,B] = [{T},{B}]")
    ws = torch.empty((lib().wdf_clipper_asym_bwd_ws_bytes(B),), dtype=torch.uint8, device=x.device)
    g = torch.empty((6,), dtype=torch.float32, device=x.device)
    rc = lib().wdf_clipper_asym_bwd(_ptr(x), _ptr(theta6), float(fs), float(tol), int(max_iter), _ptr(zstash), _ptr(gy),
                                    _ptr(ws), _ptr(g), B, T, _stream())
    _check(rc, "wdf_clipper_asym_bwd")
    return g


def clipper_asym_bwd_tp(x, theta6, fs, mode, zstash, zT, gy, n_chunks, gzT=None, want_gz0=False, ws=None):
    """Time-parallel reverse sweep of the two-different-diode clipper, either mode (wdf_clipper_asym_bwd_tp): no root
    re-solve (consecutive stash entries give b), chunks composed exactly.  zT [B]: the forward's final state.
    -> gtheta6 (and dL/dz0 [B] when want_gz0)."""
    require_gpu()
    x, theta6, zstash, gy = _f32_dev(x, "x"), _f32_dev(theta6, "theta6"), _f32_dev(zstash, "zstash"), _f32_dev(gy, "gy")
    zT, gzT = _f32_dev(zT, "zT"), _f32_dev(gzT, "gzT")
    B, T = x.shape
    if tuple(gy.shape) != (T, B) or tuple(zstash.shape) != (T, B) or zT is None or zT.numel() != B:
        raise WdfHipError(f"gy / zstash must be [T,B] = [{T},{B}], zT [{B}]")
    Lc = -(-(-(-T // max(1, int(n_chunks)))) // 8) * 8
    K = -(-T // Lc)
    need = lib().wdf_clipper_asym_bwd_tp_ws_bytes(B, K)
    if ws is None or ws.numel() < need:
        ws = torch.empty((need,), dtype=torch.uint8, device=x.device)
    g = torch.empty((6,), dtype=torch.float32, device=x.device)
    gz0 = torch.empty((B,), dtype=torch.float32, device=x.device) if want_gz0 else None
    rc = lib().wdf_clipper_asym_bwd_tp(_ptr(x), _ptr(theta6), float(fs), int(mode), _ptr(zstash), _ptr(zT), _ptr(gy), _ptr(gzT),
                                       _ptr(ws), _ptr(g), _ptr(gz0), B, T, K, _stream())
    _check(rc, "wdf_clipper_asym_bwd_tp")
    return (g, gz0) if want_gz0 else g


def asym_root(a, theta6, fs, mode, tol=1e-12, max_iter=50):
    require_gpu()
    a = _f32_dev(a, "a")
    theta6 = _f32_dev(theta6, "theta6")
    b = torch.empty(a.shape, dtype=torch.float64, device=a.device)
    _check(lib().wdf_asym_root(_ptr(a), _ptr(theta6), float(fs), int(mode), float(tol), int(max_iter), _ptr(b),
                               a.numel(), _stream()), "wdf_asym_root")
    return b


ROOT_NONE, ROOT_DIODE_PAIR, ROOT_MLP = 0, 2, 3


def ss_fwd(x, coef, ns, ni, root_kind=ROOT_NONE, rootp=None, n_up=1, n_down=1, want_stash=True, z0=None,
           want_zT=False):
    """State-space recursion.  x [B,T,ni] (or [B,T] when ni == 1) -> y [T,B], zstash [T,ns,B] | None,
    zT [ns,B] | None."""
    require_gpu()
    x = _f32_dev(x, "x")
    coef = _f32_dev(coef, "coef")
    rootp = _f32_dev(rootp, "rootp")
    z0 = _f32_dev(z0, "z0")
    B, T = x.shape[0], x.shape[1]
    if x.numel() != B * T * ni:
        raise WdfHipError(f"x has {x.numel()} elements, expected B*T*ni = {B * T * ni}")
    if coef.numel() != lib().wdf_ss_ncoef(ns, ni):
        raise WdfHipError(f"coef must hold {lib().wdf_ss_ncoef(ns, ni)} values for ns={ns}, ni={ni}")
    y = torch.empty((T, B), dtype=torch.float32, device=x.device)
    zs = torch.empty((T, ns, B), dtype=torch.float32, device=x.device) if (want_stash and ns > 0) else None
    zT = torch.empty((ns, B), dtype=torch.float32, device=x.device) if want_zT else None
    rc = lib().wdf_ss_fwd(_ptr(x), _ptr(coef), _ptr(rootp), ns, ni, root_kind, int(n_up), int(n_down),
                          _ptr(y), _ptr(zs), _ptr(z0), _ptr(zT), B, T, 0, _stream())
    _check(rc, "wdf_ss_fwd")
    return y, zs, zT


def ss_fwd_lin_tp(x, coef, ns, ni, n_chunks, want_stash=True, z0=None, want_zT=False):
    """Exact chunked forward of a linear tree (root kind NONE); same returns as ss_fwd."""
    require_gpu()
    x, coef, z0 = _f32_dev(x, "x"), _f32_dev(coef, "coef"), _f32_dev(z0, "z0")
    B, T = x.shape[0], x.shape[1]
    if x.numel() != B * T * ni or coef.numel() != lib().wdf_ss_ncoef(ns, ni):
        raise WdfHipError("ss_fwd_lin_tp: x / coef do not match ns, ni")
    y = torch.empty((T, B), dtype=torch.float32, device=x.device)
    zs = torch.empty((T, ns, B), dtype=torch.float32, device=x.device) if want_stash else None
    zT = torch.empty((ns, B), dtype=torch.float32, device=x.device) if want_zT else None
    ws = torch.empty((lib().wdf_ss_fwd_lin_tp_ws_bytes(ns, B, int(n_chunks)),), dtype=torch.uint8, device=x.device)
    rc = lib().wdf_ss_fwd_lin_tp(_ptr(x), _ptr(coef), ns, ni, _ptr(y), _ptr(zs), _ptr(z0), _ptr(zT), B, T, int(n_chunks),
                                 _ptr(ws), _stream())
    _check(rc, "wdf_ss_fwd_lin_tp")
    return y, zs, zT


def ss_bwd(x, coef, ns, ni, zstash, gy, root_kind=ROOT_NONE, rootp=None, n_up=1, n_down=1, want_gz0=False):
    """Returns (gcoef [ncoef], groot [3] | None, gz0 [ns,B] | None)."""
    require_gpu()
    x = _f32_dev(x, "x")
    coef = _f32_dev(coef, "coef")
    rootp = _f32_dev(rootp, "rootp")
    zstash = _f32_dev(zstash, "zstash")
    gy = _f32_dev(gy, "gy")
    B, T = x.shape[0], x.shape[1]
    if tuple(gy.shape) != (T, B):
        raise WdfHipError(f"gy must be [T,B] = [{T},{B}]")
    ncoef = lib().wdf_ss_ncoef(ns, ni)
    ws = torch.empty((lib().wdf_ss_bwd_ws_bytes(ns, ni, B),), dtype=torch.uint8, device=x.device)
    gcoef = torch.empty((ncoef,), dtype=torch.float32, device=x.device)
    groot = torch.empty((3,), dtype=torch.float32, device=x.device) if root_kind == ROOT_DIODE_PAIR else None
    gz0 = torch.empty((ns, B), dtype=torch.float32, device=x.device) if (want_gz0 and ns > 0) else None
    rc = lib().wdf_ss_bwd(_ptr(x), _ptr(coef), _ptr(rootp), ns, ni, root_kind, int(n_up), int(n_down),
                          _ptr(zstash), _ptr(gy), _ptr(ws), _ptr(gcoef), _ptr(groot), _ptr(gz0), B, T, 0, _stream())
    _check(rc, "wdf_ss_bwd")
    return gcoef, groot, gz0


def ss_tp_starts(T, n_chunks, warmup):
    """The sample every chunk's wave begins at (warm-up included) -> list of ints (rows of a stash to hand in as zinit)."""
    K = lib().wdf_ss_tp_chunks(int(T), int(n_chunks))
    out = (C.c_int64 * K)()
    _check(lib().wdf_ss_tp_starts(int(T), K, int(warmup), out), "wdf_ss_tp_starts")
    return list(out)


def ss_fwd_tp(x, coef, ns, ni, rootp, n_chunks, warmup, tol=1e-6, n_up=1, n_down=1, want_stash=True, z0=None, want_zT=False,
              zinit=None):
    """Time-parallel forward of a tree with a diode-pair root (include/wdf_hip.h, wdf_ss_fwd_tp): chunks warmed up from
    z = 0 (or from zinit [chunks, ns, B]: states for the samples ss_tp_starts names), boundaries verified on the device,
    missed waves re-run sequentially behind a gate.
    -> y [T,B], zstash [T,ns,B] | None, zT [ns,B] | None, status (device int32[4]: read with ss_tp_status)."""
    require_gpu()
    x, coef, rootp, z0 = _f32_dev(x, "x"), _f32_dev(coef, "coef"), _f32_dev(rootp, "rootp"), _f32_dev(z0, "z0")
    zinit = _f32_dev(zinit, "zinit")
    B, T = x.shape[0], x.shape[1]
    if x.numel() != B * T * ni or coef.numel() != lib().wdf_ss_ncoef(ns, ni):
        raise WdfHipError("ss_fwd_tp: x / coef do not match ns, ni")
    K = lib().wdf_ss_tp_chunks(T, int(n_chunks))
    y = torch.empty((T, B), dtype=torch.float32, device=x.device)
    zs = torch.empty((T, ns, B), dtype=torch.float32, device=x.device) if want_stash else None
    zT = torch.empty((ns, B), dtype=torch.float32, device=x.device) if want_zT else None
    ws = torch.empty((lib().wdf_ss_fwd_tp_ws_bytes(ns, B, K),), dtype=torch.uint8, device=x.device)
    status = torch.empty((4,), dtype=torch.int32, device=x.device)
    if zinit is not None and tuple(zinit.shape) != (K, ns, B):
        raise WdfHipError(f"zinit: expected [chunks, ns, B] = [{K},{ns},{B}]")
    rc = lib().wdf_ss_fwd_tp(_ptr(x), _ptr(coef), _ptr(rootp), ns, ni, int(n_up), int(n_down), _ptr(y), _ptr(zs), _ptr(z0),
                             _ptr(zT), B, T, K, int(warmup), float(tol), _ptr(zinit), _ptr(ws), _ptr(status), _stream())
    _check(rc, "wdf_ss_fwd_tp")
    return y, zs, zT, status


def ss_tp_status(status):
    s = status.cpu()
    return {"n_bad": int(s[0]), "max_miss": float(s[1:2].view(torch.float32)[0]), "gated_waves": int(s[2])}


def _dyn_rows_mode(rows, T, n, B, who):
    """per_sample argument of the wdf_ss_dyn_* entry points from the rows' shape: [n] -> 0 (one static row), [T,n,B] -> 1 (a row
    per sample), [1,n,B] with T > 1 -> 2 (a row per sequence: constant along the time axis)."""
    shp = tuple(rows.shape)
    if n and shp == (n,):
        return 0
    if n and shp == (T, n, B):
        return 1
    if n and shp == (1, n, B):
        return 2
    raise WdfHipError(f"{who}: rows [T,{n},B] (per sample), [1,{n},B] (per sequence) or [{n}] (static), got {shp}")


def ss_dyn_fwd(x, rows, ns, ni, root_kind=ROOT_NONE, rootp=None, w=None, hidden=0, n_tanh=0, n_up=1, n_down=1, want_stash=True,
               z0=None, want_zT=False):
    """State-space recursion with streamed coefficient rows / the MLP root on any small tree (wdf_ss_dyn_fwd).
    x [B,T,ni]; rows [T,n,B] (per sample) or [n] (one static row), n = wdf_ss_dyn_row_len(ns, ni).
    -> y [T,B], zstash [T,ns,B] | None, zT [ns,B] | None."""
    require_gpu()
    x, rows, rootp, w, z0 = _f32_dev(x, "x"), _f32_dev(rows, "rows"), _f32_dev(rootp, "rootp"), _f32_dev(w, "w"), _f32_dev(z0, "z0")
    B, T = int(x.shape[0]), int(x.shape[1])
    n = lib().wdf_ss_dyn_row_len(int(ns), int(ni))
    if x.dim() != 3 or int(x.shape[2]) != ni:
        raise WdfHipError(f"ss_dyn_fwd: x [B,T,{ni}] (got {tuple(x.shape)})")
    per = _dyn_rows_mode(rows, T, n, B, "ss_dyn_fwd")
    y = torch.empty((T, B), dtype=torch.float32, device=x.device)
    zs = torch.empty((T, max(ns, 1), B), dtype=torch.float32, device=x.device) if (want_stash and ns > 0) else None
    zT = torch.empty((max(ns, 1), B), dtype=torch.float32, device=x.device) if want_zT else None
    rc = lib().wdf_ss_dyn_fwd(_ptr(x), _ptr(rows), per, int(ns), int(ni), int(root_kind), _ptr(rootp), _ptr(w), int(hidden),
                              int(n_tanh), int(n_up), int(n_down), _ptr(y), _ptr(zs), _ptr(z0), _ptr(zT), B, T, _stream())
    _check(rc, "wdf_ss_dyn_fwd")
    return y, zs, zT


def ss_dyn_bwd(x, rows, ns, ni, zstash, gy, root_kind=ROOT_NONE, rootp=None, w=None, hidden=0, n_tanh=0, n_up=1, n_down=1, want_gz0=False):
    """Reverse sweep of ss_dyn_fwd for dL/dy = gy [T,B] (wdf_ss_dyn_bwd).
    -> grows [T,n,B] (dL/d row entry of every sample; rows constant in time -- [n] or [1,n,B] -- : [1,n,B], summed over the steps by
    the kernel), groot (diode: float64 [2] = dL/d{Is, nVt}; MLP: the flat weight gradient;
    else None), gz0 [ns,B] | None."""
    require_gpu()
    x, rows, rootp, w = _f32_dev(x, "x"), _f32_dev(rows, "rows"), _f32_dev(rootp, "rootp"), _f32_dev(w, "w")
    zstash, gy = _f32_dev(zstash, "zstash"), _f32_dev(gy, "gy")
    B, T = int(x.shape[0]), int(x.shape[1])
    n = lib().wdf_ss_dyn_row_len(int(ns), int(ni))
    per = _dyn_rows_mode(rows, T, n, B, "ss_dyn_bwd")
    dev = x.device
    # per sample: dL/d(row) of every sample [T,n,B]; rows constant in time: the kernel's own sum over the steps [1,n,B]
    grows = torch.empty((T if per == 1 else 1, n, B), dtype=torch.float32, device=dev)
    ws = torch.zeros(((B + 63) // 64, 2), dtype=torch.float64, device=dev)
    mlp = root_kind == ROOT_MLP
    gb, ain, lrin = (torch.empty((T, B), dtype=torch.float32, device=dev) for _ in range(3)) if mlp else (None, None, None)
    gz0 = torch.empty((max(ns, 1), B), dtype=torch.float32, device=dev) if want_gz0 else None
    rc = lib().wdf_ss_dyn_bwd(_ptr(x), _ptr(rows), per, int(ns), int(ni), int(root_kind), _ptr(rootp), _ptr(w), int(hidden),
                              int(n_tanh), int(n_up), int(n_down), _ptr(zstash), _ptr(gy), _ptr(grows), _ptr(ws), _ptr(gb), _ptr(ain),
                              _ptr(lrin), _ptr(gz0), B, T, _stream())
    _check(rc, "wdf_ss_dyn_bwd")
    groot = None
    if root_kind == ROOT_DIODE_PAIR:
        s = ws.sum(dim=0)
        rp = rootp.double()
        groot = torch.stack([s[0] / rp[0], s[1] - s[0] / rp[1]])
    elif mlp:
        groot = clipper_mlp_wgrad(ain.reshape(-1), lrin.reshape(-1), gb.reshape(-1), None, w, hidden, n_tanh, 1.0)
    return grows, groot, gz0


def dyn_chunks(T, n_chunks):
    """The chunk count wdf_ss_dyn_*_tp accept for a requested one: chunks are whole 8-step units."""
    Lc = -(-(-(-int(T) // max(1, int(n_chunks)))) // 8) * 8
    return -(-int(T) // Lc)


def dyn_chunk_len(T, n_chunks):
    return -(-(-(-int(T) // max(1, int(n_chunks)))) // 8) * 8


def ss_dyn_fwd_tp(x, rows, ns, ni, n_chunks, warmup, tol=1.0e-6, root_kind=ROOT_NONE, rootp=None, w=None, hidden=0, n_tanh=0, n_up=1,
                  n_down=1, want_stash=True, z0=None, want_zT=False, zinit=None):
    """ss_dyn_fwd in verified time chunks (wdf_ss_dyn_fwd_tp).  -> y, zstash | None, zT | None, status (int32 [4]: ss_tp_status)."""
    require_gpu()
    x, rows, rootp, w, z0 = _f32_dev(x, "x"), _f32_dev(rows, "rows"), _f32_dev(rootp, "rootp"), _f32_dev(w, "w"), _f32_dev(z0, "z0")
    zinit = _f32_dev(zinit, "zinit")
    B, T = int(x.shape[0]), int(x.shape[1])
    n = lib().wdf_ss_dyn_row_len(int(ns), int(ni))
    if x.dim() != 3 or int(x.shape[2]) != ni:
        raise WdfHipError(f"ss_dyn_fwd_tp: x [B,T,{ni}] (got {tuple(x.shape)})")
    per = _dyn_rows_mode(rows, T, n, B, "ss_dyn_fwd_tp")
    K = dyn_chunks(T, n_chunks)
    if zinit is not None and tuple(zinit.shape) != (K, ns, B):
        raise WdfHipError(f"ss_dyn_fwd_tp: zinit must be [{K},{ns},{B}], got {tuple(zinit.shape)}")
    y = torch.empty((T, B), dtype=torch.float32, device=x.device)
    zs = torch.empty((T, ns, B), dtype=torch.float32, device=x.device) if want_stash else None
    zT = torch.empty((ns, B), dtype=torch.float32, device=x.device) if want_zT else None
    ws = torch.empty((lib().wdf_ss_dyn_fwd_tp_ws_bytes(int(ns), B, K),), dtype=torch.uint8, device=x.device)
    status = torch.zeros((4,), dtype=torch.int32, device=x.device)
    rc = lib().wdf_ss_dyn_fwd_tp(_ptr(x), _ptr(rows), per, int(ns), int(ni), int(root_kind), _ptr(rootp), _ptr(w), int(hidden),
                                 int(n_tanh), int(n_up), int(n_down), _ptr(y), _ptr(zs), _ptr(z0), _ptr(zT), B, T, K, int(warmup),
                                 float(tol), _ptr(zinit), _ptr(ws), _ptr(status), _stream())
    _check(rc, "wdf_ss_dyn_fwd_tp")
    return y, zs, zT, status


ROWS_MAX_OPS, ROWS_MAX_CONSTS, ROWS_MAX_OUT, ROWS_MAX_PARAMS = 192, 32, 48, 15      # csrc/wdf_ss_dyn_rows.h


class RowsTape:
    """A probe tape (probe_tape.Tape.packed()) and the nodes of the row entries, as the host arrays wdf_ss_dyn_rows takes."""

    def __init__(self, ops, consts, outs):
        self.ops = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 3)
        self.consts = np.ascontiguousarray(consts, dtype=np.float64).reshape(-1)
        self.outs = np.ascontiguousarray(outs, dtype=np.int32).reshape(-1)

    def fits(self, n_params):
        return (len(self.ops) <= ROWS_MAX_OPS and len(self.consts) <= ROWS_MAX_CONSTS and len(self.outs) <= ROWS_MAX_OUT
                and n_params <= ROWS_MAX_PARAMS)

    def args(self):
        ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        return (self.ops.ctypes.data_as(ip), len(self.ops), self.consts.ctypes.data_as(dp), len(self.consts),
                self.outs.ctypes.data_as(ip), len(self.outs))


def _rows_params(params, r):
    if params.dtype != torch.float64 or not params.is_cuda or not params.is_contiguous() or params.dim() != 1:
        raise WdfHipError("ss_dyn_rows: params is a contiguous 1-d float64 tensor on the GPU")
    if r is None:
        return None, 1, 1
    r = _f32_dev(r, "r")
    if r.dim() != 2:
        raise WdfHipError("ss_dyn_rows: r is [T, B]")
    return r, int(r.shape[1]), int(r.shape[0])


def ss_dyn_rows(tape, params, chan, r):
    """rows [T, n, B] float32 (r None: [1, n, 1]) of the probed step `tape` (RowsTape) at the component values `params` (float64
    [P], device), parameter `chan` taken per sample from r [T, B] (wdf_ss_dyn_rows)."""
    require_gpu()
    r, B, T = _rows_params(params, r)
    rows = torch.empty((T, len(tape.outs), B), dtype=torch.float32, device=params.device)
    rc = lib().wdf_ss_dyn_rows(*tape.args(), _ptr(params), int(params.numel()), int(chan), _ptr(r), _ptr(rows), B, T, _stream())
    _check(rc, "wdf_ss_dyn_rows")
    return rows


def ss_dyn_rows_bwd(tape, params, chan, r, grows):
    """dLoss/dparams float64 [P] from the rows' adjoint grows [T, n, B] (wdf_ss_dyn_rows_bwd); the channel's entry is 0."""
    require_gpu()
    r, B, T = _rows_params(params, r)
    grows = _f32_dev(grows, "grows")
    if tuple(grows.shape) != (T, len(tape.outs), B):
        raise WdfHipError(f"ss_dyn_rows_bwd: grows must be [{T}, {len(tape.outs)}, {B}]")
    P = int(params.numel())
    ws = torch.empty((lib().wdf_ss_dyn_rows_bwd_ws_bytes(P, B, T),), dtype=torch.uint8, device=params.device)
    gp = torch.empty((P,), dtype=torch.float64, device=params.device)
    rc = lib().wdf_ss_dyn_rows_bwd(*tape.args(), _ptr(params), P, int(chan), _ptr(r), _ptr(grows), _ptr(ws), _ptr(gp), B, T, _stream())
    _check(rc, "wdf_ss_dyn_rows_bwd")
    return gp


def ss_dyn_bwd_tp(x, rows, ns, ni, zstash, gy, n_chunks, root_kind=ROOT_NONE, rootp=None, w=None, hidden=0, n_tanh=0, n_up=1, n_down=1,
                  want_gz0=False):
    """ss_dyn_bwd in exact time chunks (wdf_ss_dyn_bwd_tp): same results up to fp32 summation order."""
    require_gpu()
    x, rows, rootp, w = _f32_dev(x, "x"), _f32_dev(rows, "rows"), _f32_dev(rootp, "rootp"), _f32_dev(w, "w")
    zstash, gy = _f32_dev(zstash, "zstash"), _f32_dev(gy, "gy")
    B, T = int(x.shape[0]), int(x.shape[1])
    n = lib().wdf_ss_dyn_row_len(int(ns), int(ni))
    per = _dyn_rows_mode(rows, T, n, B, "ss_dyn_bwd_tp")
    dev = x.device
    K = dyn_chunks(T, n_chunks)
    # per sample: [T,n,B]; rows constant in time: one partial per chunk [K,n,B], added up below
    grows = torch.empty((T if per == 1 else K, n, B), dtype=torch.float32, device=dev)
    ws = torch.empty((lib().wdf_ss_dyn_bwd_tp_ws_bytes(int(ns), B, T, K),), dtype=torch.uint8, device=dev)
    mlp = root_kind == ROOT_MLP
    gb, ain, lrin = (torch.empty((T, B), dtype=torch.float32, device=dev) for _ in range(3)) if mlp else (None, None, None)
    gz0 = torch.empty((ns, B), dtype=torch.float32, device=dev) if want_gz0 else None
    rc = lib().wdf_ss_dyn_bwd_tp(_ptr(x), _ptr(rows), per, int(ns), int(ni), int(root_kind), _ptr(rootp), _ptr(w), int(hidden),
                                 int(n_tanh), int(n_up), int(n_down), _ptr(zstash), _ptr(gy), _ptr(grows), _ptr(ws), _ptr(gb), _ptr(ain),
                                 _ptr(lrin), _ptr(gz0), B, T, K, _stream())
    _check(rc, "wdf_ss_dyn_bwd_tp")
    groot = None
    if root_kind == ROOT_DIODE_PAIR:
        nparts = K * ((B + 63) // 64)
        s = ws[:nparts * 16].view(torch.float64).reshape(nparts, 2).sum(dim=0)
        rp = rootp.double()
        groot = torch.stack([s[0] / rp[0], s[1] - s[0] / rp[1]])
    elif mlp:
        groot = clipper_mlp_wgrad(ain.reshape(-1), lrin.reshape(-1), gb.reshape(-1), None, w, hidden, n_tanh, 1.0)
    if per != 1:
        grows = grows.sum(dim=0, keepdim=True, dtype=torch.float64).float() if K > 1 else grows      # [1,n,B]: the chunks in order
    return grows, groot, gz0


def ss_bwd_tp(x, coef, ns, ni, zstash, gy, n_chunks, root_kind=ROOT_NONE, rootp=None, n_up=1, n_down=1, want_gz0=False):
    """Exact chunked reverse sweep (wdf_ss_bwd_tp); same returns as ss_bwd."""
    require_gpu()
    x, coef, rootp = _f32_dev(x, "x"), _f32_dev(coef, "coef"), _f32_dev(rootp, "rootp")
    zstash, gy = _f32_dev(zstash, "zstash"), _f32_dev(gy, "gy")
    B, T = x.shape[0], x.shape[1]
    if tuple(gy.shape) != (T, B):
        raise WdfHipError(f"gy must be [T,B] = [{T},{B}]")
    K = lib().wdf_ss_tp_chunks(T, int(n_chunks))
    ncoef = lib().wdf_ss_ncoef(ns, ni)
    ws = torch.empty((lib().wdf_ss_bwd_tp_ws_bytes(ns, ni, B, K),), dtype=torch.uint8, device=x.device)
    gcoef = torch.empty((ncoef,), dtype=torch.float32, device=x.device)
    groot = torch.empty((3,), dtype=torch.float32, device=x.device) if root_kind == ROOT_DIODE_PAIR else None
    gz0 = torch.empty((ns, B), dtype=torch.float32, device=x.device) if want_gz0 else None
    rc = lib().wdf_ss_bwd_tp(_ptr(x), _ptr(coef), _ptr(rootp), ns, ni, root_kind, int(n_up), int(n_down), _ptr(zstash),
                             _ptr(gy), _ptr(ws), _ptr(gcoef), _ptr(groot), _ptr(gz0), B, T, K, _stream())
    _check(rc, "wdf_ss_bwd_tp")
    return gcoef, groot, gz0


def omega(x, want_iters=False):
    require_gpu()
    x = _f32_dev(x, "x")
    w = torch.empty_like(x)
    it = torch.empty(x.shape, dtype=torch.int32, device=x.device) if want_iters else None
    _check(lib().wdf_omega_f32(_ptr(x), _ptr(w), _ptr(it), x.numel(), _stream()), "wdf_omega_f32")
    return (w, it) if want_iters else w


def diode_pair(a, R_port, Is, nVt, n_up=1, n_down=1):
    require_gpu()
    a = _f32_dev(a, "a")
    R_port = _f32_dev(R_port, "R_port")
    b = torch.empty_like(a)
    _check(lib().wdf_diode_pair_f32(_ptr(a), _ptr(R_port), float(Is), float(nVt), int(n_up), int(n_down),
                                    _ptr(b), a.numel(), _stream()), "wdf_diode_pair_f32")
    return b


class Adam:
    """Device-resident tf.keras.optimizers.Adam for a small float32 parameter vector (component
    values, flat MLP weights <= 1024): per-parameter learning rates and clip constraints
    (tf_wdf.py:74,104); one kernel launch per step, no host round trip."""

    def __init__(self, n, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7, lo=None, hi=None, device="cuda"):
        require_gpu()
        self.n = int(n)
        self.b1, self.b2, self.eps = float(beta_1), float(beta_2), float(epsilon)
        f = lambda a: torch.as_tensor(a, dtype=torch.float32).expand(self.n).contiguous().to(device)  # noqa: E731
        self.lr = f(lr)
        self.lr_scalar = float(lr) if np.ndim(lr) == 0 else None
        self.lo = None if lo is None else f(lo)
        self.hi = None if hi is None else f(hi)
        self.m = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.v = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.step = torch.zeros(1, dtype=torch.int32, device=device)

    def apply(self, theta, grad):
        """theta (in place) <- Adam update with grad, then the clip constraint."""
        theta = _f32_dev(theta, "theta")
        grad = _f32_dev(grad, "grad")
        if theta.numel() != self.n or grad.numel() != self.n:
            raise WdfHipError(f"Adam: expected {self.n} parameters")
        rc = lib().wdf_adam_step(_ptr(theta), _ptr(grad), _ptr(self.m), _ptr(self.v), _ptr(self.step), _ptr(self.lr),
                                 self.b1, self.b2, self.eps, _ptr(self.lo), _ptr(self.hi), self.n, _stream())
        _check(rc, "wdf_adam_step")


class _AdamJob(C.Structure):                       # include/wdf_hip.h: wdf_adam_job
    _fields_ = [("theta", C.c_void_p), ("grad", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("step", C.c_void_p),
                ("lr", C.c_void_p), ("lo", C.c_void_p), ("hi", C.c_void_p), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("n", C.c_int)]


ADAM_MULTI_MAX = 8


def adam_jobs(jobs):
    """[(Adam, theta, grad), ...] -> the C array of wdf_adam_job."""
    arr = (_AdamJob * len(jobs))()
    for a, (opt, theta, grad) in zip(arr, jobs):
        theta, grad = _f32_dev(theta, "theta"), _f32_dev(grad, "grad")
        if theta.numel() != opt.n or grad.numel() != opt.n:
            raise WdfHipError(f"Adam: expected {opt.n} parameters")
        a.theta, a.grad, a.m, a.v, a.step, a.lr = _ptr(theta), _ptr(grad), _ptr(opt.m), _ptr(opt.v), _ptr(opt.step), _ptr(opt.lr)
        a.lo, a.hi, a.beta1, a.beta2, a.eps, a.n = _ptr(opt.lo), _ptr(opt.hi), opt.b1, opt.b2, opt.eps, opt.n
    return arr


def adam_step_multi(jobs):
    """jobs: [(Adam, theta, grad), ...] (at most ADAM_MULTI_MAX, distinct optimizers): every update in ONE launch."""
    if len(jobs) == 1:
        jobs[0][0].apply(jobs[0][1], jobs[0][2])
        return
    arr = adam_jobs(jobs)
    _check(lib().wdf_adam_step_multi(C.cast(arr, C.c_void_p), len(jobs), _stream()), "wdf_adam_step_multi")


def clock_stamp(out):
    """out: int64 device tensor [2] <- {shader clock counter, 100 MHz counter} when the stream gets here."""
    _check(lib().wdf_clock_stamp(_ptr(out), _stream()), "wdf_clock_stamp")


def device_info(device=0):
    buf = C.create_string_buffer(64)
    cus = lib().wdf_device_info(int(device), buf, 64)
    if cus < 0:
        raise WdfHipError(lib().wdf_last_error().decode())
    return buf.value.decode(), cus


class Event:
    """HIP event on the stream the kernels run on (bench.py roofline timing)."""

    def __init__(self):
        self.h = lib().wdf_event_create()
        if not self.h:
            raise WdfHipError("hipEventCreate failed")

    def record(self):
        _check(lib().wdf_event_record(self.h, _stream()), "wdf_event_record")

    def elapsed_ms(self, stop):
        ms = C.c_float(0.0)
        _check(lib().wdf_event_elapsed_ms(self.h, stop.h, C.byref(ms)), "wdf_event_elapsed_ms")
        return ms.value

    @staticmethod
    def bracket_next(start, stop):
        """Record start/stop right around the next recurrence kernel launched from this thread
        (excluding the verify / combine / reduce helpers of the same C call)."""
        lib().wdf_event_bracket_next(start.h, stop.h)

    def __del__(self):
        try:
            lib().wdf_event_destroy(self.h)
        except Exception:
            pass
