"""ctypes binding of libeasyhybrid_hip.so (include/easyhybrid_hip.h).

This is the same calling convention Julia's `@ccall` uses; the Julia stub in
`julia/EasyHybridHIP/src/EasyHybridHIP.jl` binds the identical symbols.  There is no fallback: if the
shared library is missing or a symbol cannot be resolved, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

EH_MAX_HIDDEN, EH_MAX_PARAMS, EH_MAX_FORC, EH_MAX_TARG, EH_MAX_NETS = 8, 8, 4, 4, 8
EH_MAX_PROG, EH_MAX_PROG_CONST, EH_MAX_PROG_OUT = 64, 16, 3
EH_MECH_PROGRAM = 6
EH_LOSS_PROGRAM = 7
EH_OK, EH_EINVAL, EH_EHIP, EH_ENOMEM, EH_EUNSUPPORTED, EH_ESTATE, EH_ERCCL = 0, -1, -2, -3, -4, -5, -6
EH_COMM_ID_BYTES = 128
EH_SPLIT_TRAIN, EH_SPLIT_VAL = 0, 1
EH_BUF_GRAD, EH_BUF_THETA, EH_BUF_OPT_M, EH_BUF_OPT_V, EH_BUF_GACC, EH_BUF_BNSTAT, EH_BUF_TCOUNT, EH_BUF_MOMENT = 0, 1, 2, 3, 4, 5, 6, 7

ACTIVATIONS = {"tanh": 0, "sigmoid": 1, "relu": 2, "swish": 3, "identity": 4}
EH_ACT_PER_NET = 5        # MultiNN: net k uses net_activation[k]
OPT_RULES = {"Adam": 0, "AdamW": 1, "RMSProp": 2, "Descent": 3}
TRAINING_LOSSES = {"mse": 0, "rmse": 1, "mae": 2, "nseLoss": 3, "pearsonLoss": 4, "kgeLoss": 5, "pbkgeLoss": 6}
PAR_NEURAL, PAR_GLOBAL, PAR_FIXED = 0, 1, 2


class ModelDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("n_predictors", C.c_int32), ("n_hidden", C.c_int32),
        ("hidden", C.c_int32 * EH_MAX_HIDDEN), ("activation", C.c_int32), ("scale_nn_outputs", C.c_int32), ("input_batchnorm", C.c_int32),
        ("mech", C.c_int32), ("n_params", C.c_int32),
        ("param_kind", C.c_int32 * EH_MAX_PARAMS), ("param_index", C.c_int32 * EH_MAX_PARAMS),
        ("param_default", C.c_float * EH_MAX_PARAMS), ("param_lower", C.c_float * EH_MAX_PARAMS),
        ("param_upper", C.c_float * EH_MAX_PARAMS),
        ("n_forcings", C.c_int32), ("forcing_index", C.c_int32 * EH_MAX_FORC),
        ("n_targets", C.c_int32), ("target_output", C.c_int32 * EH_MAX_TARG),
        ("n_nets", C.c_int32), ("net_n_predictors", C.c_int32 * EH_MAX_NETS), ("net_hidden", (C.c_int32 * EH_MAX_HIDDEN) * EH_MAX_NETS),
        ("net_activation", C.c_int32 * EH_MAX_NETS),
        ("net_depth", C.c_int32 * EH_MAX_NETS),
        ("prog_len", C.c_int32), ("prog_n_const", C.c_int32), ("prog_n_forc", C.c_int32), ("prog_n_out", C.c_int32),
        ("prog_out", C.c_int32 * EH_MAX_PROG_OUT), ("prog_code", C.c_uint32 * EH_MAX_PROG), ("prog_const", C.c_float * EH_MAX_PROG_CONST),
    ]


class TargetMetrics(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("n", "mse", "rmse", "mae", "r2", "nse", "pearson", "kge", "pbkge", "beta", "alpha", "sse")]


LIB_NAME = "libeasyhybrid_hip.so"
# (EASYHYBRID_HIP_LIB: another build of the same library, e.g. one of tools/ps_variants.sh's diagnostic builds)
LIB_PATH = os.environ.get("EASYHYBRID_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

_F = C.POINTER(C.c_float)
_FP = C.POINTER(_F)
_H = C.c_void_p

# name -> (restype, argtypes); every symbol include/easyhybrid_hip.h declares
SIGNATURES = {
    "eh_version": (C.c_int32, []),
    "eh_last_error": (C.c_char_p, [_H]),
    "eh_create": (C.c_int32, [C.POINTER(ModelDesc), C.POINTER(_H)]),
    "eh_destroy": (C.c_int32, [_H]),
    "eh_n_theta": (C.c_int32, [_H, C.POINTER(C.c_int64)]),
    "eh_set_stream": (C.c_int32, [_H, C.c_void_p]),
    "eh_synchronize": (C.c_int32, [_H]),
    "eh_set_data": (C.c_int32, [_H, C.c_int32, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32]),
    "eh_set_params": (C.c_int32, [_H, _F, C.c_int64]),
    "eh_get_params": (C.c_int32, [_H, _F, C.c_int64]),
    "eh_forward": (C.c_int32, [_H, C.c_int32, C.c_int64, C.c_int64, _FP, _FP]),
    "eh_mech_loss_vjp": (C.c_int32, [_H, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                     C.c_void_p, C.c_void_p, _F, _F, C.POINTER(C.c_int64)]),
    "eh_loss_and_grad": (C.c_int32, [_H, C.c_int32, C.POINTER(C.c_int32), C.c_int64, C.c_int64, _F, _F, C.POINTER(C.c_int64)]),
    "eh_get_bn_state": (C.c_int32, [_H, _F, _F, C.c_int64]),
    "eh_set_bn_state": (C.c_int32, [_H, _F, _F, C.c_int64]),
    "eh_opt_init": (C.c_int32, [_H, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "eh_get_opt_state": (C.c_int32, [_H, _F, _F, C.c_int64, _F]),
    "eh_set_opt_state": (C.c_int32, [_H, _F, _F, C.c_int64, _F]),
    "eh_comm_unique_id": (C.c_int32, [C.c_void_p, C.c_int64]),
    "eh_comm_init": (C.c_int32, [_H, C.c_void_p, C.c_int64, C.c_int32, C.c_int32]),
    "eh_comm_init_local": (C.c_int32, [C.POINTER(_H), C.c_int32]),
    "eh_comm_destroy": (C.c_int32, [_H]),
    "eh_comm_group_begin": (C.c_int32, []),
    "eh_comm_group_end": (C.c_int32, []),
    "eh_dp_allreduce": (C.c_int32, [_H, C.c_int32, C.c_int32]),
    "eh_dp_train_step": (C.c_int32, [_H, C.c_int64, C.c_int64, _F]),
    "eh_dp_train_step_group": (C.c_int32, [C.POINTER(_H), C.c_int32, C.POINTER(C.c_int64), C.c_int64, _F]),
    "eh_train_step": (C.c_int32, [_H, C.POINTER(C.c_int32), C.c_int32, C.c_int64, C.c_int64, _F]),
    "eh_train_epoch": (C.c_int32, [_H, C.c_int64, C.c_uint64, C.c_int32, _F, C.POINTER(C.c_int64)]),
    "eh_eval": (C.c_int32, [_H, C.c_int32, C.c_int64, C.c_int64, C.POINTER(TargetMetrics), _FP, _FP]),
    "eh_dp_grad": (C.c_int32, [_H, C.c_int64, C.c_int64]),
    "eh_dp_counts": (C.c_int32, [_H, C.c_int64, C.c_int64]),
    "eh_set_target_shift": (C.c_int32, [_H, C.c_int32, _F, C.c_int64]),
    "eh_set_weight_l2": (C.c_int32, [_H, C.c_float, C.c_int32]),
    "eh_set_weight_l2_coef": (C.c_int32, [_H, C.POINTER(C.c_float), C.c_int64]),
    "eh_graph_begin": (C.c_int32, [_H]),
    "eh_graph_end": (C.c_int32, [_H, C.POINTER(C.c_int32)]),
    "eh_graph_launch": (C.c_int32, [_H, C.c_int32]),
    "eh_dp_apply": (C.c_int32, [_H, _F]),
    "eh_dp_shuffle": (C.c_int32, [_H, C.c_uint64, C.c_int32]),
    "eh_p2p_init": (C.c_int32, [_H, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "eh_p2p_attach": (C.c_int32, [_H, C.c_void_p, C.c_int64]),
    "eh_p2p_selftest": (C.c_int32, [_H, C.c_int32, C.POINTER(C.c_int32)]),
    "eh_p2p_disable": (C.c_int32, [_H]),
    "eh_p2p_init_local": (C.c_int32, [C.POINTER(_H), C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "eh_p2p_check_local": (C.c_int32, [C.POINTER(_H), C.c_int32, C.POINTER(C.c_int32)]),
    "eh_set_bn_shift": (C.c_int32, [_H, _F, C.c_int64]),
    "eh_dp_bn_stats": (C.c_int32, [_H, C.c_int64, C.c_int64]),
    "eh_dp_moments": (C.c_int32, [_H, C.c_int64, C.c_int64, C.c_int32]),
    "eh_dp_fused_step": (C.c_int32, [_H, C.c_int64, C.c_int64, C.POINTER(C.c_int32)]),
    "eh_device_buffer": (C.c_int32, [_H, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "eh_profile_enable": (C.c_int32, [_H, C.c_int32]),
    "eh_profile_read": (C.c_int32, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "eh_profile_samples": (C.c_int32, [_H, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int64)]),
    "eh_jit_status": (C.c_int32, [_H, C.POINTER(C.c_int32), C.c_char_p, C.c_int64]),
    "eh_set_loss_program": (C.c_int32, [_H, C.POINTER(C.c_uint32), C.c_int32, _F, C.c_int32, C.c_int32]),
    "eh_set_target_loss_program": (C.c_int32, [_H, C.c_int32, C.POINTER(C.c_uint32), C.c_int32, _F, C.c_int32, C.c_int32]),
    "eh_debug_stamps": (C.c_int32, [_H, C.POINTER(C.c_uint64), C.c_int32]),
    "eh_set_option": (C.c_int32, [_H, C.c_char_p, C.c_int64]),
    "eh_set_target_losses": (C.c_int32, [_H, C.POINTER(C.c_int32), C.c_int32]),
    "eh_set_target_roles": (C.c_int32, [_H, C.POINTER(C.c_int32), C.c_int32]),
}


def load(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP engine is not built.  Run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C easyhybrid.jl_amd/csrc`.  There is no CPU fallback.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    return lib


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = load()
    return _lib
