"""easyhybrid.jl_amd -- MI355X-native engine for the EasyHybrid.jl training-step hot path.

Host-side mirror of the reference interface (constructHybridModel / SingleNNHybridModel / train)
over the C ABI of libeasyhybrid_hip.so.  The directory name contains a dot, so it is loaded through
the root-level shim module `easyhybrid_jl_amd` (import easyhybrid_jl_amd as eh).
"""
from . import _lib
from .engine import EngineError, HybridEngine, PerTarget
from .models import (Chain, Dense, Expo2Pool, Expo_resp_model, FluxPartModelQ10, HybridModel, LinearHM, MECH_REGISTRY, MultiNNHybridModel, ParameterContainer, RbQ10,
                     Rs_components, Rs_components3F, SingleNNHybridModel, build_parameters, constructHybridModel, hard_sigmoid,
                     inv_hard_sigmoid, inv_sigmoid, scale_single_param, scale_single_param_minmax, sigmoid)
from .train import (Adam, AdamW, DataConfig, Descent, EpochSnapshot, RMSProp, TrainConfig, TrainResults, WeightL2,
                    check_training_loss, isbetter, prepare_data, split_data, train, validate_config)
from . import dp, synthetic

EH_SPLIT_TRAIN, EH_SPLIT_VAL = _lib.EH_SPLIT_TRAIN, _lib.EH_SPLIT_VAL
