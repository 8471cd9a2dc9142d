"""Host-side mirror of the reference's model constructor API.

  constructHybridModel(predictors, forcing, targets, mechanistic_model, parameters,
                       neural_param_names, global_param_names; hidden_layers=[32, 32], activation=tanh,
                       scale_nn_outputs=false, input_batchnorm=false, start_from_default=true)
                                                   -- src/models/GenericHybridModel.jl:89-140
  SingleNNHybridModel (alias HybridModel)          -- src/models/GenericHybridModel.jl:44-63
  build_parameters / ParameterContainer            -- src/models/helpers_for_HybridModel.jl:39-52,95-102
  scale_single_param, inv_sigmoid, ...             -- src/models/GenericHybridModel.jl:348-365

The reference accepts any Julia closure as `mechanistic_model`.  The engine has hand-derived device code
for the models of its registry (the reference's own RbQ10, Expo, Linear, Rs_components, FluxPart formulas
plus the build-defined two-pool Expo of BASELINE config 3), selected by name or by the tagged Python
callables below; any other callable `f(**forcings, **params) -> dict` of elementwise arithmetic is called
once with tracer values and handed to the device as a straight-line program (program.py, EH_MECH_PROGRAM),
which the step kernel evaluates and differentiates per sample.  What cannot be recorded raises
NotImplementedError -- never a silent CPU fallback.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib as L

# ---------------------------------------------------------------------------------------------
# mechanistic model registry (names = kwargs of the reference's Julia functions)
# ---------------------------------------------------------------------------------------------


@dataclass(frozen=True)
class MechSpec:
    id: int
    name: str
    params: Tuple[str, ...]
    forcings: Tuple[str, ...]
    outputs: Tuple[str, ...]
    program: Optional[object] = None     # program.Program for a traced closure (id 6)
    fn: Optional[object] = None          # ... and the closure itself (an extra loss over several of its outputs traces it again, program.trace_extra_loss_mixed)


MECH_REGISTRY: Dict[str, MechSpec] = {
    "RbQ10": MechSpec(0, "RbQ10", ("rb", "Q10"), ("ta",), ("reco",)),
    "Expo_resp_model": MechSpec(1, "Expo_resp_model", ("Resp0", "k"), ("T",), ("Resp_obs",)),
    "LinearHM": MechSpec(2, "LinearHM", ("alpha", "beta"), ("x",), ("obs",)),
    "Expo2Pool": MechSpec(3, "Expo2Pool", ("R0a", "ka", "R0b", "kb"), ("T",), ("Resp_obs",)),
    "Rs_components": MechSpec(4, "Rs_components", ("Rb_het", "Rb_root", "Rb_myc", "Q10_het", "Q10_root", "Q10_myc"),
                              ("ta",), ("R_soil",)),
    "Rs_components3F": MechSpec(7, "Rs_components3F", ("Rb_het", "Rb_root", "Rb_myc", "Q10_het", "Q10_root", "Q10_myc"),
                                ("ta", "sw_in", "vpd"), ("R_soil",)),
    "FluxPartModelQ10": MechSpec(5, "FluxPartModelQ10", ("RUE", "Rb", "Q10"), ("SW_IN", "TA"), ("NEE", "GPP", "RECO")),
}


def _tag(name):
    def deco(fn):
        fn.eh_mech = MECH_REGISTRY[name]
        return fn
    return deco


@_tag("RbQ10")
def RbQ10(*, ta, Q10, rb, tref=15.0):
    """test/test_split_data_train.jl:36-39 (documentation of the formula; the device evaluates it)."""
    raise NotImplementedError("mechanistic models are evaluated on the device; this callable is a registry tag")


@_tag("Expo_resp_model")
def Expo_resp_model(*, T, Resp0, k):
    """projects/ExpoHybrid/ExpoHybridEstim.jl:69-85"""
    raise NotImplementedError("registry tag")


@_tag("LinearHM")
def LinearHM(*, x, alpha, beta):
    """src/models/LinearHM.jl:61-68"""
    raise NotImplementedError("registry tag")


@_tag("Expo2Pool")
def Expo2Pool(*, T, R0a, ka, R0b, kb):
    """build-defined: R0a*exp(ka*T) + R0b*exp(kb*T) (BASELINE.json config 3)"""
    raise NotImplementedError("registry tag")


@_tag("Rs_components")
def Rs_components(*, ta, Rb_het, Rb_root, Rb_myc, Q10_het, Q10_root, Q10_myc):
    """src/models/Rs_components.jl:40-57"""
    raise NotImplementedError("registry tag")


@_tag("Rs_components3F")
def Rs_components3F(*, ta, sw_in, vpd, Rb_het, Rb_root, Rb_myc, Q10_het, Q10_root, Q10_myc):
    """build-defined (BASELINE.json config 5, "3 forcings ... RbQ10-family M"): R_het + sw_in*R_root + vpd*R_myc with the pools of
    src/models/Rs_components.jl:45-55, R_c = Rb_c*Q10_c^(0.1(ta-15))"""
    raise NotImplementedError("registry tag")


@_tag("FluxPartModelQ10")
def FluxPartModelQ10(*, SW_IN, TA, RUE, Rb, Q10):
    """src/models/FluxPartModel_Q10_Lux.jl:50-79: GPP = SW_IN*RUE/12.011, RECO = Rb*Q10^(0.1(TA-15)), NEE = RECO - GPP"""
    raise NotImplementedError("registry tag")


def resolve_mech(m, params=None, forcing=None, targets=None) -> MechSpec:
    if isinstance(m, MechSpec):
        return m
    if isinstance(m, str):
        if m in MECH_REGISTRY:
            return MECH_REGISTRY[m]
        raise NotImplementedError(f"mechanistic model {m!r} is not in the device registry {sorted(MECH_REGISTRY)}")
    spec = getattr(m, "eh_mech", None)
    if spec is None:
        if not callable(m) or params is None:
            raise NotImplementedError(f"mechanistic model {m!r}: pass a registry name {sorted(MECH_REGISTRY)} or a callable")
        from .program import trace
        prog = trace(m, params, forcing, targets)      # GenericHybridModel.jl:420-425: f(; forcing..., params...)
        spec = MechSpec(L.EH_MECH_PROGRAM, getattr(m, "__name__", "closure"), prog.params, prog.forcings, prog.outputs, prog, m)
    return spec


# ---------------------------------------------------------------------------------------------
# parameter table + scaling helpers
# ---------------------------------------------------------------------------------------------


class ParameterContainer:
    """(default, lower, upper) table; helpers_for_HybridModel.jl:95-102, GenericHybridModel.jl:22-30."""

    def __init__(self, values: Dict[str, Tuple[float, float, float]]):
        for k, v in values.items():
            if len(v) != 3:
                raise ValueError(f"parameter {k} must be (default, lower, upper)")
        self.values = {k: tuple(np.float32(x) for x in v) for k, v in values.items()}

    def names(self):
        return list(self.values)

    def default(self, n): return self.values[n][0]
    def lower(self, n): return self.values[n][1]
    def upper(self, n): return self.values[n][2]


def build_parameters(parameters, f=None) -> ParameterContainer:
    return parameters if isinstance(parameters, ParameterContainer) else ParameterContainer(dict(parameters))


def sigmoid(x):
    x = np.asarray(x, np.float32)
    return (1.0 / (1.0 + np.exp(-x))).astype(np.float32)


def inv_sigmoid(y):
    y = np.asarray(y, np.float32)
    return np.log(y / (1 - y))                                   # GenericHybridModel.jl:354


def scale_single_param(name, raw_val, hm: ParameterContainer):
    lo, hi = hm.lower(name), hm.upper(name)
    return lo + (hi - lo) * sigmoid(raw_val)                       # GenericHybridModel.jl:348-352


def scale_single_param_minmax(name, hm: ParameterContainer):
    lo, hi = hm.lower(name), hm.upper(name)
    return inv_sigmoid((hm.default(name) - lo) / (hi - lo))       # GenericHybridModel.jl:361-365


def hard_sigmoid(x):
    return np.clip(0.2 * np.asarray(x) + 0.5, 0.0, 1.0)           # GenericHybridModel.jl:9-11


def inv_hard_sigmoid(y):
    return (np.asarray(y) - 0.5) / 0.2                            # GenericHybridModel.jl:16-18


# ---------------------------------------------------------------------------------------------
# model
# ---------------------------------------------------------------------------------------------

_ACT_ALIASES = {"tanh": "tanh", "sigmoid": "sigmoid", "relu": "relu", "swish": "swish", "identity": "identity",
                "σ": "sigmoid", "sigmoid_fast": "sigmoid", "tanh_fast": "tanh"}
_ACT_GAIN = {"tanh": 5.0 / 3.0, "relu": math.sqrt(2.0), "sigmoid": 1.0, "swish": 1.0, "identity": 1.0}


def _act_name(a) -> str:
    name = a if isinstance(a, str) else getattr(a, "__name__", str(a))
    if name not in _ACT_ALIASES:
        raise NotImplementedError(f"activation {name!r} has no device implementation (have {sorted(set(_ACT_ALIASES.values()))})")
    return _ACT_ALIASES[name]


class Dense:
    """Lux `Dense(in => out, activation)` as a DESCRIPTION of a layer (no weights): what `hidden_layers = Chain(...)` is made of."""

    def __init__(self, in_dims: int, out_dims: int, activation="identity"):
        self.in_dims, self.out_dims, self.activation = int(in_dims), int(out_dims), activation

    def __repr__(self):
        return f"Dense({self.in_dims} => {self.out_dims}, {self.activation if isinstance(self.activation, str) else getattr(self.activation, '__name__', self.activation)})"


class Chain:
    """`hidden_layers::Chain` (NNModels.jl:145-219): the user gives the HIDDEN layers only; the reference wraps them as
    Dense(in_dim, first_h, activation) -> layers... -> Dense(last_h, out_dim), first_h / last_h read off the chain's own dimensions."""

    def __init__(self, *layers):
        self.layers = list(layers)


def _hidden_widths(hidden_layers, act: str, per_layer: bool = False):
    """hidden_layers as the vector of widths the device kernels are built around.  A Chain of Dense layers IS such a vector --
    [first_h, out_1, ..., out_n], the first hidden layer (the one the reference puts in front, Dense(in_dim, first_h, activation)) using
    the model's `activation` and layer i the one it was written with (NNModels.jl:205-211).  Other layer types have no kernel and are
    refused with the reason.  per_layer: also return the hidden layers' activations when they are not all the model's (None otherwise);
    without it such a chain is refused (the MultiNN form: its networks differ by network, not by layer)."""
    if isinstance(hidden_layers, Chain):
        ls = hidden_layers.layers
        if not ls:
            raise ValueError("hidden_layers: an empty Chain has no dimensions (NNModels.jl: 'Could not determine input dimension of hidden_layers Chain.')")
        acts = [act]
        for i, l in enumerate(ls):
            if not isinstance(l, Dense):
                raise NotImplementedError(f"hidden_layers Chain: layer {i + 1} is {type(l).__name__}; only Dense layers have a device kernel")
            acts.append(_act_name(l.activation))
            if acts[-1] != act and not per_layer:
                raise NotImplementedError(f"hidden_layers Chain: layer {i + 1} uses {acts[-1]!r}, the model {act!r}: the networks of a MultiNN model "
                                          "apply ONE activation to all their hidden layers (per NETWORK activations exist: activation = {...}; "
                                          "a single network takes an activation per layer)")
            if i and l.in_dims != ls[i - 1].out_dims:
                raise ValueError(f"hidden_layers Chain: layer {i + 1} takes {l.in_dims} inputs, layer {i} gives {ls[i - 1].out_dims}")
        widths = [ls[0].in_dims] + [l.out_dims for l in ls]
        return (widths, acts if any(a != act for a in acts) else None) if per_layer else widths
    if not isinstance(hidden_layers, (list, tuple)):
        raise NotImplementedError(f"hidden_layers of type {type(hidden_layers).__name__}: pass the widths or a Chain of Dense layers")
    widths = [int(h) for h in hidden_layers]
    return (widths, None) if per_layer else widths


@dataclass
class SingleNNHybridModel:
    """Field-for-field mirror of the reference struct (GenericHybridModel.jl:44-63); `NN` is the
    list of Dense (out, in) shapes that prepare_hidden_chain (NNModels.jl:220-231) would build."""
    NN: List[Tuple[int, int]]
    predictors: List[str]
    forcing: List[str]
    targets: List[str]
    mechanistic_model: MechSpec
    parameters: ParameterContainer
    neural_param_names: List[str]
    global_param_names: List[str]
    fixed_param_names: List[str]
    scale_nn_outputs: bool
    start_from_default: bool
    config: dict = field(default_factory=dict)
    # MultiNNHybridModel: NNs[name] = Dense shapes of the single-output net predicting `name`, predictor_sets[name] = its columns
    NNs: Optional[Dict[str, List[Tuple[int, int]]]] = None
    predictor_sets: Optional[Dict[str, List[str]]] = None
    net_activations: Optional[List[str]] = None      # MultiNN with activation::NamedTuple: the activation of net k (None = one for all)
    layer_activations: Optional[List[str]] = None    # single network from `hidden_layers::Chain` whose layers differ: the activation of hidden layer l (None = one for all)

    # -- sizes ---------------------------------------------------------------------------------
    @property
    def hidden_layers(self) -> List[int]:
        return [o for o, _ in self.NN[:-1]]

    @property
    def nets(self) -> List[List[Tuple[int, int]]]:
        if not self.NN and self.NNs is None:
            return []                  # no neural parameter: `NN = Chain()` (GenericHybridModel.jl:112-125), theta holds the raw globals only
        return [self.NN] if self.NNs is None else [self.NNs[k] for k in self.neural_param_names]

    @property
    def n_nn(self) -> int:
        return sum(o * i + o for net in self.nets for o, i in net)

    @property
    def n_theta(self) -> int:
        return self.n_nn + len(self.global_param_names)

    @property
    def activation(self):
        """the name, or {network: name} when the networks differ (the reference keeps the NamedTuple in `config`)"""
        return self.config["activation"]

    def activation_of(self, k: int, layer: Optional[int] = None) -> str:
        if self.layer_activations is not None and layer is not None:
            return self.layer_activations[layer]
        return self.config["activation"] if self.net_activations is None else self.net_activations[k]

    # -- LuxCore.initialparameters analogue (GenericHybridModel.jl:236-256) ----------------------
    def initialparameters(self, rng: Union[int, np.random.Generator] = 0) -> np.ndarray:
        """Flat theta in the reference's ComponentArray order.  Lux >= 1.0 Dense defaults (third
        party, from its docs): weight kaiming_uniform with the activation's gain, bias
        U(+-1/sqrt(fan_in)).  NumPy's stream, not Julia's Xoshiro -- parity tests inject theta."""
        rng = np.random.default_rng(rng) if not isinstance(rng, np.random.Generator) else rng
        parts = []
        for k, net in enumerate(self.nets):
            for li, (o, i) in enumerate(net):
                gain = _ACT_GAIN[self.activation_of(k, li)] if li < len(net) - 1 else 1.0
                bw = gain * math.sqrt(3.0 / i)
                parts.append(rng.uniform(-bw, bw, (o, i)).astype(np.float32).flatten(order="F"))
                parts.append(rng.uniform(-1 / math.sqrt(i), 1 / math.sqrt(i), o).astype(np.float32))
        for g in self.global_param_names:
            if self.start_from_default:
                parts.append(np.asarray([scale_single_param_minmax(g, self.parameters)], np.float32))
            else:
                parts.append(rng.random(1).astype(np.float32))
        return np.concatenate(parts).astype(np.float32)

    def weight_mask(self) -> np.ndarray:
        """True at the flat-theta positions of the Dense weight matrices (the leaves weight_l2 sums, src/utils/extract_weights.jl:69-91)"""
        m = np.zeros(self.n_theta, bool)
        off = 0
        for net in self.nets:
            for o, i in net:
                m[off:off + o * i] = True
                off += o * i + o
        return m

    def l2_mask(self, net=None, key: str = "weight") -> np.ndarray:
        """the flat-theta positions weight_l2(ps or ps.<net>; key) walks (extract_weights.jl:69-91): the leaves named `key`
        (:weight or :bias) of every network, or of the network predicting `net` (a MultiNNHybridModel's `ps.<net>`)"""
        if key not in ("weight", "bias"):
            raise ValueError(f"weight_l2: key = {key!r} (Dense layers have :weight and :bias leaves)")
        names = [None] if self.NNs is None else list(self.neural_param_names)
        if net is not None and net not in names:
            raise KeyError(f"weight_l2: no network named {net!r} (networks: {[n for n in names if n is not None]})")
        m = np.zeros(self.n_theta, bool)
        off = 0
        for name, dims in zip(names, self.nets):
            for o, i in dims:
                if net is None or net == name:
                    if key == "weight": m[off:off + o * i] = True
                    else: m[off + o * i:off + o * i + o] = True
                off += o * i + o
        return m

    def l2_coefficients(self, terms) -> np.ndarray:
        """terms [WeightL2...] -> one coefficient per flat-theta entry: sum of the terms == sum_i coef[i] * theta_i^2 (eh_set_weight_l2_coef)"""
        c = np.zeros(self.n_theta, np.float64)
        for t in terms:
            m = self.l2_mask(t.net, t.key)
            n = int(m.sum())
            c[m] += float(t.lam) / (n if t.normalize and n > 0 else 1)
        return c.astype(np.float32)

    def unpack(self, theta: np.ndarray):
        """flat theta -> (ps = [(weight (out,in), bias)...], {global: raw})"""
        off, nets = 0, []
        for net in self.nets:
            layers = []
            for o, i in net:
                W = theta[off:off + o * i].reshape((o, i), order="F"); off += o * i
                b = theta[off:off + o]; off += o
                layers.append((W, b))
            nets.append(layers)
        glob = {g: theta[off + j:off + j + 1] for j, g in enumerate(self.global_param_names)}
        return ((nets[0] if nets else []) if self.NNs is None else dict(zip(self.neural_param_names, nets))), glob

    # -- C descriptor ----------------------------------------------------------------------------
    def to_desc(self, device: int = 0, extra_outputs=(), mech: Optional[MechSpec] = None) -> L.ModelDesc:
        ms = mech if mech is not None else self.mechanistic_model      # (mech: the model's closure with the extra loss's entries as outputs of their own)
        d = L.ModelDesc()
        d.struct_size = __import__("ctypes").sizeof(L.ModelDesc)
        d.device = device
        d.n_predictors = len(self.predictors)
        hl = self.hidden_layers
        if not (1 if self.NN else 0) <= len(hl) <= L.EH_MAX_HIDDEN:
            raise NotImplementedError(f"{len(hl)} hidden layers (device kernels: 1..{L.EH_MAX_HIDDEN})")
        d.n_hidden = len(hl)
        for k, w in enumerate(hl):
            d.hidden[k] = w
        if self.NNs is not None:
            if len(self.NNs) > L.EH_MAX_NETS:
                raise NotImplementedError(f"{len(self.NNs)} neural networks (device limit {L.EH_MAX_NETS})")
            d.n_nets = len(self.NNs)
            for k, name in enumerate(self.neural_param_names):
                net = self.NNs[name]
                d.net_n_predictors[k] = net[0][1]
                d.net_depth[k] = len(net) - 1
                for l, (o, _) in enumerate(net[:-1]):
                    d.net_hidden[k][l] = o
        if self.net_activations is not None:       # per-net activations: kernels compiled at run time around the descriptor
            d.activation = L.EH_ACT_PER_NET
            for k, a in enumerate(self.net_activations):
                d.net_activation[k] = L.ACTIVATIONS[a]
        elif self.layer_activations is not None:   # an activation per hidden layer (n_nets = 0): net_activation[l] is layer l's
            d.activation = L.EH_ACT_PER_NET
            for l, a in enumerate(self.layer_activations):
                d.net_activation[l] = L.ACTIVATIONS[a]
        else:
            d.activation = L.ACTIVATIONS[self.activation]
        d.scale_nn_outputs = int(self.scale_nn_outputs)
        d.input_batchnorm = int(bool(self.config.get("input_batchnorm", False)))
        d.mech = ms.id
        d.n_params = len(ms.params)
        for j, p in enumerate(ms.params):
            if p in self.neural_param_names:
                d.param_kind[j], d.param_index[j] = L.PAR_NEURAL, self.neural_param_names.index(p)
            elif p in self.global_param_names:
                d.param_kind[j], d.param_index[j] = L.PAR_GLOBAL, self.global_param_names.index(p)
            else:
                d.param_kind[j], d.param_index[j] = L.PAR_FIXED, 0
            d.param_default[j] = self.parameters.default(p)
            d.param_lower[j] = self.parameters.lower(p)
            d.param_upper[j] = self.parameters.upper(p)
        d.n_forcings = len(self.forcing)
        for f, name in enumerate(ms.forcings):
            d.forcing_index[f] = self.forcing.index(name)
        # (extra_outputs: entries of an extra loss that is a function of the predictions ride on targets of their own, behind the data
        #  targets -- include/easyhybrid_hip.h: eh_set_target_roles; each observes the output its entry reads)
        all_t = list(self.targets) + list(extra_outputs)
        if len(all_t) > L.EH_MAX_TARG:
            raise NotImplementedError(f"{len(self.targets)} targets + {len(extra_outputs)} extra-loss entries of the predictions: the device holds {L.EH_MAX_TARG} in all")
        d.n_targets = len(all_t)
        for t, name in enumerate(all_t):
            d.target_output[t] = ms.outputs.index(name)
        if ms.program is not None:
            pg = ms.program
            d.prog_len, d.prog_n_const, d.prog_n_forc, d.prog_n_out = len(pg.code), len(pg.consts), len(pg.forcings), len(pg.out)
            for i, w in enumerate(pg.words()):
                d.prog_code[i] = w
            for i, c in enumerate(pg.consts):
                d.prog_const[i] = c
            for i, o in enumerate(pg.out):
                d.prog_out[i] = o
        return d

    def engine(self, device: int = 0, extra_fn=None):
        """precision (a build extension, BASELINE.json config 5; the reference is Float32 end to end; csrc/eh_wide_bf16.hpp):
        "bf16_fwd" = Dense products of the forward pass on bf16 operands with fp32 accumulation, fp32-exact backward;
        "bf16" = bf16 operands in both passes (every backward delta rounded to bf16 once), fp32 accumulation."""
        from .engine import HybridEngine
        entries, mech = [], None
        if extra_fn is not None:                      # extra_loss(yhat[, ps]) of the predictions: recorded, one more target per entry
            from .program import trace_extra_loss, trace_extra_loss_mixed
            ms = self.mechanistic_model
            try:
                entries = trace_extra_loss(extra_fn, list(self.targets))
            except (NotImplementedError, KeyError, TypeError) as e:
                # an entry over several predictions, over outputs that are no targets, or over predictions and global parameters
                # (compute_loss.jl:31-34): possible where the mechanistic model is a recorded closure -- its per-sample expression
                # becomes one more output of the model's program
                if ms.fn is None:
                    raise NotImplementedError(f"extra_loss: {e} -- entries that mix predictions (or read global parameters) need the mechanistic model "
                                              f"as a Python closure (the recorder adds them to its program); {ms.name!r} is a built-in of the device registry") from e
                bounds = {p: (float(self.parameters.lower(p)), float(self.parameters.upper(p))) for p in self.global_param_names}
                prog, entries = trace_extra_loss_mixed(ms.fn, extra_fn, list(ms.params), list(self.forcing), list(self.targets), list(self.global_param_names), bounds)
                mech = MechSpec(ms.id, ms.name, prog.params, prog.forcings, prog.outputs, prog, ms.fn)
        eng = HybridEngine(self.to_desc(device, [e[1] for e in entries], mech), len(self.mechanistic_model.params), self.targets,
                           list(self.mechanistic_model.params), n_pseudo=len(entries))
        if entries:
            eng.set_extra_entries(entries)
        prec = self.config.get("precision", "f32")
        if prec not in ("f32", "bf16_fwd", "bf16"):
            raise ValueError(f"precision {prec!r}: 'f32', 'bf16_fwd' or 'bf16'")
        if prec != "f32":
            eng.set_option("precision", 1 if prec == "bf16_fwd" else 2)
        return eng


HybridModel = SingleNNHybridModel     # spelling used by BASELINE.json's north_star


def _construct_multi(predictors: Dict[str, Sequence[str]], forcing, targets, mechanistic_model, parameters, global_param_names, *,
                     hidden_layers, activation, scale_nn_outputs, input_batchnorm, start_from_default, **kwargs):
    """MultiNNHybridModel: one single-output MLP per key of `predictors` (= neural parameter) on its own predictor set."""
    parameters = build_parameters(parameters, mechanistic_model)
    ms = resolve_mech(mechanistic_model, parameters.names(), list(forcing), list(targets))
    all_names = parameters.names()
    neural = list(predictors)
    glob = list(global_param_names or [])
    if not all(n in all_names for n in neural):
        raise AssertionError("neural_param_names ⊆ param_names")
    net_acts = None
    if isinstance(activation, dict):               # activation::NamedTuple (GenericHybridModel.jl:168-176)
        if not isinstance(hidden_layers, dict):    # the reference reads activation[nn_name] only next to hidden_layers[nn_name]
            raise TypeError("activation given per network needs hidden_layers given per network as well")
        net_acts = [_act_name(activation[k]) for k in neural]
        act = net_acts[0]
        if len(set(net_acts)) == 1:
            net_acts = None
    else:
        act = _act_name(activation)
    acts_k = dict(zip(neural, net_acts)) if net_acts is not None else {k: act for k in neural}
    hl = ({k: _hidden_widths(hidden_layers[k], acts_k[k]) for k in neural} if isinstance(hidden_layers, dict)
          else {k: _hidden_widths(hidden_layers, acts_k[k]) for k in neural})
    hidden_layers = hl if isinstance(hidden_layers, dict) else next(iter(hl.values()), [])
    if any(len(v) < 1 for v in hl.values()):
        raise NotImplementedError("a network without a hidden layer")
    for p in ms.params:
        if p not in all_names:
            raise ValueError(f"mechanistic model {ms.name} needs parameter {p!r}; the table has {all_names}")
    forcing, targets = list(forcing), list(targets)
    for f in ms.forcings:
        if f not in forcing:
            raise ValueError(f"mechanistic model {ms.name} needs forcing {f!r}; got {forcing}")
    for t in targets:
        if t not in ms.outputs:
            raise ValueError(f"target {t!r} is not an output of {ms.name} {ms.outputs}")
    NNs, flat_pred = {}, []
    for k in neural:
        preds = list(predictors[k])
        if not preds:
            raise NotImplementedError("a network without predictors")
        dims = [len(preds)] + [int(w) for w in hl[k]] + [1]
        NNs[k] = [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]
        flat_pred += preds                                   # the per-net predictor matrices stacked row-wise
    # hidden_layers::NamedTuple may give the nets different depths (test/test_generic_hybrid_model.jl:346): in the block-diagonal
    # envelope a shallower net is carried to the output layer by identity blocks as wide as its last hidden layer
    nl = max(len(v) for v in hl.values())
    tot = [sum(hl[k][min(l, len(hl[k]) - 1)] for k in neural) for l in range(nl)]
    NN = [(a, b) for a, b in zip(tot + [len(neural)], [len(flat_pred)] + tot)]     # the block-diagonal envelope
    fixed = [n for n in all_names if n not in neural and n not in glob]
    config = dict(hidden_layers=hidden_layers, activation=act if net_acts is None else dict(zip(neural, net_acts)),
                  scale_nn_outputs=scale_nn_outputs, input_batchnorm=input_batchnorm, start_from_default=start_from_default, **kwargs)
    return SingleNNHybridModel(NN, flat_pred, forcing, targets, ms, parameters, neural, glob, fixed, bool(scale_nn_outputs),
                               bool(start_from_default), config, NNs=NNs, predictor_sets={k: list(v) for k, v in predictors.items()},
                               net_activations=net_acts)


MultiNNHybridModel = SingleNNHybridModel      # one class serves both; `.NNs is not None` marks the multi-network form


def constructHybridModel(predictors, forcing: Sequence[str], targets: Sequence[str], mechanistic_model,
                         parameters, neural_param_names: Sequence[str] = None, global_param_names: Sequence[str] = None, *,
                         hidden_layers: Sequence[int] = (32, 32), activation="tanh", scale_nn_outputs: bool = False,
                         input_batchnorm: bool = False, start_from_default: bool = True, **kwargs) -> SingleNNHybridModel:
    """GenericHybridModel.jl:89-140 (Vector{Symbol} predictors form)."""
    if isinstance(predictors, dict):
        # NamedTuple-predictors form (GenericHybridModel.jl:142-206): here `neural_param_names` is the reference's
        # `global_param_names` argument position; accept both call shapes
        return _construct_multi(predictors, forcing, targets, mechanistic_model, parameters,
                                global_param_names if global_param_names is not None else neural_param_names,
                                hidden_layers=hidden_layers, activation=activation, scale_nn_outputs=scale_nn_outputs,
                                input_batchnorm=input_batchnorm, start_from_default=start_from_default, **kwargs)
    parameters = build_parameters(parameters, mechanistic_model)
    ms = resolve_mech(mechanistic_model, parameters.names(), list(forcing), list(targets))
    all_names = parameters.names()
    if not all(n in all_names for n in neural_param_names):
        raise AssertionError("neural_param_names ⊆ param_names")                      # GenericHybridModel.jl:110
    predictors, forcing, targets = list(predictors), list(forcing), list(targets)
    neural_param_names, global_param_names = list(neural_param_names), list(global_param_names)
    no_nn = len(predictors) == 0 or len(neural_param_names) == 0          # "if empty predictors do not construct NN" (GenericHybridModel.jl:112-125)
    if no_nn and neural_param_names:
        # the reference builds `NN = Chain()` here and its forward then fails on the missing NN outputs; said up front instead
        raise ValueError("neural_param_names given but no predictors: a neural parameter needs a network input")
    if no_nn and not global_param_names:
        raise ValueError("a model with neither neural nor global parameters has nothing to train")
    for p in ms.params:
        if p not in all_names:
            raise ValueError(f"mechanistic model {ms.name} needs parameter {p!r}; the table has {all_names}")
    extra = [n for n in all_names if n not in ms.params]
    if extra:
        raise ValueError(f"parameters {extra} are not arguments of {ms.name}{ms.params}")
    for f in ms.forcings:
        if f not in forcing:
            raise ValueError(f"mechanistic model {ms.name} needs forcing {f!r}; got {forcing}")
    for t in targets:
        if t not in ms.outputs:
            raise ValueError(f"target {t!r} is not an output of {ms.name} {ms.outputs}")
    act = _act_name(activation)
    hidden_layers, layer_acts = _hidden_widths(hidden_layers, act, per_layer=True)
    dims = [len(predictors)] + hidden_layers + [len(neural_param_names)]
    NN = [] if no_nn else [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]
    if no_nn:
        input_batchnorm = False                                  # (nothing to normalise: the predictors feed no network)
    fixed = [n for n in all_names if n not in neural_param_names and n not in global_param_names]    # :127
    config = dict(hidden_layers=list(hidden_layers), activation=act, scale_nn_outputs=scale_nn_outputs,
                  input_batchnorm=input_batchnorm, start_from_default=start_from_default, **kwargs)
    if layer_acts is not None and not no_nn:
        config["layer_activations"] = list(layer_acts)
    return SingleNNHybridModel(NN, predictors, forcing, targets, ms, parameters, neural_param_names, global_param_names,
                               fixed, bool(scale_nn_outputs), bool(start_from_default), config,
                               layer_activations=None if no_nn else layer_acts)
