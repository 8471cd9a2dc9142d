"""CPU oracle for the EasyHybrid training-step hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain NumPy restatement of what the reference computes on its
hot path.  It exists to CHECK the HIP kernels; nothing in the
product path (`easyhybrid.jl_amd/`) may import it.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` use it.

Pinning status
--------------
The reference is pure Julia (Lux + Zygote + Optimisers) and cannot run in the
build container (no `julia`, no network).  The oracle is therefore pinned on
the numeric known-answers the reference's own tests hold for this path
(tests/test_oracle_pins.py):
  * `scale_single_param` = 1.0 / 2.0 at raw 0      (test/test_generic_hybrid_model.jl:109-117)
  * `scale_single_param_minmax` = 0 at mid-range    (test/test_generic_hybrid_model.jl:119-126)
  * MSE [1,2,3,4] vs [1.1,1.9,3.2,3.8] = 0.025; masked [T,T,F,T] = 0.02
                                                    (test/test_loss_fn.jl:6-8,20,90-96)
  * every other metric vs its closed form           (test/test_loss_fn.jl:17-74)
  * multi-target loss = sum_t mean-square           (test/test_compute_loss.jl:69-79)
GRADIENT VALUES, DENSE NUMERICS AND THE ADAM TRAJECTORY ARE *PARITY UNPINNED*
by the reference's own tests (they are smoke tests only,
test/test_autodiff_backend.jl:21-37).  For those the oracle is cross-checked
three ways instead (tests/test_oracle_selfcheck.py): hand VJP (this file) vs
PyTorch-CPU autograd of the same forward (oracle/torch_twin.py) vs central
finite differences in fp64.

What each function follows in the reference (paths relative to /root/reference)
-------------------------------------------------------------------------------
  sigmoid scaling / inverse      src/models/GenericHybridModel.jl:348-365
  flat theta order               src/models/GenericHybridModel.jl:236-256,
                                 src/training/initialization.jl:42-44 (ComponentArray, one leaf)
  MLP chain shape                src/models/NNModels.jl:220-231 (Dense+act ..., last Dense linear)
  forward                        src/models/GenericHybridModel.jl:370-431
  RbQ10 formula                  test/test_split_data_train.jl:36-39, src/models/Respiration_Rb_Q10.jl:39-41
  Expo formula                   projects/ExpoHybrid/ExpoHybridEstim.jl:69-85
  Linear formula                 src/models/LinearHM.jl:61-68
  Rs_components formula          src/models/Rs_components.jl:40-57
  loss                           src/losses/loss_fn.jl:58-179, src/losses/compute_loss.jl:50-66,115-126
  valid mask / empty batch       src/training/train.jl:221-232, src/training/epoch.jl:17-19,35-37
  Adam                           Optimisers.jl `Adam` (third-party, un-vendored; reference default
                                 `Adam(0.01)` at src/config/TrainingConfig.jl:43, beta=(0.9,0.999), eps=1e-8):
                                 m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
                                 theta -= lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps), t = 1,2,...
  AdamW / RMSProp                Optimisers.jl rules of the same names (EasyHybrid.jl:59 re-exports them)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# ----------------------------------------------------------------------------------------------
# activations (Lux Dense applies `act.(W*x .+ b)`; NNModels.jl:225-230)
# ----------------------------------------------------------------------------------------------


def _sigmoid(x):
    # numerically stable logistic; same value as Lux.sigmoid to rounding
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    ex = np.exp(x[~pos])
    out[~pos] = ex / (1.0 + ex)
    return out


def act_fwd(name: str, z):
    if name == "tanh":
        return np.tanh(z)
    if name == "sigmoid":
        return _sigmoid(z)
    if name == "relu":
        return np.maximum(z, 0)
    if name == "swish":
        return z * _sigmoid(z)
    if name == "identity":
        return z.copy()
    raise ValueError(f"unknown activation {name}")


def act_bwd(name: str, z, h):
    """d act / d z given pre-activation z and output h."""
    one = z.dtype.type(1)
    if name == "tanh":
        return one - h * h
    if name == "sigmoid":
        return h * (one - h)
    if name == "relu":
        return (z > 0).astype(z.dtype)
    if name == "swish":
        s = _sigmoid(z)
        return s * (one + z * (one - s))
    if name == "identity":
        return np.ones_like(z)
    raise ValueError(f"unknown activation {name}")


def round_bf16(x):
    """Round to the nearest bfloat16 (ties to even, what v_cvt_pk_bf16_f32 does) and return the value in x's dtype.  A float64
    input goes through float32 first, like the device, where the value rounded is an fp32 number."""
    x = np.asarray(x)
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(x.dtype)


# ----------------------------------------------------------------------------------------------
# mechanistic model registry: forward + hand VJP.  par/frc are dicts name -> array (B,) or scalar.
# ----------------------------------------------------------------------------------------------


@dataclass(frozen=True)
class MechModel:
    name: str
    params: Tuple[str, ...]      # canonical parameter names (kwargs of the Julia function)
    forcings: Tuple[str, ...]    # canonical forcing names
    outputs: Tuple[str, ...]     # outputs that can be used as targets


def _rbq10_fwd(par, frc, dt):
    # reco = rb .* Q10 .^ (0.1f0 .* (ta .- tref)), tref = 15f0   (test_split_data_train.jl:36-39)
    e = dt.type(0.1) * (frc["ta"] - dt.type(15.0))
    p = np.power(par["Q10"], e)
    return {"reco": par["rb"] * p}, {"e": e, "p": p}


def _rbq10_vjp(par, frc, out, aux, dout, dt):
    d = dout["reco"]
    return {"rb": d * aux["p"], "Q10": d * out["reco"] * aux["e"] / par["Q10"]}


def _expo_fwd(par, frc, dt):
    # Resp_obs = Resp0 .* exp.(k .* T)   (ExpoHybridEstim.jl:83)
    ex = np.exp(par["k"] * frc["T"])
    return {"Resp_obs": par["Resp0"] * ex}, {"ex": ex}


def _expo_vjp(par, frc, out, aux, dout, dt):
    d = dout["Resp_obs"]
    return {"Resp0": d * aux["ex"], "k": d * out["Resp_obs"] * frc["T"]}


def _linear_fwd(par, frc, dt):
    # y = alpha .* x .+ beta   (LinearHM.jl:65)
    return {"obs": par["alpha"] * frc["x"] + par["beta"]}, {}


def _linear_vjp(par, frc, out, aux, dout, dt):
    d = dout["obs"]
    return {"alpha": d * frc["x"], "beta": d}


def _expo2pool_fwd(par, frc, dt):
    # build-defined 4-parameter variant of the Expo model (BASELINE.json config 3; not in the
    # reference): two pools, each Resp0_i * exp(k_i * T)
    ea = np.exp(par["ka"] * frc["T"])
    eb = np.exp(par["kb"] * frc["T"])
    return {"Resp_obs": par["R0a"] * ea + par["R0b"] * eb}, {"ea": ea, "eb": eb}


def _expo2pool_vjp(par, frc, out, aux, dout, dt):
    d = dout["Resp_obs"]
    T = frc["T"]
    return {
        "R0a": d * aux["ea"], "ka": d * par["R0a"] * aux["ea"] * T,
        "R0b": d * aux["eb"], "kb": d * par["R0b"] * aux["eb"] * T,
    }


def _rs_fwd(par, frc, dt):
    # R_soil = sum_c Rb_c * Q10_c^(0.1 (T - 15))   (Rs_components.jl:45-55)
    e = dt.type(0.1) * (frc["ta"] - dt.type(15.0))
    aux = {"e": e}
    tot = 0
    for c in ("het", "root", "myc"):
        p = np.power(par[f"Q10_{c}"], e)
        aux[f"p_{c}"] = p
        aux[f"R_{c}"] = par[f"Rb_{c}"] * p
        tot = tot + aux[f"R_{c}"]
    return {"R_soil": tot}, aux


def _rs_vjp(par, frc, out, aux, dout, dt):
    d = dout["R_soil"]
    g = {}
    for c in ("het", "root", "myc"):
        g[f"Rb_{c}"] = d * aux[f"p_{c}"]
        g[f"Q10_{c}"] = d * aux[f"R_{c}"] * aux["e"] / par[f"Q10_{c}"]
    return g


def _rs3f_fwd(par, frc, dt):
    # build-defined (BASELINE.json configs[4]: "3 forcings ... RbQ10-family M", the reference has no such model): the three
    # Rs_components pools (Rs_components.jl:45-55) with the root pool scaled by an irradiance-like forcing and the mycorrhizal
    # pool by a vapour-pressure-deficit-like one:  R_soil = R_het + sw_in R_root + vpd R_myc,  R_c = Rb_c Q10_c^(0.1 (ta - 15))
    e = dt.type(0.1) * (frc["ta"] - dt.type(15.0))
    aux = {"e": e}
    tot = 0
    for c, w in (("het", None), ("root", "sw_in"), ("myc", "vpd")):
        p = np.power(par[f"Q10_{c}"], e)
        wv = dt.type(1) if w is None else frc[w]
        aux[f"p_{c}"] = wv * p
        aux[f"R_{c}"] = par[f"Rb_{c}"] * aux[f"p_{c}"]
        tot = tot + aux[f"R_{c}"]
    return {"R_soil": tot}, aux


def _fluxpart_fwd(par, frc, dt):
    # GPP = sw_in .* RUE ./ 12.011f0 ; RECO = Rb .* Q10 .^ (0.1f0 .* (ta .- 15f0)) ; NEE = RECO .- GPP   (FluxPartModel_Q10_Lux.jl:66-74)
    e = dt.type(0.1) * (frc["TA"] - dt.type(15.0))
    p = np.power(par["Q10"], e)
    gpp = frc["SW_IN"] * par["RUE"] / dt.type(12.011)
    reco = par["Rb"] * p
    return {"NEE": reco - gpp, "GPP": gpp, "RECO": reco}, {"e": e, "p": p}


def _fluxpart_vjp(par, frc, out, aux, dout, dt):
    dr = dout["RECO"] + dout["NEE"]
    dg = dout["GPP"] - dout["NEE"]
    return {"RUE": dg * frc["SW_IN"] / dt.type(12.011), "Rb": dr * aux["p"], "Q10": dr * out["RECO"] * aux["e"] / par["Q10"]}


MECH: Dict[str, Tuple[MechModel, callable, callable]] = {
    "rbq10": (MechModel("rbq10", ("rb", "Q10"), ("ta",), ("reco",)), _rbq10_fwd, _rbq10_vjp),
    "expo": (MechModel("expo", ("Resp0", "k"), ("T",), ("Resp_obs",)), _expo_fwd, _expo_vjp),
    "linear": (MechModel("linear", ("alpha", "beta"), ("x",), ("obs",)), _linear_fwd, _linear_vjp),
    "expo2pool": (MechModel("expo2pool", ("R0a", "ka", "R0b", "kb"), ("T",), ("Resp_obs",)),
                  _expo2pool_fwd, _expo2pool_vjp),
    "rs_components": (MechModel("rs_components",
                                ("Rb_het", "Rb_root", "Rb_myc", "Q10_het", "Q10_root", "Q10_myc"),
                                ("ta",), ("R_soil",)), _rs_fwd, _rs_vjp),
    "rs_components3f": (MechModel("rs_components3f",
                                  ("Rb_het", "Rb_root", "Rb_myc", "Q10_het", "Q10_root", "Q10_myc"),
                                  ("ta", "sw_in", "vpd"), ("R_soil",)), _rs3f_fwd, _rs_vjp),
    "fluxpart": (MechModel("fluxpart", ("RUE", "Rb", "Q10"), ("SW_IN", "TA"), ("NEE", "GPP", "RECO")), _fluxpart_fwd, _fluxpart_vjp),
}



# ----------------------------------------------------------------------------------------------
# user closures (GenericHybridModel.jl:420-425: `mechanistic_model(; forcing..., params...)`, differentiated by Zygote).
# The oracle runs the closure itself on NumPy arrays for the forward values; its VJP is the reverse sweep over the
# recorded straight-line program (plain data: slots 0..7 parameters, 8..11 forcings, 12..27 constants, 28+i instruction
# i; include/easyhybrid_hip.h `eh_prog_op`), which tests/test_program.py checks against central differences of the closure.
# ----------------------------------------------------------------------------------------------
PROG_OPS = ("add", "sub", "mul", "div", "neg", "exp", "log", "pow", "sqrt", "tanh", "sigmoid", "max", "min", "abs", "sin", "cos", "select", "gt")


def program_values(prog: dict, par, frc, dt):
    """All slot values of one evaluation of the program (list indexed by slot; None = unused slot)."""
    val = [None] * (28 + len(prog["code"]))
    for j, n in enumerate(prog["params"]):
        val[j] = np.asarray(par[n], dt)
    for j, n in enumerate(prog["forcings"]):
        val[8 + j] = np.asarray(frc[n], dt)
    for j, c in enumerate(prog["consts"]):
        val[12 + j] = dt.type(c)
    one, zero = dt.type(1), dt.type(0)
    with np.errstate(all="ignore"):
        for i, (op, a, b, c) in enumerate(prog["code"]):
            x, y, z = val[a], val[b], val[c]
            k = PROG_OPS[op]
            if k == "add": r = x + y
            elif k == "sub": r = x - y
            elif k == "mul": r = x * y
            elif k == "div": r = x / y
            elif k == "neg": r = -x
            elif k == "exp": r = np.exp(x)
            elif k == "log": r = np.log(x)
            elif k == "pow": r = np.power(x, y)
            elif k == "sqrt": r = np.sqrt(x)
            elif k == "tanh": r = np.tanh(x)
            elif k == "sigmoid": r = one / (one + np.exp(-x))
            elif k == "max": r = np.maximum(x, y)
            elif k == "min": r = np.minimum(x, y)
            elif k == "abs": r = np.abs(x)
            elif k == "sin": r = np.sin(x)
            elif k == "cos": r = np.cos(x)
            elif k == "select": r = np.where(x > 0, y, z)
            elif k == "gt": r = np.where(x > y, one, zero)
            else: raise ValueError(f"opcode {op}")
            val[28 + i] = np.asarray(r, dt)
    return val


def program_mech(name: str, prog: dict, closure=None):
    """Register a recorded closure as mechanistic model `name`.  With `closure` the forward values come from calling it on
    the arrays (as the reference does); the program then only supplies the tape for the VJP."""
    mm = MechModel(name, tuple(prog["params"]), tuple(prog["forcings"]), tuple(prog["outputs"]))

    def fwd(par, frc, dt):
        val = program_values(prog, par, frc, dt)
        if closure is not None:
            res = closure(**{n: np.asarray(frc[n], dt) for n in prog["forcings"]}, **{n: np.asarray(par[n], dt) for n in prog["params"]})
            out = {o: np.asarray(res[o], dt) for o in prog["outputs"]}
        else:
            out = {o: val[s] for o, s in zip(prog["outputs"], prog["out"])}
        return out, {"val": val}

    def vjp(par, frc, out, aux, dout, dt):
        val = aux["val"]
        B = max([np.size(v) for v in val if v is not None] + [1])
        adj = [np.zeros(B, dt) for _ in val]
        for o, s in zip(prog["outputs"], prog["out"]):
            adj[s] = adj[s] + dout[o]
        one, half = dt.type(1), dt.type(0.5)
        with np.errstate(all="ignore"):
            for i in reversed(range(len(prog["code"]))):
                op, a, b, c = prog["code"][i]
                x, y, r, g = val[a], val[b], val[28 + i], adj[28 + i]
                k = PROG_OPS[op]
                if k == "add": adj[a] = adj[a] + g; adj[b] = adj[b] + g
                elif k == "sub": adj[a] = adj[a] + g; adj[b] = adj[b] - g
                elif k == "mul": adj[a] = adj[a] + g * y; adj[b] = adj[b] + g * x
                elif k == "div": adj[a] = adj[a] + g / y; adj[b] = adj[b] - g * r / y
                elif k == "neg": adj[a] = adj[a] - g
                elif k == "exp": adj[a] = adj[a] + g * r
                elif k == "log": adj[a] = adj[a] + g / x
                elif k == "pow": adj[a] = adj[a] + g * y * r / x; adj[b] = adj[b] + g * r * np.log(x)
                elif k == "sqrt": adj[a] = adj[a] + g * half / r
                elif k == "tanh": adj[a] = adj[a] + g * (one - r * r)
                elif k == "sigmoid": adj[a] = adj[a] + g * r * (one - r)
                elif k == "max": adj[a] = adj[a] + np.where(x >= y, g, 0); adj[b] = adj[b] + np.where(x >= y, 0, g)
                elif k == "min": adj[a] = adj[a] + np.where(x <= y, g, 0); adj[b] = adj[b] + np.where(x <= y, 0, g)
                elif k == "abs": adj[a] = adj[a] + g * np.sign(x)
                elif k == "sin": adj[a] = adj[a] + g * np.cos(x)
                elif k == "cos": adj[a] = adj[a] - g * np.sin(x)
                elif k == "select": adj[b] = adj[b] + np.where(x > 0, g, 0); adj[c] = adj[c] + np.where(x > 0, 0, g)
        return {n: adj[j].astype(dt) for j, n in enumerate(prog["params"])}

    MECH[name] = (mm, fwd, vjp)
    return mm

CUSTOM_LOSS: Dict[str, Tuple[str, Optional[callable]]] = {}


def loss_program(name: str, prog: dict, closure=None):
    """Register a recorded custom training loss `closure(yhat, y) -> mean of per-sample terms` as training-loss kind `name`:
    the value comes from calling the function on the valid samples (as the reference does), the derivative from the reverse
    sweep over the recorded per-sample program (value slots: yhat, y)."""
    program_mech("_loss_" + name, prog, None)
    CUSTOM_LOSS[name] = ("_loss_" + name, closure)


# ----------------------------------------------------------------------------------------------
# model spec  (mirror of constructHybridModel's arguments, GenericHybridModel.jl:89-140)
# ----------------------------------------------------------------------------------------------


@dataclass
class HybridSpec:
    n_pred: int
    hidden: List[int]
    mech: str
    parameters: Dict[str, Tuple[float, float, float]]   # name -> (default, lower, upper)
    neural: List[str]
    glob: List[str]
    targets: List[str] = field(default_factory=list)
    activation: str = "tanh"
    scale_nn_outputs: bool = False
    input_batchnorm: bool = False        # InputBatchNorm(in_dim, affine=false) in front of the chain (NNModels.jl:226)
    # MultiNNHybridModel (GenericHybridModel.jl:142-206,458-530): one MLP with ONE output per neural parameter, each on
    # its own predictor rows.  nets[k] = (rows of X feeding net k, hidden widths of net k); None = SingleNN.
    nets: Optional[List[Tuple[List[int], List[int]]]] = None
    # MultiNN with activation::NamedTuple (GenericHybridModel.jl:168-176): the activation of net k; None = `activation` for all
    net_activations: Optional[List[str]] = None
    # SingleNN built from `hidden_layers::Chain` whose Dense layers carry activations of their own (NNModels.jl:205-211): layer l's; None = `activation`
    layer_activations: Optional[List[str]] = None
    # "f32": the reference's arithmetic (Float32 end to end, src/data/prepare_data.jl:58-60).
    # "bf16_fwd": BASELINE.json configs[4] "bf16 fwd / fp32 accumulate" (NOT a reference mode; build-defined): every Dense
    # product takes its two operands -- weights and the layer's input (predictors, hidden activations) -- rounded to bfloat16
    # and accumulates exactly (fp32 on the device); biases, activations, sigma-scaling, the mechanistic model and the loss stay
    # fp32.  What a layer hands on IS the rounded activation, so the backward pass -- fp32 -- is the exact derivative of that
    # function with round() treated as the identity (straight-through): dW = dZ * bf16(h)^T, dH = bf16(W)^T dZ, and act' taken
    # from the stored (rounded) activation, as a mixed-precision framework that keeps bf16 activations does.
    # "bf16": the usual reading of "bf16 / fp32 accumulate" -- bf16 operands in BOTH passes.  The forward of "bf16_fwd"; in the
    # backward pass every delta (d loss / d pre-activation of a layer, the NN-output one included) is computed in fp32 and rounded
    # to bfloat16 before it enters the two Dense products it feeds: dW = bf16(dZ) * bf16(h)^T, dH = bf16(W)^T bf16(dZ), both
    # accumulated exactly (fp32 on the device).  The bias gradients are the row sums of the same rounded deltas (dZ * 1: a product like the others; rounds 2-4 summed the un-rounded deltas).
    # The deltas are rounded IN THE SCALE THE STEP CARRIES THEM: a one-target model back-propagates the UN-normalised loss (sum of
    # squared / absolute residuals; the division by n, 2 n rmse or sum (y - ybar)^2 is applied to the finished gradient in fp32 --
    # n is only known once the pass is over, and under data parallelism only after the all-reduce), so what is rounded is n times the
    # normalised delta: loss scaling by the batch's own normaliser, independent of how the batch is sharded.  Multi-target models and the
    # two-pass losses carry exact per-target weights through the pass and round the normalised delta.
    precision: str = "f32"

    def act_of(self, k: int, layer: Optional[int] = None) -> str:
        """activation of hidden layer `layer` of net k.  SingleNN with `hidden_layers::Chain` (NNModels.jl:145-219: the reference
        wraps the user's layers as Dense(in, first_h, activation) -> layers... -> Dense(last_h, out)): every hidden layer its own."""
        if self.layer_activations is not None and layer is not None:
            return self.layer_activations[layer]
        return self.activation if self.net_activations is None else self.net_activations[k]

    def __post_init__(self):
        mm = MECH[self.mech][0]
        for n in mm.params:
            if n not in self.parameters:
                raise ValueError(f"parameter table lacks {n}")
        for n in self.neural + self.glob:
            if n not in self.parameters:
                raise AssertionError("neural_param_names ⊆ param_names")   # GenericHybridModel.jl:110
        if not self.targets:
            self.targets = [mm.outputs[0]]
        if self.precision not in ("f32", "bf16_fwd", "bf16"):
            raise ValueError(f"precision {self.precision}")
        if self.precision != "f32" and any(self.act_of(k) == "swish" for k in range(len(self.neural) if self.nets else 1)):
            raise NotImplementedError("bf16_fwd keeps only the rounded activation: swish needs the pre-activation")
        self.fixed = [n for n in self.parameters if n not in self.neural and n not in self.glob]

    # -- sizes / flat layout (a11 in SURVEY section 8a) ---------------------------------------
    @property
    def layer_dims(self) -> List[Tuple[int, int]]:
        dims = [self.n_pred] + list(self.hidden) + [len(self.neural)]
        return [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]   # (out, in)

    @property
    def net_list(self) -> List[Tuple[List[int], List[Tuple[int, int]]]]:
        """[(predictor rows, [(out, in) per Dense layer])] -- one entry for SingleNN, one per neural parameter for MultiNN"""
        if self.nets is None:
            if not self.neural:
                return []              # no neural parameter: the reference builds no network (`NN = Chain()`, GenericHybridModel.jl:112-125)
            return [(list(range(self.n_pred)), self.layer_dims)]
        out = []
        for rows, hidden in self.nets:
            dims = [len(rows)] + list(hidden) + [1]
            out.append((list(rows), [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]))
        return out

    @property
    def n_nn(self) -> int:
        return sum(o * i + o for _, dims in self.net_list for o, i in dims)

    @property
    def n_theta(self) -> int:
        return self.n_nn + len(self.glob)

    def lo(self, n): return self.parameters[n][1]
    def hi(self, n): return self.parameters[n][2]
    def default(self, n): return self.parameters[n][0]


def inv_sigmoid(y):
    return np.log(y / (1 - y))                       # GenericHybridModel.jl:354


def scale_single_param(raw, lo, hi):
    return lo + (hi - lo) * _sigmoid(np.asarray(raw))  # GenericHybridModel.jl:348-352


def scale_single_param_minmax(default, lo, hi):
    return inv_sigmoid((default - lo) / (hi - lo))   # GenericHybridModel.jl:361-365


def unpack(spec: HybridSpec, theta):
    """flat theta -> ([[(W (out,in), b (out,)), ...] per net], raw globals).  Weights are Julia
    column-major (out,in) inside the flat vector (ComponentArray of Lux Dense params); for a
    MultiNN model the nets follow each other in neural_param_names order (GenericHybridModel.jl:259-287)."""
    nets, off = [], 0
    for _, dims in spec.net_list:
        Ws = []
        for o, i in dims:
            W = theta[off:off + o * i].reshape((o, i), order="F"); off += o * i
            b = theta[off:off + o]; off += o
            Ws.append((W, b))
        nets.append(Ws)
    raw = theta[off:off + len(spec.glob)]
    return nets, raw


def pack(spec: HybridSpec, nets, raw, dtype=np.float64):
    parts = []
    for Ws in nets:
        for W, b in Ws:
            parts += [np.asarray(W, dtype).flatten(order="F"), np.asarray(b, dtype)]
    parts.append(np.asarray(raw, dtype).reshape(-1))
    return np.concatenate(parts)


def init_theta(spec: HybridSpec, seed: int, dtype=np.float32):
    """Explicit theta0 for tests/bench: W, b ~ U(+-1/sqrt(fan_in)); global raws start from the
    table default (start_from_default=true, GenericHybridModel.jl:244-249).  Lux's own initialiser
    and Julia's RNG stream are not reproducible here, so parity tests always inject theta."""
    rng = np.random.default_rng(seed)
    nets = []
    for _, dims in spec.net_list:
        Ws = []
        for o, i in dims:
            s = 1.0 / np.sqrt(i)
            Ws.append((rng.uniform(-s, s, (o, i)), rng.uniform(-s, s, (o,))))
        nets.append(Ws)
    raw = [scale_single_param_minmax(np.float32(spec.default(g)), np.float32(spec.lo(g)), np.float32(spec.hi(g)))
           for g in spec.glob]
    return pack(spec, nets, raw, dtype)


# ----------------------------------------------------------------------------------------------
# forward / loss / VJP
# ----------------------------------------------------------------------------------------------


BN_EPS, BN_MOMENTUM = 1e-5, 0.1          # Lux.BatchNorm defaults (epsilon = 1f-5, momentum = 0.1f0)


def bn_init(spec):
    """LuxCore.initialstates of BatchNorm: running_mean = 0, running_var = 1."""
    return {"mean": np.zeros(spec.n_pred), "var": np.ones(spec.n_pred)}


def batchnorm_input(X, bn_state, train_mode: bool, dt):
    """Lux BatchNorm(affine = false) on (features x batch) input.  Train mode: batch statistics
    (biased variance) and the running-statistics update with the unbiased correction m/(m-1);
    test mode: running statistics.  No parameters, and X is data: nothing to back-propagate."""
    if train_mode:
        m = X.shape[1]
        mu = X.mean(axis=1)
        var = X.var(axis=1)
        new = None
        if bn_state is not None:
            corr = m / (m - 1.0) if m > 1 else 1.0
            new = {"mean": (1 - BN_MOMENTUM) * bn_state["mean"] + BN_MOMENTUM * mu,
                   "var": (1 - BN_MOMENTUM) * bn_state["var"] + BN_MOMENTUM * corr * var}
    else:
        mu, var, new = np.asarray(bn_state["mean"]), np.asarray(bn_state["var"]), bn_state
    Xn = ((X - mu[:, None].astype(dt)) / np.sqrt(var[:, None].astype(dt) + dt.type(BN_EPS))).astype(dt)
    return Xn, new


def forward(spec: HybridSpec, theta, X, forcings: Dict[str, np.ndarray], dtype=np.float64, keep=False, bn_state=None, train_mode=True):
    """X is (P, B) like the reference (features x batch).  Returns dict with the mech outputs,
    'parameters' (all physical params) and, if keep, the tape for the VJP.  With
    spec.input_batchnorm the predictors are normalised first (batch statistics in train mode,
    `bn_state` running statistics otherwise); the updated running statistics come back as '_bn'."""
    dt = np.dtype(dtype)
    theta = np.asarray(theta, dt)
    X = np.asarray(X, dt)
    bn_new = None
    if spec.input_batchnorm:
        X, bn_new = batchnorm_input(X, bn_state, train_mode, dt)
    nets, raw = unpack(spec, theta)
    # k1: global params
    glob = {}
    for g, r in zip(spec.glob, raw):
        glob[g] = dt.type(spec.lo(g)) + dt.type(spec.hi(g) - spec.lo(g)) * _sigmoid(r.reshape(1))
    # k2: MLP(s): one chain for SingleNN, one single-output chain per neural parameter for MultiNN
    tapes, outs = [], []
    for k_net, ((rows, _), Ws) in enumerate(zip(spec.net_list, nets)):
        bf = spec.precision in ("bf16_fwd", "bf16")
        h = round_bf16(X[rows]) if bf else X[rows]
        if bf:
            Ws = [(round_bf16(W), b) for W, b in Ws]          # the rounded weights are what forward AND backward multiply by
        zs, hs = [], [h]
        for li, (W, b) in enumerate(Ws):
            z = (W @ h + b[:, None]).astype(dt)
            last = li == len(Ws) - 1
            h = z if last else act_fwd(spec.act_of(k_net, li), z).astype(dt)
            if bf and not last:
                h = round_bf16(h)
            zs.append(z); hs.append(h)
        tapes.append((Ws, zs, hs))
        outs.append(h)
    o = np.concatenate(outs, axis=0) if outs else np.zeros((0, X.shape[1]), dt)      # (K, B)
    # k3: optional sigmoid scaling of NN outputs
    nn = {}
    for k, n in enumerate(spec.neural):
        nn[n] = (dt.type(spec.lo(n)) + dt.type(spec.hi(n) - spec.lo(n)) * _sigmoid(o[k])) if spec.scale_nn_outputs else o[k]
    # k4: fixed
    fixed = {f: np.full(1, spec.default(f), dt) for f in spec.fixed}
    par = {**nn, **glob, **fixed}
    frc = {k: np.asarray(v, dt) for k, v in forcings.items()}
    mm, fwd, _ = MECH[spec.mech]
    out, aux = fwd(par, frc, dt)
    out = {k: v.astype(dt) for k, v in out.items()}
    res = dict(out)
    res["parameters"] = par
    res["_bn"] = bn_new
    if keep:
        res["_tape"] = dict(nets=tapes, raw=raw, o=o, par=par, frc=frc, aux=aux, out=out)
    return res


def valid_mask(y):
    return ~np.isnan(y)                                               # train.jl:221-232


def loss_fn(yhat, y, mask, kind: str):
    """src/losses/loss_fn.jl:58-179 on yhat[mask], y[mask]."""
    a, b = yhat[mask], y[mask]
    if kind == "mse":
        return np.mean((a - b) ** 2)
    if kind == "rmse":
        return np.sqrt(np.mean((a - b) ** 2))
    if kind == "mae":
        return np.mean(np.abs(a - b))
    if kind == "pearson":
        return np.corrcoef(a, b)[0, 1]
    if kind == "r2":
        return 1 - np.sum((b - a) ** 2) / np.sum((b - np.mean(b)) ** 2)
    if kind == "pearsonLoss":
        return 1 - np.corrcoef(a, b)[0, 1]
    if kind == "nseLoss":
        return np.sum((a - b) ** 2) / np.sum((b - np.mean(b)) ** 2)
    if kind == "nse":
        return 1 - np.sum((a - b) ** 2) / np.sum((b - np.mean(b)) ** 2)
    if kind in ("kgeLoss", "kge"):
        r = np.corrcoef(a, b)[0, 1]
        al = np.std(a, ddof=1) / np.std(b, ddof=1)
        be = np.mean(a) / np.mean(b)
        l = np.sqrt((r - 1) ** 2 + (al - 1) ** 2 + (be - 1) ** 2)
        return l if kind == "kgeLoss" else 1 - l
    if kind in ("pbkgeLoss", "pbkge"):
        r = np.corrcoef(a, b)[0, 1]
        be = np.mean(a) / np.mean(b)
        l = np.sqrt((r - 1) ** 2 + (be - 1) ** 2)
        return l if kind == "pbkgeLoss" else 1 - l
    if kind == "β":
        return np.mean(a) / np.mean(b)
    if kind == "α":
        return np.std(a, ddof=1) / np.std(b, ddof=1)
    raise ValueError(kind)


def compute_loss(spec, theta, X, forcings, targets: Dict[str, np.ndarray], dtype=np.float64, kind="mse"):
    """train-mode compute_loss (compute_loss.jl:20-35): agg = sum over targets."""
    res = forward(spec, theta, X, forcings, dtype)
    tot = np.dtype(dtype).type(0)
    for t in spec.targets:
        y = np.asarray(targets[t], dtype)
        tot = tot + loss_fn(res[t], y, valid_mask(y), kind)
    return tot


def weight_mask(spec: HybridSpec):
    """True at the flat-theta positions of the Dense WEIGHT matrices (what weight_l2 walks: leaves named :weight,
    src/utils/extract_weights.jl:69-91); biases and raw global parameters are False."""
    m = np.zeros(spec.n_theta, bool)
    off = 0
    for _, dims in spec.net_list:
        for o, i in dims:
            m[off:off + o * i] = True
            off += o * i + o
    return m


def weight_l2(spec: HybridSpec, theta, lam, normalize=False):
    """lam * weight_l2(ps; normalize): sum (or mean) of the squared Dense weights; returns (value, gradient)."""
    m = weight_mask(spec)
    th = np.asarray(theta)
    c = th.dtype.type(lam) / (th.dtype.type(m.sum()) if normalize else th.dtype.type(1))
    g = np.zeros_like(th)
    g[m] = 2 * c * th[m]
    return c * np.sum(th[m] * th[m]), g


def weight_l2_terms(spec: HybridSpec, theta, terms):
    """several extra-loss terms, each lam * weight_l2(ps or ps.<net>; key, normalize) (src/utils/extract_weights.jl:64-91: the
    walk collects the leaves named `key` -- :weight or :bias -- below the node it is given: the whole tree, or one network of a
    MultiNNHybridModel).  terms: [(lam, normalize, net index or None, "weight" | "bias")].  Returns ([values], gradient)."""
    th = np.asarray(theta)
    g = np.zeros_like(th)
    vals = []
    for lam, normalize, net, key in terms:
        m = np.zeros(spec.n_theta, bool)
        off = 0
        for k, (_, dims) in enumerate(spec.net_list):
            for o, i in dims:
                if net is None or net == k:
                    if key == "weight": m[off:off + o * i] = True
                    else: m[off + o * i:off + o * i + o] = True
                off += o * i + o
        n = int(m.sum())
        c = th.dtype.type(lam) / (th.dtype.type(n) if normalize and n > 0 else th.dtype.type(1))
        g[m] += 2 * c * th[m]
        vals.append(c * np.sum(th[m] * th[m]))
    return vals, g


def _backprop(spec, tp, dout, dt, B, dout_un=None, defer=None):
    """d loss / d mechanistic outputs (`dout`: one (B,) seed per output) -> gradient wrt flat theta: the pullback through the
    mechanistic model, the sigmoid scaling and the MLP(s).  dout_un / defer: the un-normalised seed and the
    factor applied after the pass, for precision = "bf16" (HybridSpec.precision)."""
    dout = dict(dout)
    dout_un = dout_un or {}
    defer = dt.type(1) if defer is None else defer
    mm, _, vjp = MECH[spec.mech]
    for oname in mm.outputs:
        dout.setdefault(oname, np.zeros(B, dt))
    dpar = vjp(tp["par"], tp["frc"], tp["out"], tp["aux"], dout, dt)
    # globals: sum over samples, chain through the sigmoid scaling
    graw = []
    for g, r in zip(spec.glob, tp["raw"]):
        s = _sigmoid(r.reshape(1))[0]
        graw.append(np.sum(dpar[g]) * dt.type(spec.hi(g) - spec.lo(g)) * s * (1 - s))
    # NN outputs
    def nn_output_grads(dp):
        do = np.zeros_like(tp["o"])
        for k, n in enumerate(spec.neural):
            d = np.broadcast_to(dp[n], (B,)).astype(dt)
            if spec.scale_nn_outputs:
                s = _sigmoid(tp["o"][k])
                d = d * dt.type(spec.hi(n) - spec.lo(n)) * s * (1 - s)
            do[k] = d
        return do
    do = nn_output_grads(dpar)
    # precision = "bf16" rounds the deltas in the scale the step carries them (HybridSpec.precision): un-normalised for a one-target
    # model with a one-pass loss -- the chain re-run from the un-normalised seed, as the engine forms it, not do / defer
    bfb = spec.precision == "bf16"
    deferred = bfb and len(spec.targets) == 1 and spec.targets[0] in dout_un
    sc = defer if deferred else dt.type(1)
    if deferred:
        seeds = {oname: np.zeros(B, dt) for oname in mm.outputs}
        seeds[spec.targets[0]] = dout_un[spec.targets[0]]
        do = nn_output_grads(vjp(tp["par"], tp["frc"], tp["out"], tp["aux"], seeds, dt))
    # MLP backward (each net sees the rows of dO that belong to its outputs)
    gnets, k0 = [], 0
    for k_net, (Ws, zs, hs) in enumerate(tp["nets"]):
        kout = Ws[-1][0].shape[0]
        delta = do[k0:k0 + kout]
        k0 += kout
        gWs = []
        # bfb: bf16 operands in the backward products too -- the delta (un-normalised where `deferred`) is rounded where it enters them
        for li in reversed(range(len(Ws))):
            W, b = Ws[li]
            dq = round_bf16(delta) if bfb else delta
            # (the bias gradient is the row sum of the SAME operand -- dZ * 1 -- so "bf16" sums the once-rounded delta, as a framework that
            #  hands a bfloat16 dZ to both reductions does; round 5: the device forms it as a product with a vector of ones)
            gWs.append(((dq @ hs[li].T) * sc, dq.sum(axis=1) * sc))
            if li > 0:
                delta = (W.T @ dq) * act_bwd(spec.act_of(k_net, li - 1), zs[li - 1], hs[li])
        gWs.reverse()
        gnets.append(gWs)
    return pack(spec, gnets, graw, dt)


def vjp_from_output_seed(spec, theta, X, forcings, seeds: Dict[str, np.ndarray], dtype=np.float64, bn_state=None):
    """sum_i seeds[o][i] * d out_o[i] / d theta for given per-sample seeds on the mechanistic outputs (missing outputs: zero) -- the
    pullback a data-parallel shard applies once the global batch statistics have fixed d loss / d yhat_i (tests/test_dp_gloo.py), and
    what an extra loss of the predictions adds to the gradient"""
    dt = np.dtype(dtype)
    res = forward(spec, theta, X, forcings, dtype, keep=True, bn_state=bn_state, train_mode=True)
    return _backprop(spec, res["_tape"], {k: np.asarray(v, dt) for k, v in seeds.items()}, dt, X.shape[1])


def loss_and_grad(spec, theta, X, forcings, targets, dtype=np.float64, kind="mse", bn_state=None, l2=None, agg="sum", extra=None, empty_target="zero"):
    """Training loss (`kind` in mse / rmse / mae / nseLoss / pearsonLoss / kgeLoss / pbkgeLoss, loss_fn.jl:58-174; agg=sum over targets)
    and its gradient wrt flat theta: the hand-derived VJP of SURVEY.md section 8(a).  Returns
    (loss, grad, n_valid per target).

    A target with NO valid sample inside a batch that has some (another target's): the reference evaluates `mean(abs2, yhat[mask] .- y[mask])` on
    the empty selection (loss_fn.jl:61-63) -- 0 / 0 = NaN for the VALUE, while the pullback of a mean over an empty selection scatters nothing
    back: that target adds zero to the GRADIENT and the other targets' terms arrive intact (Zygote: `getindex` adjoint of an empty mask; the
    step's loss value is discarded by run_epoch!, epoch.jl:20).  So the gradient below IS the reference's; for the value, empty_target = "zero"
    (the engine's default: the target contributes 0, the sum of the others is reported) or "nan" (the reference's value; the engine's
    `empty_target_nan` option at the objective seam, eh_loss_and_grad).  Only the batch whose masks are ALL empty is skipped (epoch.jl:17-19,35-37)."""
    dt = np.dtype(dtype)
    res = forward(spec, theta, X, forcings, dtype, keep=True, bn_state=bn_state, train_mode=True)
    tp = res["_tape"]
    B = X.shape[1]
    loss = dt.type(0)
    dout = {}
    nvalid = []
    kinds = list(kind) if isinstance(kind, (list, tuple)) else [kind] * len(spec.targets)      # PerTarget((l_1, ..., l_T)), compute_loss.jl:128-145
    if len(kinds) != len(spec.targets):
        raise AssertionError("Length of targets and PerTarget losses tuple must match")
    defer = dt.type(1)           # one-target, one-pass losses: the factor the engine applies AFTER the pass (what "bf16" rounds is delta / defer)
    dout_un = {}                 # ... and the un-normalised seed d (loss / defer) / d yhat itself, formed the way the engine forms it (no division)
    for t, kind in zip(spec.targets, kinds):
        y = np.asarray(targets[t], dt)
        m = valid_mask(y)
        n = int(m.sum())
        nvalid.append(n)
        d = np.zeros(B, dt)
        if n == 0 and empty_target == "nan":
            loss = loss + dt.type(np.nan)
        if n > 0:
            r = np.where(m, res[t] - np.where(m, y, 0), 0).astype(dt)
            if kind == "mse":
                loss = loss + np.sum(r * r) / dt.type(n)
                d = dt.type(2) * r / dt.type(n)
                defer = dt.type(1) / dt.type(n); dout_un[t] = dt.type(2) * r
            elif kind == "rmse":
                rm = np.sqrt(np.sum(r * r) / dt.type(n))
                loss = loss + rm
                d = r / (dt.type(n) * rm)
                defer = dt.type(1) / (dt.type(2) * dt.type(n) * rm); dout_un[t] = dt.type(2) * r
            elif kind == "mae":
                loss = loss + np.sum(np.abs(r)) / dt.type(n)
                d = np.sign(r) / dt.type(n)
                defer = dt.type(1) / dt.type(n); dout_un[t] = np.sign(r).astype(dt)
            elif kind == "nseLoss":
                yv = y[m]
                D = np.sum((yv - np.mean(yv)) ** 2)
                loss = loss + np.sum(r * r) / D
                d = dt.type(2) * r / D
                defer = dt.type(1) / D; dout_un[t] = dt.type(2) * r
            elif kind in ("pearsonLoss", "kgeLoss", "pbkgeLoss"):          # loss_fn.jl:75-77,105-174 (Statistics.cor / std, n-1 cancels)
                yh, yv = res[t][m].astype(dt), y[m]
                mu_s, mu_o = np.mean(yh), np.mean(yv)
                ds, do_ = yh - mu_s, yv - mu_o
                Suu, Sww, Suw = np.sum(ds * ds), np.sum(do_ * do_), np.sum(ds * do_)
                rr = Suw / np.sqrt(Suu * Sww)
                drr = do_ / np.sqrt(Suu * Sww) - rr * ds / Suu
                if kind == "pearsonLoss":
                    lt, dl = 1 - rr, -drr
                else:
                    alpha, beta = np.sqrt(Suu / Sww), mu_s / mu_o
                    dalpha, dbeta = ds / (alpha * Sww), np.full(n, 1 / (n * mu_o), dt)
                    if kind == "kgeLoss":
                        lt = np.sqrt((rr - 1) ** 2 + (alpha - 1) ** 2 + (beta - 1) ** 2)
                        dl = ((rr - 1) * drr + (alpha - 1) * dalpha + (beta - 1) * dbeta) / lt
                    else:
                        lt = np.sqrt((rr - 1) ** 2 + (beta - 1) ** 2)
                        dl = ((rr - 1) * drr + (beta - 1) * dbeta) / lt
                loss = loss + lt
                d[m] = dl
            elif kind in CUSTOM_LOSS:            # training_loss::Function (loss_fn.jl): mean over the valid samples of l(yhat, y)
                pname, closure = CUSTOM_LOSS[kind]
                _, lfwd, lvjp = MECH[pname]
                yh, yv = np.broadcast_to(res[t], (B,))[m].astype(dt), y[m]      # (an output no per-sample input reaches has one element)
                lpar = {"yhat": yh, "y": yv}
                lout, laux = lfwd(lpar, {}, dt)
                value = np.mean(lout["loss"]) if closure is None else dt.type(closure(yh, yv))       # the function itself gives the value
                loss = loss + value
                d[m] = lvjp(lpar, {}, lout, laux, {"loss": np.full(n, 1.0 / n, dt)}, dt)["yhat"]
                defer = dt.type(1) / dt.type(n)
                du = np.zeros(B, dt); du[m] = lvjp(lpar, {}, lout, laux, {"loss": np.ones(n, dt)}, dt)["yhat"]; dout_un[t] = du
            else:
                raise ValueError(f"training loss {kind}")
        dout[t] = d
    # extra_loss as a function of the PREDICTIONS (compute_loss.jl:31-34; test/test_compute_loss.jl:257-285): entries
    # (output name, recorded per-sample function registered with loss_program(), "sum" | "mean") over ALL samples of the batch -- their
    # derivative joins the seed of the output they read, with the factor `agg` puts on an extra entry
    extra = list(extra or [])
    n_l2 = 0 if l2 is None else (len(l2) if isinstance(l2, list) else 1)
    fac_data = dt.type(1) / dt.type(len(spec.targets) * (1 + n_l2 + len(extra))) if agg == "mean" else dt.type(1)
    fac_extra = dt.type(1) / dt.type(1 + n_l2 + len(extra)) if agg == "mean" else dt.type(1)
    extra_value = dt.type(0)
    if extra and sum(nvalid) > 0:
        mm_ = MECH[spec.mech][0]
        for oname in mm_.outputs:
            dout.setdefault(oname, np.zeros(B, dt))
        for oname, kname, red in extra:
            pname, closure = CUSTOM_LOSS[kname]
            _, lfwd, lvjp = MECH[pname]
            yh = np.broadcast_to(res[oname], (B,)).astype(dt)
            lpar = {"yhat": yh, "y": np.zeros(B, dt)}
            lout, laux = lfwd(lpar, {}, dt)
            w = dt.type(1) if red == "sum" else dt.type(1) / dt.type(B)
            extra_value = extra_value + w * np.sum(lout["loss"])
            # (seeded relative to the data loss's factor: the whole gradient is scaled by fac_data below)
            dout[oname] = dout[oname] + lvjp(lpar, {}, lout, laux, {"loss": np.full(B, w * fac_extra / fac_data, dt)}, dt)["yhat"]
    grad = _backprop(spec, tp, dout, dt, B, dout_un, defer)
    # agg (TrainingConfig.jl:76-77): loss = agg(per-target losses) (compute_loss.jl:50-53); with an extra loss
    # loss = agg([loss, extra entries...]) (compute_loss.jl:31-34).  sum or mean.
    if agg not in ("sum", "mean"):
        raise ValueError(f"agg {agg}")
    loss, grad = loss * fac_data, grad * fac_data
    if sum(nvalid) > 0:
        loss = loss + fac_extra * extra_value
    if l2 is not None and sum(nvalid) > 0:
        if isinstance(l2, list):                      # several terms: agg([loss_value, extra_loss_value...])
            lvs, lg = weight_l2_terms(spec, np.asarray(theta, dt), l2)
            lv = sum(lvs)
        else:
            lv, lg = weight_l2(spec, np.asarray(theta, dt), *l2)
        loss, grad = loss + fac_extra * lv, grad + fac_extra * lg
    return loss, grad, nvalid


def mech_loss_vjp(spec: HybridSpec, theta, o, forcings, targets, dtype=np.float64, kind="mse"):
    """The mechanistic stage on its own: o (K, B) = raw NN outputs of a network evaluated elsewhere -> physical parameters
    (GenericHybridModel.jl:404-411) -> M -> masked MSE summed over targets (compute_loss.jl:50-53, loss_fn.jl:61-63) and its
    pullback to o and to the raw global parameters.  Returns (loss, d loss / d o (K, B), d loss / d raw globals (G,), n_valid per
    target, {target: yhat}).  Same arithmetic as forward() / loss_and_grad() from k3 on.  kind: "mse" or "mae" (loss_fn.jl:61-66)."""
    if kind not in ("mse", "mae"):
        raise ValueError(f"training loss {kind}")
    dt = np.dtype(dtype)
    o = np.asarray(o, dt)
    B = o.shape[1]
    _, raw = unpack(spec, np.asarray(theta, dt))
    glob = {g: dt.type(spec.lo(g)) + dt.type(spec.hi(g) - spec.lo(g)) * _sigmoid(r.reshape(1)) for g, r in zip(spec.glob, raw)}
    nn = {n: (dt.type(spec.lo(n)) + dt.type(spec.hi(n) - spec.lo(n)) * _sigmoid(o[k])) if spec.scale_nn_outputs else o[k]
          for k, n in enumerate(spec.neural)}
    par = {**nn, **glob, **{f: np.full(1, spec.default(f), dt) for f in spec.fixed}}
    frc = {k: np.asarray(v, dt) for k, v in forcings.items()}
    mm, fwd, vjp = MECH[spec.mech]
    out, aux = fwd(par, frc, dt)
    out = {k: np.broadcast_to(v, (B,)).astype(dt) for k, v in out.items()}
    loss, dout, nvalid = dt.type(0), {}, []
    for t in spec.targets:
        y = np.asarray(targets[t], dt)
        m = valid_mask(y)
        n = int(m.sum()); nvalid.append(n)
        d = np.zeros(B, dt)
        if n > 0:
            r = np.where(m, out[t] - np.where(m, y, 0), 0).astype(dt)
            if kind == "mae":
                loss = loss + np.sum(np.abs(r)) / dt.type(n)
                d = np.sign(r) / dt.type(n)
            else:
                loss = loss + np.sum(r * r) / dt.type(n)
                d = dt.type(2) * r / dt.type(n)
        dout[t] = d
    for oname in mm.outputs:
        dout.setdefault(oname, np.zeros(B, dt))
    dpar = vjp(par, frc, out, aux, dout, dt)
    graw = []
    for g, r in zip(spec.glob, raw):
        s = _sigmoid(r.reshape(1))[0]
        graw.append(np.sum(dpar[g]) * dt.type(spec.hi(g) - spec.lo(g)) * s * (1 - s))
    do = np.zeros_like(o)
    for k, n in enumerate(spec.neural):
        d = np.broadcast_to(dpar[n], (B,)).astype(dt)
        if spec.scale_nn_outputs:
            s = _sigmoid(o[k])
            d = d * dt.type(spec.hi(n) - spec.lo(n)) * s * (1 - s)
        do[k] = d
    return float(loss), do, np.asarray(graw, dt), nvalid, {t: out[t] for t in spec.targets}


# ----------------------------------------------------------------------------------------------
# optimiser rules (Optimisers.jl restated; state is a dict)
# ----------------------------------------------------------------------------------------------


def adam_init(n, dtype=np.float32):
    return dict(m=np.zeros(n, dtype), v=np.zeros(n, dtype), t=0)


def adam_step(theta, grad, st, lr=0.01, b1=0.9, b2=0.999, eps=1e-8, weight_decay=0.0):
    """One Optimisers.Adam update in the dtype of theta (fp32 op order as Optimisers.jl:
    mt/(1-bt1) / (sqrt(vt/(1-bt2)) + eps) * eta).  weight_decay != 0 gives Optimisers.AdamW
    with couple=true: theta -= eta*(adam_dir + lambda*theta)."""
    T = theta.dtype.type
    st["t"] += 1
    t = st["t"]
    st["m"] = T(b1) * st["m"] + (T(1) - T(b1)) * grad
    st["v"] = T(b2) * st["v"] + (T(1) - T(b2)) * grad * grad
    bt1 = T(b1) ** t
    bt2 = T(b2) ** t
    upd = st["m"] / (T(1) - bt1) / (np.sqrt(st["v"] / (T(1) - bt2)) + T(eps)) * T(lr)
    if weight_decay:
        upd = upd + T(lr) * T(weight_decay) * theta
    return (theta - upd).astype(theta.dtype)


def train_steps(spec, theta0, X, forcings, targets, batches: Sequence[Tuple[int, int]], lr=0.01, dtype=np.float32, kind="mse",
                bn_state=None, l2=None, agg="sum"):
    """Run Adam over contiguous batches [(first, count), ...]; all-masked batches are skipped
    (epoch.jl:17-19).  Returns (theta, [loss per batch]); with input_batchnorm `bn_state` (a dict
    from bn_init) is updated in place with the running statistics of every batch that ran."""
    theta = np.asarray(theta0, dtype).copy()
    st = adam_init(theta.size, dtype)
    losses = []
    for first, count in batches:
        sl = slice(first, first + count)
        yb = {k: v[sl] for k, v in targets.items()}
        if not any((~np.isnan(v)).any() for v in yb.values()):
            losses.append(float("nan"))          # isemptybatch: the step (and its state update) never runs
            continue
        l, g, nv = loss_and_grad(spec, theta, X[:, sl], {k: v[sl] for k, v in forcings.items()}, yb, dtype, kind, bn_state, l2, agg)
        if spec.input_batchnorm and bn_state is not None:
            _, new = batchnorm_input(np.asarray(X[:, sl], np.float64), bn_state, True, np.dtype(np.float64))
            bn_state.update(new)
        theta = adam_step(theta, g.astype(dtype), st, lr)
        losses.append(float(l))
    return theta, losses


# ----------------------------------------------------------------------------------------------
# evaluation metrics (evaluate_acc -> compute_loss eval branch, compute_loss.jl:36-66)
# ----------------------------------------------------------------------------------------------


def evaluate(spec, theta, X, forcings, targets, loss_types=("mse", "r2"), dtype=np.float64, bn_state=None):
    res = forward(spec, theta, X, forcings, dtype, bn_state=bn_state, train_mode=False)
    out = {}
    for lt in loss_types:
        per = {}
        for t in spec.targets:
            y = np.asarray(targets[t], dtype)
            per[t] = loss_fn(res[t], y, valid_mask(y), lt)
        per["sum"] = sum(per[t] for t in spec.targets)
        out[lt] = per
    return out, {t: res[t] for t in spec.targets}


# ----------------------------------------------------------------------------------------------
# synthetic workloads (distributions of the reference fixtures; NumPy PCG64, not Julia's RNG)
# ----------------------------------------------------------------------------------------------

RBQ10_PARAMS = {"rb": (3.0, 0.0, 13.0), "Q10": (2.0, 1.0, 4.0)}      # test_split_data_train.jl:42-45
EXPO_PARAMS = {"k": (0.01, 0.0, 0.2), "Resp0": (2.0, 0.0, 8.0)}       # ExpoHybridEstim.jl:26-30
EXPO2POOL_PARAMS = {"R0a": (1.0, 0.0, 8.0), "ka": (0.05, 0.0, 0.2), "R0b": (0.5, 0.0, 8.0), "kb": (0.02, 0.0, 0.2)}


def rbq10_spec(hidden=(16, 16), activation="tanh", scale_nn_outputs=False):
    return HybridSpec(2, list(hidden), "rbq10", dict(RBQ10_PARAMS), ["rb"], ["Q10"], ["reco"], activation, scale_nn_outputs)


def make_synth_rbq10(n: int, seed: int = 42, nan_frac: float = 0.0):
    """test/test_split_data_train.jl:15-31 (make_synth_df) with NumPy's generator."""
    rng = np.random.default_rng(seed)
    ta = 10 + 10 * rng.standard_normal(n)
    sw_pot = np.abs(50 + 20 * rng.standard_normal(n))
    dsw_pot = np.concatenate([[0.0], np.diff(sw_pot)])
    rb_true = 3.0 + 0.02 * (sw_pot - sw_pot.mean())
    reco = rb_true * 2.0 ** (0.1 * (ta - 15.0)) + 0.1 * rng.standard_normal(n)
    if nan_frac > 0:
        reco[rng.random(n) < nan_frac] = np.nan
    X = np.stack([sw_pot, dsw_pot]).astype(np.float32)               # (P, N)
    return X, {"ta": ta.astype(np.float32)}, {"reco": reco.astype(np.float32)}


def expo2pool_spec(hidden=(64, 64), activation="tanh", scale_nn_outputs=True):
    return HybridSpec(8, list(hidden), "expo2pool", dict(EXPO2POOL_PARAMS), ["R0a", "ka", "R0b", "kb"], [],
                      ["Resp_obs"], activation, scale_nn_outputs)


def make_synth_expo2pool(n: int, seed: int = 42, nan_frac: float = 0.0):
    """BASELINE.json config 3 inputs: 8 predictors U(0,1), T = U(-10,30) as in
    ExpoHybridEstim.jl:39-46, target from the two-pool formula + 5 % noise."""
    rng = np.random.default_rng(seed)
    X = rng.random((8, n))
    T = rng.random(n) * 40 - 10
    sm = X[0] * 0.8 + 0.1
    R0a = 1.1 * np.exp(-8.0 * (sm - 0.6) ** 2)
    R0b = 0.3 + 0.4 * X[1]
    resp = R0a * np.exp(0.07 * T) + R0b * np.exp(0.02 * T)
    resp = resp + 0.05 * resp.mean() * rng.standard_normal(n)
    if nan_frac > 0:
        resp[rng.random(n) < nan_frac] = np.nan
    return X.astype(np.float32), {"T": T.astype(np.float32)}, {"Resp_obs": resp.astype(np.float32)}


RS6_PARAMS = {**{f"Rb_{c}": (1.0, 0.0, 5.0) for c in ("het", "root", "myc")},
              **{f"Q10_{c}": (2.0 + 0.3 * i, 1.0, 4.0) for i, c in enumerate(("het", "root", "myc"))}}


def c5_spec(hidden=(128, 128), activation="tanh", precision="bf16_fwd", n_pred=32):
    """BASELINE.json configs[4]: MLP [32,128,128,6] -> the six parameters of the three-forcing Rs_components model, all neural,
    sigma-scaled; bf16 forward / fp32 accumulate."""
    return HybridSpec(n_pred, list(hidden), "rs_components3f", dict(RS6_PARAMS), list(RS6_PARAMS), [], ["R_soil"], activation, True,
                      precision=precision)


def make_synth_c5(n: int, seed: int = 42, nan_frac: float = 0.0, n_pred=32):
    """same distributions as easyhybrid.jl_amd/synthetic.py make_synth_fluxnet32_3f (kept apart: the product never imports oracle/)"""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n_pred, n)).astype(np.float32) * 0.5
    ta = (10 + 10 * rng.standard_normal(n)).astype(np.float32)
    sw = (0.2 + rng.random(n)).astype(np.float32)
    vpd = (0.2 + rng.random(n)).astype(np.float32)
    e = 0.1 * (ta - 15.0)
    rb = [1.0 + 0.8 * np.tanh(X[(3 * c) % n_pred] + 0.5 * X[(3 * c + 1) % n_pred]) for c in range(3)]
    w = [1.0, sw, vpd]
    y = sum(w[c] * rb[c] * np.power(1.6 + 0.4 * c, e) for c in range(3)).astype(np.float32)
    y *= (1 + 0.05 * rng.standard_normal(n)).astype(np.float32)
    if nan_frac > 0:
        y[rng.random(n) < nan_frac] = np.nan
    return X, {"ta": ta, "sw_in": sw, "vpd": vpd}, {"R_soil": y}
