"""Shared helpers for the tests: oracle spec <-> package model, standard cases."""
import numpy as np

import easyhybrid_jl_amd as eh
from oracle import hybrid_oracle as ho

MECH_NAME = {"rbq10": "RbQ10", "expo": "Expo_resp_model", "linear": "LinearHM", "expo2pool": "Expo2Pool",
             "rs_components": "Rs_components", "rs_components3f": "Rs_components3F", "fluxpart": "FluxPartModelQ10"}


def register_closure(name, fn, params, forcings, targets):
    """A user closure as mechanistic model `name` on both sides: the package records it when the model is constructed
    (MECH_NAME[name] = the callable); the oracle gets the recorded program as plain data plus the closure itself."""
    from easyhybrid_jl_amd.program import trace
    prog = trace(fn, list(params), list(forcings), list(targets))
    MECH_NAME[name] = fn
    return ho.program_mech(name, prog.as_dict(), fn)


def register_loss(name, fn):
    """A custom training loss on both sides: returns what the package takes (the callable); the oracle gets the recorded
    per-sample program plus the function itself under training-loss kind `name`."""
    from easyhybrid_jl_amd.program import trace_loss
    wrapped = fn
    if name == "relative_sq":                        # (a function that returns per-sample terms: the oracle needs the scalar)
        wrapped = lambda yh, y: np.mean(fn(yh, y))
    ho.loss_program(name, trace_loss(fn).as_dict(), wrapped)
    return fn


def model_from_spec(spec: ho.HybridSpec):
    mm = ho.MECH[spec.mech][0]
    if spec.nets is not None:
        preds = {n: [f"x{i}" for i in rows] for n, (rows, _) in zip(spec.neural, spec.nets)}
        hl = {n: list(h) for n, (_, h) in zip(spec.neural, spec.nets)}
        return eh.constructHybridModel(preds, list(mm.forcings), list(spec.targets), MECH_NAME[spec.mech], dict(spec.parameters),
                                       list(spec.glob), hidden_layers=hl,
                                       activation=spec.activation if spec.net_activations is None else dict(zip(spec.neural, spec.net_activations)),
                                       scale_nn_outputs=spec.scale_nn_outputs, input_batchnorm=getattr(spec, "input_batchnorm", False),
                                       precision=getattr(spec, "precision", "f32"))
    hidden, act = list(spec.hidden), spec.activation
    if getattr(spec, "layer_activations", None) is not None:      # hidden_layers::Chain: the first hidden layer takes the model's activation, the others their own
        la = spec.layer_activations
        act = la[0]
        hidden = eh.Chain(*[eh.Dense(hidden[i], hidden[i + 1], la[i + 1]) for i in range(len(hidden) - 1)]) if len(hidden) > 1 else hidden
    return eh.constructHybridModel([f"x{i}" for i in range(spec.n_pred)], list(mm.forcings), list(spec.targets),
                                   MECH_NAME[spec.mech], dict(spec.parameters), list(spec.neural), list(spec.glob),
                                   hidden_layers=hidden, activation=act,
                                   scale_nn_outputs=spec.scale_nn_outputs, input_batchnorm=getattr(spec, "input_batchnorm", False),
                                   precision=getattr(spec, "precision", "f32"))


def load_engine(spec, theta, X, forcings, targets, split=0, engine=None):
    mm = ho.MECH[spec.mech][0]
    eng = engine or model_from_spec(spec).engine()
    # the parity tests are about the generic kernels and the ones compiled at run time; the kernels specialised AHEAD of time for the
    # canonical descriptors (csrc/eh_spec.hip) have their own tests (tests/test_gpu_headline.py, and every train() front-door test runs them)
    eng.set_option("aot_spec", 0)
    if spec.nets is not None:                         # the per-net predictor matrices stacked row-wise
        X = np.concatenate([X[rows] for rows, _ in spec.nets], axis=0)
    eng.set_data(split, X, [forcings[f] for f in mm.forcings], [targets[t] for t in spec.targets])
    eng.set_params(np.asarray(theta, np.float32))
    return eng


def relerr(a, b):
    """max |a - b| over max |b|: the error of the largest entries (what a norm-wise bound sees)"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def elem_relerr(a, b, floor_frac=1e-3):
    """element-wise relative error max_i |a_i - b_i| / max(|b_i|, floor), floor = floor_frac * max |b|: every entry that is not
    tiny against the largest one has to be right to the stated number of digits on its own"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    floor = max(floor_frac * float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def rbq10_case(B, act="tanh", scale=False, nan_frac=0.0, seed=42, hidden=(16, 16), theta_seed=1):
    spec = ho.rbq10_spec(hidden, act, scale)
    X, f, y = ho.make_synth_rbq10(B, seed, nan_frac)
    X = (X / np.float32(50.0)).astype(np.float32)      # keep activations out of saturation for a sharp test
    theta = ho.init_theta(spec, theta_seed, np.float32)
    return spec, theta, X, f, y
