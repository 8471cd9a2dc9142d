"""Generates tests/golden/*.npz from the CPU oracle (oracle/hybrid_oracle.py).

The reference (Julia) cannot run in the build container, so these vectors are the ORACLE's outputs,
not the reference's; what pins the oracle to the reference is tests/test_oracle_pins.py.  The
fixtures freeze the oracle so that (a) an accidental change of the oracle is caught on CPU and
(b) the GPU box can check the HIP path against committed numbers.

    python tests/golden/make_golden.py                 # all of them
    python tests/golden/make_golden.py NAME [NAME ...]  # only these (the others stay as committed)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hybrid_oracle as ho  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def spec_dict(spec):
    d = dict(n_pred=spec.n_pred, hidden=list(spec.hidden), mech=spec.mech, parameters={k: list(map(float, v)) for k, v in spec.parameters.items()},
             neural=list(spec.neural), glob=list(spec.glob), targets=list(spec.targets), activation=spec.activation,
             scale_nn_outputs=bool(spec.scale_nn_outputs))
    if spec.nets is not None:              # MultiNN fixtures only (the keys are absent from the older files)
        d["nets"] = [[list(map(int, rows)), list(map(int, hidden))] for rows, hidden in spec.nets]
        d["net_activations"] = None if spec.net_activations is None else list(spec.net_activations)
    return d


ONLY = set(sys.argv[1:])


def emit(name, spec, theta, X, f, y, batch):
    if ONLY and name not in ONLY:
        return
    th64 = theta.astype(np.float64)
    loss, grad, nv = ho.loss_and_grad(spec, th64, X, f, y)
    fw = ho.forward(spec, th64, X, f)
    n = X.shape[1]
    batches = [(i, min(batch, n - i)) for i in range(0, n, batch)]
    th1, l1 = ho.train_steps(spec, theta, X, f, y, batches[:1], dtype=np.float32)
    th_ep, l_ep = ho.train_steps(spec, theta, X, f, y, (batches * 10)[:10], dtype=np.float32)
    out = dict(spec=json.dumps(spec_dict(spec)), theta=theta, X=X, loss=np.float64(loss), grad=grad, n_valid=np.array(nv),
               batch=np.int64(batch), theta_after_1=th1, theta_after_10=th_ep, losses_10=np.array(l_ep, np.float64))
    for k, v in f.items():
        out["forcing_" + k] = v
    for k, v in y.items():
        out["target_" + k] = v
    for t in spec.targets:
        out["yhat_" + t] = fw[t]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", loss, "n_theta", theta.size)


def main():
    k = 0
    for act in ("tanh", "sigmoid", "relu", "swish"):
        for scale in (False, True):
            for B, nan in ((12, 0.0), (64, 0.2), (1024, 0.2)):
                spec = ho.rbq10_spec((16, 16), act, scale)
                X, f, y = ho.make_synth_rbq10(B, 100 + k, nan)
                X = (X / np.float32(50)).astype(np.float32)
                theta = ho.init_theta(spec, 200 + k, np.float32)
                emit(f"rbq10_{act}_{'scaled' if scale else 'raw'}_B{B}", spec, theta, X, f, y, min(B, 256))
                k += 1
    # one batch that is entirely masked in the middle of a run (epoch.jl:17-19: skipped)
    spec = ho.rbq10_spec((16, 16), "tanh", True)
    X, f, y = ho.make_synth_rbq10(192, 7, 0.1)
    X = (X / np.float32(50)).astype(np.float32)
    y["reco"][64:128] = np.nan
    emit("rbq10_allmasked_batch", spec, ho.init_theta(spec, 8, np.float32), X, f, y, 64)
    # BASELINE config 3 shape (build-defined Expo2Pool, [8,64,64,4])
    spec = ho.expo2pool_spec((64, 64), "tanh", True)
    X, f, y = ho.make_synth_expo2pool(512, 9, 0.1)
    emit("expo2pool_64x64_B512", spec, ho.init_theta(spec, 10, np.float32), X, f, y, 256)
    # the reference's own Expo model: 1 neural (Resp0) + 1 global (k), sigmoid, [16,16]   (ExpoHybridEstim.jl:87-98)
    spec = ho.HybridSpec(1, [16, 16], "expo", dict(ho.EXPO_PARAMS), ["Resp0"], ["k"], ["Resp_obs"], "sigmoid", False)
    rng = np.random.default_rng(11)
    T = rng.random(300) * 40 - 10
    SM = rng.random(300) * 0.8 + 0.1
    resp = 1.1 * np.exp(-8.0 * (SM - 0.6) ** 2) * np.exp(0.07 * T)
    resp = resp + rng.standard_normal(300) * 0.05 * resp.mean()
    # a user closure (recorded as a device program): three forcings, two targets, neural + global + fixed parameters
    from tests import closures as cl
    from tests import util
    fn, table, forc = cl.CLOSURES["flux_closure"]
    util.register_closure("flux_closure", fn, list(table), forc, ["nee", "gpp"])
    cspec = ho.HybridSpec(3, [24, 16], "flux_closure", dict(table), ["alpha", "rref"], ["gmax", "e0"], ["nee", "gpp"], "tanh", True)
    crng = np.random.default_rng(21)
    cX = crng.uniform(-1, 1, (3, 400)).astype(np.float32)
    cf = {"sw": crng.uniform(0, 800, 400).astype(np.float32), "ta": crng.uniform(-5, 30, 400).astype(np.float32), "vpd": crng.uniform(0, 30, 400).astype(np.float32)}
    truth = ho.forward(cspec, ho.init_theta(cspec, 22, np.float32).astype(np.float64), cX, cf)
    cy = {}
    for t in ("nee", "gpp"):
        v = truth[t] * (1.0 + 0.05 * crng.normal(size=400))
        v[crng.uniform(size=400) < 0.1] = np.nan
        cy[t] = v.astype(np.float32)
    emit("closure_flux_B400", cspec, ho.init_theta(cspec, 23, np.float32), cX, cf, cy, 200)
    # MultiNNHybridModel, the reference's own constructor case: hidden_layers = (a = [16, 8], d = [8]), activation = (a = tanh,
    # d = sigmoid) (test/test_generic_hybrid_model.jl:346-347) -- two nets of different depth and activation, on RbQ10
    mspec = ho.HybridSpec(3, [1], "rbq10", dict(ho.RBQ10_PARAMS), ["rb", "Q10"], [], ["reco"], "tanh", True,
                          nets=[([0, 1], [16, 8]), ([2], [8])], net_activations=["tanh", "sigmoid"])
    mrng = np.random.default_rng(31)
    mX = mrng.standard_normal((3, 500)).astype(np.float32)
    mf = {"ta": mrng.uniform(0, 30, 500).astype(np.float32)}
    my = mrng.uniform(1, 9, 500).astype(np.float32); my[mrng.random(500) < 0.15] = np.nan
    emit("multinn_depth_act_B500", mspec, ho.init_theta(mspec, 32, np.float32), mX, mf, {"reco": my}, 250)
    emit("expo_ref_B300", spec, ho.init_theta(spec, 12, np.float32), SM[None].astype(np.float32), {"T": T.astype(np.float32)},
         {"Resp_obs": resp.astype(np.float32)}, 100)


if __name__ == "__main__":
    main()
