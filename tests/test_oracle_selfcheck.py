"""The oracle's hand VJP against two independent derivations (torch autograd, central finite
differences), the plain-C port against the NumPy oracle, and the committed golden fixtures
against a fresh oracle run.  Gradients / Adam are PARITY UNPINNED by the reference's own tests
(SURVEY.md section 8c); this three-way agreement is what stands in for them."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import hybrid_oracle as ho
from oracle import torch_twin as tt

CASES = [(a, s) for a in ("tanh", "sigmoid", "relu", "swish") for s in (False, True)]


def _case(act, scale, B=96, nan=0.2):
    spec = ho.rbq10_spec((16, 16), act, scale)
    X, f, y = ho.make_synth_rbq10(B, 5, nan)
    X = X / 50
    return spec, ho.init_theta(spec, 6, np.float64), X, f, y


@pytest.mark.parametrize("act,scale", CASES)
def test_hand_vjp_matches_autograd(act, scale):
    spec, th, X, f, y = _case(act, scale)
    l, g, _ = ho.loss_and_grad(spec, th, X, f, y)
    l2, g2 = tt.loss_and_grad(spec, th, X, f, y)
    assert l == pytest.approx(l2, rel=1e-12)
    assert np.max(np.abs(g - g2)) <= 1e-11 * np.max(np.abs(g2))


@pytest.mark.parametrize("kind", ["rmse", "mae", "nseLoss"])
def test_other_training_losses_vjp(kind):
    # loss_fn.jl:58-86: the hand VJP of each supported training loss against autograd and its closed form
    spec, th, X, f, y = _case("tanh", True, B=150)
    l, g, _ = ho.loss_and_grad(spec, th, X, f, y, kind=kind)
    l2, g2 = tt.loss_and_grad(spec, th, X, f, y, kind=kind)
    yh = ho.forward(spec, th, X, f)["reco"]
    assert l == pytest.approx(ho.loss_fn(yh, y["reco"].astype(np.float64), ho.valid_mask(y["reco"]), kind), rel=1e-12)
    assert l == pytest.approx(l2, rel=1e-12) and np.max(np.abs(g - g2)) <= 1e-11 * np.max(np.abs(g2))


@pytest.mark.parametrize("act,scale", [("tanh", True), ("sigmoid", False), ("swish", True)])
def test_hand_vjp_matches_finite_differences(act, scale):
    spec, th, X, f, y = _case(act, scale, B=40)
    _, g, _ = ho.loss_and_grad(spec, th, X, f, y)
    rng = np.random.default_rng(0)
    for k in rng.choice(th.size, 25, replace=False):
        e = np.zeros_like(th); e[k] = 1e-6
        fd = (ho.compute_loss(spec, th + e, X, f, y) - ho.compute_loss(spec, th - e, X, f, y)) / 2e-6
        assert fd == pytest.approx(g[k], rel=2e-5, abs=1e-8)


@pytest.mark.parametrize("mech", ["rbq10", "expo2pool", "fluxpart"])
def test_mech_stage_alone_is_the_tail_of_the_full_path(mech):
    """ho.mech_loss_vjp (the oracle of eh_mech_loss_vjp) fed with the NN outputs of the full forward: same loss, same gradient of
    the globals, and d loss / d o pushed through the network's own backward gives the full gradient (chain rule); d loss / d o
    itself against central differences of the loss in o."""
    tabs = {"rbq10": (dict(ho.RBQ10_PARAMS), ["rb"], ["Q10"], None), "expo2pool": (dict(ho.EXPO2POOL_PARAMS), ["R0a", "R0b"], ["ka", "kb"], None),
            "fluxpart": ({"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}, ["RUE", "Rb"], ["Q10"], ["NEE", "RECO"])}
    tab, neural, glob, targets = tabs[mech]
    mm = ho.MECH[mech][0]
    targets = targets or [mm.outputs[0]]
    spec = ho.HybridSpec(3, [8, 8], mech, tab, neural, glob, targets, "tanh", True)
    rng = np.random.default_rng(8)
    B = 70
    X = rng.standard_normal((3, B))
    f = {k: rng.uniform(1, 25, B) for k in mm.forcings}
    y = {t: rng.uniform(0.5, 6, B) for t in targets}
    for v in y.values():
        v[rng.random(B) < 0.2] = np.nan
    th = ho.init_theta(spec, 9, np.float64)
    res = ho.forward(spec, th, X, f, keep=True)
    o = res["_tape"]["o"]
    l0, g0, nv0 = ho.loss_and_grad(spec, th, X, f, y)
    l, do, gg, nv, yh = ho.mech_loss_vjp(spec, th, o, f, y)
    assert nv == nv0 and l == pytest.approx(l0, rel=1e-13)
    assert np.allclose(gg, g0[spec.n_nn:], rtol=1e-12, atol=1e-15)
    for t in targets:
        assert np.allclose(yh[t], res[t], rtol=1e-13)
    # the network's backward applied to do: last layer first
    (Ws, zs, hs), = res["_tape"]["nets"]
    delta, gW = do, []
    for li in reversed(range(len(Ws))):
        W, b = Ws[li]
        gW.append((delta @ hs[li].T, delta.sum(axis=1)))
        if li > 0:
            delta = (W.T @ delta) * ho.act_bwd("tanh", zs[li - 1], hs[li])
    gW.reverse()
    assert np.allclose(ho.pack(spec, [gW], gg, np.float64), g0, rtol=1e-11, atol=1e-14)
    for _ in range(10):
        k, i = int(rng.integers(o.shape[0])), int(rng.integers(B))
        e = np.zeros_like(o); e[k, i] = 1e-6
        fd = (ho.mech_loss_vjp(spec, th, o + e, f, y)[0] - ho.mech_loss_vjp(spec, th, o - e, f, y)[0]) / 2e-6
        assert fd == pytest.approx(do[k, i], rel=2e-5, abs=2e-8)


def test_nets_of_different_depth_vjp():
    # hidden_layers = (a = [16, 8], d = [8]) (test/test_generic_hybrid_model.jl:346): theta holds every net with its own layers
    nets = [([0, 1], [16, 8]), ([2], [8])]
    spec = ho.HybridSpec(3, [1], "rbq10", dict(ho.RBQ10_PARAMS), ["rb", "Q10"], [], ["reco"], "tanh", True, nets=nets, net_activations=["tanh", "sigmoid"])
    assert spec.n_theta == (32 + 16 + 128 + 8 + 8 + 1) + (8 + 8 + 8 + 1)
    rng = np.random.default_rng(3)
    B = 50
    X = rng.standard_normal((3, B)); f = {"ta": rng.uniform(0, 30, B)}; y = {"reco": rng.uniform(1, 9, B)}
    th = ho.init_theta(spec, 4, np.float64)
    l, g, _ = ho.loss_and_grad(spec, th, X, f, y)
    l2, g2 = tt.loss_and_grad(spec, th, X, f, y)
    assert l == pytest.approx(l2, rel=1e-12) and np.max(np.abs(g - g2)) <= 1e-11 * np.max(np.abs(g2))


@pytest.mark.parametrize("acts", [["tanh", "swish"], ["relu", "sigmoid"], ["swish", "identity"]])
def test_per_net_activations_vjp(acts):
    # MultiNN with activation::NamedTuple (GenericHybridModel.jl:168-176): hand VJP against autograd and central differences,
    # and the forward against the two nets evaluated one by one with their own activation
    nets = [([0, 1], [8, 6]), ([1, 2], [5, 7])]
    spec = ho.HybridSpec(3, [1], "rbq10", dict(ho.RBQ10_PARAMS), ["rb", "Q10"], [], ["reco"], "tanh", True, nets=nets, net_activations=acts)
    rng = np.random.default_rng(2)
    B = 60
    X = rng.standard_normal((3, B)); f = {"ta": rng.uniform(0, 30, B)}
    y = {"reco": rng.uniform(1, 9, B)}; y["reco"][::7] = np.nan
    th = ho.init_theta(spec, 4, np.float64)
    l, g, _ = ho.loss_and_grad(spec, th, X, f, y)
    l2, g2 = tt.loss_and_grad(spec, th, X, f, y)
    assert l == pytest.approx(l2, rel=1e-12) and np.max(np.abs(g - g2)) <= 1e-11 * np.max(np.abs(g2))
    for k in rng.choice(th.size, 20, replace=False):
        e = np.zeros_like(th); e[k] = 1e-6
        fd = (ho.compute_loss(spec, th + e, X, f, y) - ho.compute_loss(spec, th - e, X, f, y)) / 2e-6
        assert fd == pytest.approx(g[k], rel=2e-5, abs=2e-7)        # (abs: rounding of a loss of O(10) over a 2e-6 step)
    nets_w, _ = ho.unpack(spec, th)
    par = ho.forward(spec, th, X, f)["parameters"]
    for k, (name, (rows, _)) in enumerate(zip(spec.neural, nets)):
        h = X[rows]
        for li, (W, b) in enumerate(nets_w[k]):
            z = W @ h + b[:, None]
            h = z if li == 2 else ho.act_fwd(acts[k], z)
        lo, hi = spec.lo(name), spec.hi(name)
        assert np.allclose(par[name], lo + (hi - lo) / (1 + np.exp(-h[0])), rtol=1e-12)


@pytest.mark.parametrize("acts", [["tanh", "relu", "sigmoid"], ["swish", "tanh", "identity"], ["sigmoid", "swish", "relu"]])
def test_per_layer_activations_vjp(acts):
    # SingleNN from `hidden_layers::Chain` whose Dense layers carry activations of their own (NNModels.jl:145-219: the reference wraps
    # them as Dense(in, first_h, activation) -> layers... -> Dense(last_h, out)): hand VJP against autograd and central differences,
    # and the forward against the chain evaluated layer by layer
    spec = ho.HybridSpec(3, [12, 9, 7], "rbq10", dict(ho.RBQ10_PARAMS), ["rb"], ["Q10"], ["reco"], acts[0], True, layer_activations=acts)
    rng = np.random.default_rng(5)
    B = 70
    X = rng.standard_normal((3, B)); f = {"ta": rng.uniform(0, 30, B)}
    y = {"reco": rng.uniform(1, 9, B)}; y["reco"][::6] = np.nan
    th = ho.init_theta(spec, 8, np.float64)
    l, g, _ = ho.loss_and_grad(spec, th, X, f, y)
    l2, g2 = tt.loss_and_grad(spec, th, X, f, y)
    assert l == pytest.approx(l2, rel=1e-12) and np.max(np.abs(g - g2)) <= 1e-11 * np.max(np.abs(g2))
    for k in rng.choice(th.size, 25, replace=False):
        e = np.zeros_like(th); e[k] = 1e-6
        fd = (ho.compute_loss(spec, th + e, X, f, y) - ho.compute_loss(spec, th - e, X, f, y)) / 2e-6
        assert fd == pytest.approx(g[k], rel=2e-5, abs=2e-7)
    nets_w, _ = ho.unpack(spec, th)
    h = X
    for li, (W, b) in enumerate(nets_w[0]):
        z = W @ h + b[:, None]
        h = z if li == 3 else ho.act_fwd(acts[li], z)
    lo, hi = spec.lo("rb"), spec.hi("rb")
    assert np.allclose(ho.forward(spec, th, X, f)["parameters"]["rb"], lo + (hi - lo) / (1 + np.exp(-h[0])), rtol=1e-12)
    # the same activation everywhere IS the plain model
    plain = ho.HybridSpec(3, [12, 9, 7], "rbq10", dict(ho.RBQ10_PARAMS), ["rb"], ["Q10"], ["reco"], "tanh", True)
    same = ho.HybridSpec(3, [12, 9, 7], "rbq10", dict(ho.RBQ10_PARAMS), ["rb"], ["Q10"], ["reco"], "relu", True, layer_activations=["tanh"] * 3)
    assert ho.compute_loss(plain, th, X, f, y) == ho.compute_loss(same, th, X, f, y)


@pytest.mark.parametrize("mech", ["expo", "linear", "expo2pool", "rs_components"])
def test_other_mech_models_vjp(mech):
    rng = np.random.default_rng(3)
    mm = ho.MECH[mech][0]
    tabs = {"expo": dict(ho.EXPO_PARAMS), "linear": {"alpha": (1.0, -2.0, 3.0), "beta": (0.5, -1.0, 2.0)},
            "expo2pool": dict(ho.EXPO2POOL_PARAMS),
            "rs_components": {**{f"Rb_{c}": (1.0, 0.0, 5.0) for c in ("het", "root", "myc")},
                              **{f"Q10_{c}": (2.0, 1.0, 4.0) for c in ("het", "root", "myc")}}}
    names = list(mm.params)
    neural, glob = names[: max(1, len(names) // 2)], names[max(1, len(names) // 2):-1] if len(names) > 2 else names[1:]
    spec = ho.HybridSpec(3, [8, 8], mech, tabs[mech], neural, glob, [mm.outputs[0]], "tanh", True)
    B = 50
    X = rng.standard_normal((3, B))
    frc = {mm.forcings[0]: rng.uniform(-5, 25, B)}
    y = {mm.outputs[0]: rng.uniform(0.5, 4, B)}
    th = ho.init_theta(spec, 1, np.float64)
    l, g, _ = ho.loss_and_grad(spec, th, X, frc, y)
    l2, g2 = tt.loss_and_grad(spec, th, X, frc, y)
    assert l == pytest.approx(l2, rel=1e-12)
    assert np.max(np.abs(g - g2)) <= 1e-10 * max(np.max(np.abs(g2)), 1e-12)


@pytest.mark.parametrize("act,scale", [("tanh", True), ("swish", False)])
def test_c_port_matches_numpy_oracle(act, scale):
    spec, th, X, f, y = _case(act, scale, B=500)
    l, g, nv = co.loss_and_grad(spec, th.astype(np.float32), X, f, y, nthreads=3)
    l0, g0, nv0 = ho.loss_and_grad(spec, th, X, f, y)
    assert nv == nv0 and l == pytest.approx(l0, rel=2e-6)
    assert np.max(np.abs(g - g0)) <= 2e-6 * np.max(np.abs(g0))


@pytest.mark.parametrize("act,scale,B", [("tanh", True, 500), ("sigmoid", False, 333), ("relu", True, 37)])
def test_blocked_c_port_matches_the_checker_and_the_numpy_oracle(act, scale, B):
    """oracle/eh_oracle_fast.c -- what bench.py times as `cpu_baseline`: sixteen samples per SIMD block, the rational tanh, vector exp /
    log -- against the scalar checker and the fp64 oracle (ragged last block, missing targets, several threads)"""
    spec, th, X, f, y = _case(act, scale, B=B, nan=0.15)
    l, g, nv = co.loss_and_grad(spec, th.astype(np.float32), X, f, y, nthreads=3, fast=True)
    lc, gc, nvc = co.loss_and_grad(spec, th.astype(np.float32), X, f, y, nthreads=2)
    l0, g0, nv0 = ho.loss_and_grad(spec, th, X, f, y)
    assert nv == nvc == nv0 and l == pytest.approx(l0, rel=1e-5) and l == pytest.approx(lc, rel=1e-5)
    assert np.max(np.abs(g - g0)) <= 1e-5 * np.max(np.abs(g0)) and np.max(np.abs(g - gc)) <= 1e-5 * np.max(np.abs(gc))
    th32 = th.astype(np.float32)
    a, _ = co.train_steps(spec, th32, X, f, y, min(64, B // 2), 4)
    b, _ = co.train_steps(spec, th32, X, f, y, min(64, B // 2), 4, fast=True)
    assert np.max(np.abs(a - b)) <= 5e-5
    yn = {k: np.full_like(v, np.nan) for k, v in y.items()}
    ln, gn, nvn = co.loss_and_grad(spec, th32, X, f, yn, nthreads=2, fast=True)
    assert np.isnan(ln) and not gn.any() and nvn == [0]


def test_c_port_adam_trajectory():
    spec, th, X, f, y = _case("tanh", True, B=512, nan=0.1)
    th32 = th.astype(np.float32)
    a, _ = ho.train_steps(spec, th32, X, f, y, [(i * 128, 128) for i in range(4)], dtype=np.float32)
    b, _ = co.train_steps(spec, th32, X, f, y, 128, 4)
    assert np.max(np.abs(a - b)) <= 5e-5      # Adam's sign-like first steps amplify rounding; both are fp32


def test_all_masked_batch_is_skipped():
    spec, th, X, f, y = _case("tanh", False, B=64)
    y = {"reco": np.full(64, np.nan, np.float32)}
    th32 = th.astype(np.float32)
    out, losses = ho.train_steps(spec, th32, X, f, y, [(0, 64)], dtype=np.float32)
    assert np.array_equal(out, th32) and np.isnan(losses[0])          # epoch.jl:17-19


def _load_spec(d):
    s = json.loads(str(d["spec"]))
    if s["mech"] not in ho.MECH:                           # a fixture of a user closure: tests/closures.py holds the function
        from tests import closures as cl
        from tests import util
        fn, table, forc = cl.CLOSURES[s["mech"]]
        util.register_closure(s["mech"], fn, list(table), forc, s["targets"])
    return ho.HybridSpec(s["n_pred"], s["hidden"], s["mech"], {k: tuple(v) for k, v in s["parameters"].items()}, s["neural"],
                         s["glob"], s["targets"], s["activation"], s["scale_nn_outputs"],
                         nets=[(r, h) for r, h in s["nets"]] if s.get("nets") else None, net_activations=s.get("net_activations"))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_golden_fixtures_reproduce(path):
    d = np.load(path)
    spec = _load_spec(d)
    f = {k[8:]: d[k] for k in d.files if k.startswith("forcing_")}
    y = {k[7:]: d[k] for k in d.files if k.startswith("target_")}
    l, g, nv = ho.loss_and_grad(spec, d["theta"].astype(np.float64), d["X"], f, y)
    assert l == pytest.approx(float(d["loss"]), rel=1e-12)
    assert np.allclose(g, d["grad"], rtol=1e-10, atol=1e-14)
    assert list(nv) == list(d["n_valid"])


# ---- bf16-forward / fp32-accumulate mode (BASELINE.json configs[4]; build-defined, not a reference mode) -----------------
def test_round_bf16_is_round_to_nearest_even():
    import torch
    r = np.random.default_rng(3).standard_normal(200000) * np.exp(np.random.default_rng(4).uniform(-30, 30, 200000))
    want = torch.tensor(r, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()
    assert np.array_equal(ho.round_bf16(r), want)
    assert ho.round_bf16(np.array([1.00390625]))[0] == 1.0 and ho.round_bf16(np.array([1.01171875]))[0] == 1.015625     # ties go to the even mantissa


@pytest.mark.parametrize("act", ["tanh", "sigmoid", "relu"])
def test_bf16_forward_vjp_matches_straight_through_autograd(act):
    """the hand VJP of the bf16-forward mode against autograd of the same forward with round() as the identity and act' taken
    from the stored rounded activation (oracle/torch_twin.py `_RoundedAct`)"""
    spec = ho.c5_spec(hidden=(24, 20), activation=act, n_pred=7)
    X, f, y = ho.make_synth_c5(300, 5, 0.1, n_pred=7)
    theta = ho.init_theta(spec, 2, np.float64)
    l0, g0, nv = ho.loss_and_grad(spec, theta, X, f, y)
    l1, g1 = tt.loss_and_grad(spec, theta, X, f, y)
    assert abs(l0 - l1) <= 1e-12 * abs(l1)
    assert np.max(np.abs(g0 - g1)) <= 1e-11 * np.max(np.abs(g1))
    # and it IS a different function from the fp32 one: the rounding shows at the 1e-3 level
    spec32 = ho.c5_spec(hidden=(24, 20), activation=act, n_pred=7, precision="f32")
    l32, g32, _ = ho.loss_and_grad(spec32, theta, X, f, y)
    assert 1e-6 < abs(l32 - l0) / abs(l32) < 5e-2


@pytest.mark.parametrize("act", ["tanh", "sigmoid", "relu"])
def test_bf16_both_passes_vjp_matches_autograd_with_rounded_deltas(act):
    """precision = "bf16" (bf16 operands in both passes): the hand VJP against autograd of the same forward in which the gradient
    entering each Dense product is rounded to bfloat16 (oracle/torch_twin.py `_RoundedGrad`), the bias gradients un-rounded.  Both
    sides round the SAME fp64 deltas here, so they agree to rounding of the sums; and the mode differs measurably from bf16_fwd."""
    spec = ho.c5_spec(hidden=(24, 20), activation=act, n_pred=7, precision="bf16")
    X, f, y = ho.make_synth_c5(300, 5, 0.1, n_pred=7)
    theta = ho.init_theta(spec, 2, np.float64)
    l0, g0, nv = ho.loss_and_grad(spec, theta, X, f, y)
    l1, g1 = tt.loss_and_grad(spec, theta, X, f, y)
    assert abs(l0 - l1) <= 1e-12 * abs(l1)
    # a delta that sits within 1e-16 of a bf16 rounding boundary may round differently on the two sides (different summation order
    # upstream): rare, and worth one bf16 unit of one delta -- the bar allows for a handful
    assert np.max(np.abs(g0 - g1)) <= 1e-6 * np.max(np.abs(g1))
    specf = ho.c5_spec(hidden=(24, 20), activation=act, n_pred=7, precision="bf16_fwd")
    lf, gf, _ = ho.loss_and_grad(specf, theta, X, f, y)
    assert lf == l0                                      # same forward
    rel = np.max(np.abs(gf - g0)) / np.max(np.abs(gf))
    assert 1e-6 < rel < 2e-2, rel                        # the rounded deltas show at the 1e-3 level


def test_bf16_forward_refuses_swish():
    with pytest.raises(NotImplementedError):
        ho.c5_spec(activation="swish")


def test_three_forcing_components_model_vjp():
    spec = ho.c5_spec(hidden=(12,), activation="tanh", n_pred=5, precision="f32")
    X, f, y = ho.make_synth_c5(200, 9, 0.05, n_pred=5)
    theta = ho.init_theta(spec, 4, np.float64)
    l0, g0, _ = ho.loss_and_grad(spec, theta, X, f, y)
    l1, g1 = tt.loss_and_grad(spec, theta, X, f, y)
    assert abs(l0 - l1) <= 1e-12 * abs(l1) and np.max(np.abs(g0 - g1)) <= 1e-11 * np.max(np.abs(g1))


def test_per_target_losses_vjp_matches_finite_differences():
    """PerTarget((l_1, l_2)) (compute_loss.jl:128-145): each target its own loss, summed; hand VJP against central differences"""
    rng = np.random.default_rng(4)
    B = 80
    pars = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}
    spec = ho.HybridSpec(3, [8], "fluxpart", pars, ["RUE", "Rb"], ["Q10"], ["NEE", "GPP"], "tanh", True)
    X = rng.standard_normal((3, B)); f = {"SW_IN": rng.random(B) * 400, "TA": rng.random(B) * 30}
    y = {"NEE": rng.standard_normal(B), "GPP": rng.random(B) * 3}
    y["NEE"][rng.random(B) < 0.2] = np.nan
    theta = ho.init_theta(spec, 5, np.float64)
    for kinds in (("mse", "mae"), ("nseLoss", "mse"), ("mae", "nseLoss")):
        l0, g0, _ = ho.loss_and_grad(spec, theta, X, f, y, kind=kinds)
        ref = sum(ho.loss_fn(ho.forward(spec, theta, X, f)[t], y[t], ho.valid_mask(y[t]), k) for t, k in zip(spec.targets, kinds))
        assert l0 == pytest.approx(ref, rel=1e-13)
        for i in rng.choice(theta.size, 12, replace=False):
            e = np.zeros_like(theta); e[i] = 1e-6
            lp = ho.loss_and_grad(spec, theta + e, X, f, y, kind=kinds)[0]; lm = ho.loss_and_grad(spec, theta - e, X, f, y, kind=kinds)[0]
            assert (lp - lm) / 2e-6 == pytest.approx(g0[i], rel=2e-5, abs=1e-8)
    with pytest.raises(AssertionError):
        ho.loss_and_grad(spec, theta, X, f, y, kind=("mse",))


def test_model_without_a_network_gradient_matches_finite_differences():
    """no neural parameter: `NN = Chain()` in the reference (src/models/GenericHybridModel.jl:112-125), theta = the raw globals"""
    spec = ho.HybridSpec(0, [], "rbq10", dict(ho.RBQ10_PARAMS), [], ["rb", "Q10"], ["reco"], "tanh", False)
    assert spec.n_theta == 2 and spec.net_list == []
    X0, f, y = ho.make_synth_rbq10(400, 9, 0.15)
    X = X0[:0]
    th = ho.init_theta(spec, 1, np.float64) + 0.3
    l, g, nv = ho.loss_and_grad(spec, th, X, f, y)
    eps = 1e-6
    fd = [(ho.loss_and_grad(spec, th + eps * e, X, f, y)[0] - ho.loss_and_grad(spec, th - eps * e, X, f, y)[0]) / (2 * eps) for e in np.eye(2)]
    assert np.allclose(g, fd, rtol=1e-6)
    # at the defaults (raw = inv_sigmoid of the mid-range default) the model is rb = 3, Q10 = 2 for every sample
    out = ho.forward(spec, ho.init_theta(spec, 1, np.float64), X, f)
    assert np.allclose(out["reco"], 3.0 * 2.0 ** (0.1 * (f["ta"].astype(np.float64) - 15.0)))


def test_weight_l2_terms_gradient_matches_finite_differences():
    """several extra-loss terms (src/utils/extract_weights.jl:64-91: per network, weights or biases, normalised or not): the oracle's
    gradient against central differences of its own value, and the single whole-tree term against weight_l2"""
    spec = ho.HybridSpec(4, [1], "rbq10", dict(ho.RBQ10_PARAMS), ["rb", "Q10"], [], ["reco"], "tanh", True, nets=[([0, 1], [5, 3]), ([2, 3], [4])])
    theta = ho.init_theta(spec, 3, np.float64)
    terms = [(0.3, True, 0, "weight"), (0.7, False, 1, "weight"), (0.2, False, None, "bias"), (0.05, True, None, "weight")]
    vals, g = ho.weight_l2_terms(spec, theta, terms)
    fd = np.zeros_like(theta)
    for i in range(theta.size):
        e = np.zeros_like(theta); e[i] = 1e-6
        fd[i] = (sum(ho.weight_l2_terms(spec, theta + e, terms)[0]) - sum(ho.weight_l2_terms(spec, theta - e, terms)[0])) / 2e-6
    assert np.allclose(g, fd, rtol=1e-7, atol=1e-9)
    v1, g1 = ho.weight_l2(spec, theta, 0.4, True)
    vt, gt = ho.weight_l2_terms(spec, theta, [(0.4, True, None, "weight")])
    assert vt[0] == pytest.approx(v1, rel=1e-14) and np.allclose(gt, g1, rtol=1e-14)
    # the mean of the squared weights of ONE network: test/test_extract_weights.jl's relation, per network
    m = np.zeros(spec.n_theta, bool); m[:5 * 2] = True; m[5 * 2 + 5:5 * 2 + 5 + 3 * 5] = True; m[5 * 2 + 5 + 3 * 5 + 3:5 * 2 + 5 + 3 * 5 + 3 + 3] = True
    assert vals[0] == pytest.approx(0.3 * np.mean(theta[m] ** 2), rel=1e-12)
