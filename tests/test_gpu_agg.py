"""-m gpu: `agg` of the training configuration (`agg::Function`, /root/reference/src/config/TrainingConfig.jl:76-77; the reference's own
test builds `LoggingLoss(agg = mean, ...)`, test/test_loss_types.jl:36).  The training loss is agg(per-target losses)
(src/losses/compute_loss.jl:50-53) and, with an extra loss, agg([that, extra entries...]) (:31-34): for agg = mean
(1 / (1 + E)) [ (1 / T) sum_t L_t + sum_i e_i ].  Through the C ABI (eh_set_option "agg" / "extra_terms") against the oracle, on every
kernel family and step mode; bar 1e-5 (1e-4 for the moment-based losses, as everywhere)."""
import numpy as np
import pytest

import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.engine import HybridEngine
from oracle import hybrid_oracle as ho
from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _flux(B, hidden, T=2, seed=31, P=4):
    rng = np.random.default_rng(seed)
    pars = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}
    targets = ["NEE", "GPP", "RECO"][:T]
    spec = ho.HybridSpec(P, list(hidden), "fluxpart", pars, ["RUE", "Rb"], ["Q10"], targets, "tanh", True)
    X = rng.standard_normal((P, B)).astype(np.float32)
    f = {"SW_IN": (rng.random(B) * 400).astype(np.float32), "TA": (rng.random(B) * 30).astype(np.float32)}
    y = {"NEE": (5 + rng.standard_normal(B)).astype(np.float32), "GPP": (rng.random(B) * 3).astype(np.float32), "RECO": (2 + rng.random(B) * 3).astype(np.float32)}
    y = {t: y[t] for t in targets}
    y["NEE"][rng.random(B) < 0.3] = np.nan
    return spec, ho.init_theta(spec, 6, np.float32), X, f, y


@pytest.mark.parametrize("hidden", [(16, 8), (100, 40), (160, 48, 24)])          # per-wave kernel, row-split kernel, layer-wise form
@pytest.mark.parametrize("T", [2, 3])
def test_mean_over_targets_on_every_kernel_family(hidden, T):
    spec, theta, X, f, y = _flux(1500, hidden, T)
    eng = util.load_engine(spec, theta, X, f, y)
    ls, gs, _ = eng.loss_and_grad()
    eng.set_agg("mean")
    lm, gm, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, agg="mean")
    assert nv == sum(nv0) and lm == pytest.approx(l0, rel=TOL) and util.relerr(gm, g0) <= TOL
    assert lm == pytest.approx(ls / T, rel=2e-6) and util.relerr(gm * T, gs) <= 2e-6          # (and it is the sum over T)
    eng.set_agg("sum")
    l2, g2, _ = eng.loss_and_grad()
    assert l2 == ls and np.array_equal(g2, gs)
    eng.close()


@pytest.mark.parametrize("kinds", [("mse", "mae"), ("nseLoss", "rmse"), ("pearsonLoss", "mse"), ("kgeLoss", "mae")])
def test_mean_with_per_target_and_two_pass_losses(kinds):
    """per-target kinds incl. the two-pass ones (their coefficients k0 k1 k2 and loss values carry the factor)"""
    spec, theta, X, f, y = _flux(1200, (24, 12), 2, seed=7)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_training_loss(kinds)
    eng.set_agg("mean")
    lm, gm, _ = eng.loss_and_grad()
    l0, g0, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=list(kinds), agg="mean")
    tol = 1e-4 if any(k in ("pearsonLoss", "kgeLoss", "pbkgeLoss") for k in kinds) else TOL
    assert lm == pytest.approx(l0, rel=tol) and util.relerr(gm, g0) <= tol
    eng.close()


@pytest.mark.parametrize("shape", ["single", "multinn", "layerwise"])
def test_mean_with_extra_loss_terms(shape):
    """agg([loss, extra entries...]) with agg = mean: one target, E = 2 / 3 weight_l2 entries -- the data loss and every entry enter
    with the factor 1 / (1 + E) (compute_loss.jl:31-34); Adam trajectory included"""
    rng = np.random.default_rng(5)
    if shape == "single":
        spec, theta, X, f, y = util.rbq10_case(900, "tanh", True, 0.1)
        oterms = [(0.4, True, None, "weight"), (0.05, False, None, "bias")]
        terms = [eh.WeightL2(0.4, normalize=True, name="w"), eh.WeightL2(0.05, key="bias")]
    else:
        hid = [[8, 8], [16, 8]] if shape == "multinn" else [[160, 8], [16, 8, 8, 8]]
        spec = ho.HybridSpec(4, [1], "rbq10", dict(ho.RBQ10_PARAMS), ["rb", "Q10"], [], ["reco"], "tanh", True, nets=[([0, 1], hid[0]), ([2, 3], hid[1])])
        X = rng.standard_normal((4, 900)).astype(np.float32)
        f = {"ta": rng.uniform(0, 30, 900).astype(np.float32)}
        y = {"reco": rng.uniform(1, 9, 900).astype(np.float32)}
        theta = ho.init_theta(spec, 6, np.float32)
        nrb = int(sum(o * i for o, i in spec.net_list[0][1]))
        oterms = [(0.04 * nrb, True, 0, "weight"), (0.11, False, 1, "weight"), (0.2, False, 0, "bias")]
        terms = {"l2_rb": eh.WeightL2(0.04 * nrb, normalize=True, net="rb"), "l2_Q10": eh.WeightL2(0.11, net="Q10"), "b_rb": eh.WeightL2(0.2, net="rb", key="bias")}
    model = util.model_from_spec(spec)
    from easyhybrid_jl_amd.train import _apply_extra_loss, _extra_terms
    eng = util.load_engine(spec, theta, X, f, y)
    _apply_extra_loss(eng, model, _extra_terms(terms), "mean")
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, l2=oterms, agg="mean")
    ls, gsum, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, l2=oterms)
    assert l0 == pytest.approx(ls / (1 + len(oterms)), rel=1e-12)
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    eng.opt_init("Adam", 0.003)
    batches = [(i * 300, 300) for i in range(3)] * 2
    losses = [eng.train_step(*b) for b in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, lr=0.003, dtype=np.float32, l2=oterms, agg="mean")
    assert np.allclose(losses, l_ref, rtol=1e-4)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    eng.close()


@pytest.mark.parametrize("fused", [0, 1])
def test_mean_in_the_training_step_both_step_modes(fused):
    """T = 2 on the per-wave family, two-kernel and one-kernel-per-step mode: six Adam steps against the oracle's trajectory"""
    spec, theta, X, f, y = _flux(2048, (16, 16), 2, seed=3)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.opt_init("Adam", 0.01)
    eng.set_agg("mean")
    eng.set_option("fused_update", fused)
    batches = [(i * 512, 512) for i in range(4)] + [(256, 1024), (0, 2048)]
    for b in batches:
        eng.train_step(*b, want_loss=False)
    eng.synchronize()
    th_ref, _ = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32, agg="mean")
    d = np.abs(eng.get_params() - th_ref)
    assert np.mean(d <= 2e-5) >= 0.98 and np.max(d) <= 0.01 * len(batches), (np.mean(d <= 2e-5), np.max(d))       # (Adam's sign-like first steps: bar on the bulk)
    eng.close()


def test_mean_under_the_local_group_of_two_handles():
    """data parallel: the per-target weights of the GLOBAL batch carry the factor (eh_weights_from_counts_kernel)"""
    from easyhybrid_jl_amd import dp
    spec, theta, X, f, y = _flux(2048, (24, 12), 2, seed=8)
    engs = []
    for r in range(2):
        lo, hi = dp.shard_range(2048, r, 2)
        e = util.load_engine(spec, theta, X[:, lo:hi], {k: v[lo:hi] for k, v in f.items()}, {k: v[lo:hi] for k, v in y.items()})
        e.opt_init("Descent", 0.05)
        e.set_agg("mean")
        e.set_target_shift([float(np.nanmean(y[t])) for t in spec.targets])
        engs.append(e)
    HybridEngine.comm_init_local(engs)
    loss = HybridEngine.dp_train_step_group(engs, [0, 0], 1024, want_loss=True)
    l0, g0, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, agg="mean")
    assert loss == pytest.approx(l0, rel=TOL)
    step = (theta.astype(np.float64) - engs[0].get_params().astype(np.float64)) / 0.05
    assert util.relerr(step, g0) <= 2e-5
    assert np.array_equal(engs[0].get_params(), engs[1].get_params())
    for e in engs:
        e.close()


def test_train_front_door_with_agg_mean():
    """train(...; agg = mean): the aggregate of every loss type is reported under `mean` (NamedTuple{(targets..., Symbol(agg))},
    compute_loss.jl:55-66) and is the mean over the targets; early stopping follows it"""
    rng = np.random.default_rng(3)
    n = 6000
    cols = {f"x{i}": rng.standard_normal(n).astype(np.float32) for i in range(4)}
    cols["SW_IN"] = (rng.random(n) * 400).astype(np.float32); cols["TA"] = (rng.random(n) * 30).astype(np.float32)
    gpp = 0.004 * cols["SW_IN"] * (1 + 0.3 * np.tanh(cols["x0"])); reco = (1.5 + 0.5 * np.tanh(cols["x1"])) * 1.6 ** (0.1 * (cols["TA"] - 15))
    cols["GPP"] = (gpp + 0.05 * rng.standard_normal(n)).astype(np.float32); cols["NEE"] = (reco - gpp + 0.05 * rng.standard_normal(n)).astype(np.float32)
    cols["NEE"][rng.random(n) < 0.2] = np.nan
    model = eh.constructHybridModel([f"x{i}" for i in range(4)], ["SW_IN", "TA"], ["NEE", "GPP"], eh.FluxPartModelQ10,
                                    {"RUE": (0.005, 0.0, 0.02), "Rb": (1.5, 0.0, 6.0), "Q10": (1.6, 1.0, 4.0)}, ["RUE", "Rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    out = eh.train(model, cols, nepochs=4, batchsize=512, opt=eh.Adam(0.01), loss_types=["mse", "r2"], agg=np.mean, random_seed=11,
                   extra_loss=eh.WeightL2(1e-3))
    last = out.val_history[-1]
    assert set(last["mse"]) == {"NEE", "GPP", "mean"} and last["mse"]["mean"] == pytest.approx(0.5 * (last["mse"]["NEE"] + last["mse"]["GPP"]))
    assert set(last["extra_loss"]) == {"weight_l2", "mean"}
    assert out.best_loss == pytest.approx(min(h["mse"]["mean"] for h in out.val_history))
    assert out.val_history[-1]["mse"]["mean"] < 0.5 * out.val_history[0]["mse"]["mean"]
    with pytest.raises(NotImplementedError):
        eh.train(model, cols, nepochs=1, batchsize=512, agg=max)
