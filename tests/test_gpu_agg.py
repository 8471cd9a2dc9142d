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


# ---- extra_loss as a function of the PREDICTIONS (src/losses/compute_loss.jl:31-34) ----------------------------------------------
def _extra_case(fn, kinds_red):
    """registers the entries of `fn` with the oracle (program + closure per entry) -> oracle `extra` list"""
    from easyhybrid_jl_amd.program import trace_extra_loss
    ent = trace_extra_loss(fn, ["NEE", "GPP"])
    extra = []
    for i, (name, out, red, pg) in enumerate(ent):
        kname = f"xl_{id(fn)}_{i}"
        ho.loss_program(kname, pg.as_dict(), None)
        extra.append((out, kname, red))
    assert [e[2] for e in extra] == kinds_red
    return extra


@pytest.mark.parametrize("hidden", [(16, 8), (100, 40), (160, 48, 24)])          # per-wave kernel, row-split kernel, layer-wise form
@pytest.mark.parametrize("agg", ["sum", "mean"])
def test_extra_loss_of_the_predictions_the_references_own_case(hidden, agg):
    """`extra_loss_func(yhat, ps) = [sum(abs, yhat.var1), sum(abs, yhat.var2)]` of the reference's test (test/test_compute_loss.jl:
    257-285): the training loss is agg([main_loss, extra entries...]).  Each entry rides on a target of its own (eh_set_target_roles):
    loss and gradient against the oracle, whose extra entries are over ALL samples (the masked ones included, as `yhat.var1` is)."""
    spec, theta, X, f, y = _flux(1400, hidden, 2, seed=12)
    fn = lambda yhat, ps: [np.sum(np.abs(yhat["NEE"])), np.sum(np.abs(yhat["GPP"]))]
    extra = _extra_case(fn, ["sum", "sum"])
    model = util.model_from_spec(spec)
    eng = model.engine(0, extra_fn=fn)
    eng.set_data(eh.EH_SPLIT_TRAIN, X, [f["SW_IN"], f["TA"]], [y["NEE"], y["GPP"]])
    eng.set_params(theta)
    eng.set_agg(agg, eng.n_pseudo)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, agg=agg, extra=extra)
    lm, gm, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
    res = ho.forward(spec, theta.astype(np.float64), X, f)
    want = lm + np.sum(np.abs(res["NEE"])) + np.sum(np.abs(res["GPP"]))          # expected_loss = sum([main_loss, extra_loss_vals...])
    assert (agg != "sum" or l0 == pytest.approx(want, rel=1e-12)) and (agg != "mean" or l0 == pytest.approx((lm / 2 + want - lm) / 3, rel=1e-12))
    assert loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL, (loss, l0, util.relerr(grad, g0))
    assert util.relerr(g0, gm) > 1e-2                                # the entries are not negligible here
    # the caller sees the data targets only
    m, yh = eng.eval(eh.EH_SPLIT_TRAIN, predictions=True)
    assert len(m) == 2 and set(yh) == {"NEE", "GPP"} and set(eng.forward(eh.EH_SPLIT_TRAIN, params=False)) == {"NEE", "GPP"}
    eng.close()


def test_extra_loss_mean_entry_next_to_weight_l2_and_a_training_trajectory():
    """one `mean` entry of one output + a WeightL2 term, mae / mse per target, agg = mean, six Adam steps"""
    spec, theta, X, f, y = _flux(1800, (24, 12), 2, seed=4)
    fn = lambda yhat: {"smooth_gpp": 0.3 * np.mean(yhat["GPP"] ** 2)}
    extra = _extra_case(fn, ["mean"])
    model = util.model_from_spec(spec)
    from easyhybrid_jl_amd.train import _apply_extra_loss, _extra_terms
    eng = model.engine(0, extra_fn=fn)
    eng.set_data(eh.EH_SPLIT_TRAIN, X, [f["SW_IN"], f["TA"]], [y["NEE"], y["GPP"]])
    eng.set_params(theta)
    eng.set_training_loss(eh.PerTarget(("mae", "mse")))
    _apply_extra_loss(eng, model, _extra_terms(eh.WeightL2(0.02)), "mean", eng.n_pseudo)
    loss, grad, _ = eng.loss_and_grad()
    l0, g0, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=["mae", "mse"], l2=(0.02, False), agg="mean", extra=extra)
    assert loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    eng.opt_init("Adam", 0.003)
    batches = [(i * 300, 300) for i in range(6)]
    losses = [eng.train_step(*b) for b in batches]
    th_ref = theta.copy(); st = ho.adam_init(theta.size, np.float32); l_ref = []
    for a, n in batches:
        sl = slice(a, a + n)
        l_, g_, _ = ho.loss_and_grad(spec, th_ref, X[:, sl], {k: v[sl] for k, v in f.items()}, {k: v[sl] for k, v in y.items()}, dtype=np.float32, kind=["mae", "mse"],
                                     l2=(0.02, False), agg="mean", extra=extra)
        th_ref = ho.adam_step(th_ref, g_.astype(np.float32), st, 0.003); l_ref.append(float(l_))
    assert np.allclose(losses, l_ref, rtol=1e-4) and np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5
    eng.close()


def test_train_front_door_with_an_extra_loss_of_the_predictions():
    """TrainConfig.extra_loss = a function of the predictions (next to a WeightL2 term): recorded when the engine is created; the history
    reports every entry and their aggregate like the reference's eval mode (compute_loss.jl:39-44); the penalty does what it says"""
    rng = np.random.default_rng(3)
    n = 6000
    cols = {f"x{i}": rng.standard_normal(n).astype(np.float32) for i in range(4)}
    cols["SW_IN"] = (rng.random(n) * 400).astype(np.float32); cols["TA"] = (rng.random(n) * 30).astype(np.float32)
    gpp = 0.004 * cols["SW_IN"] * (1 + 0.3 * np.tanh(cols["x0"])); reco = (1.5 + 0.5 * np.tanh(cols["x1"])) * 1.6 ** (0.1 * (cols["TA"] - 15))
    cols["GPP"] = (gpp + 0.05 * rng.standard_normal(n)).astype(np.float32); cols["NEE"] = (reco - gpp + 0.05 * rng.standard_normal(n)).astype(np.float32)
    model = eh.constructHybridModel([f"x{i}" for i in range(4)], ["SW_IN", "TA"], ["NEE", "GPP"], eh.FluxPartModelQ10,
                                    {"RUE": (0.005, 0.0, 0.02), "Rb": (1.5, 0.0, 6.0), "Q10": (1.6, 1.0, 4.0)}, ["RUE", "Rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    kw = dict(nepochs=4, batchsize=512, opt=eh.Adam(0.01), loss_types=["mse", "r2"], random_seed=11, return_model="final")      # (final: the predictions returned are the last epoch's)
    plain = eh.train(model, cols, **kw)
    pen = eh.train(model, cols, extra_loss=[eh.WeightL2(1e-4), lambda yhat, ps: {"gpp_size": 5.0 * np.mean(yhat["GPP"] ** 2)}], **kw)
    last = pen.val_history[-1]["extra_loss"]
    assert set(last) == {"weight_l2", "gpp_size", "sum"} and last["sum"] == pytest.approx(last["weight_l2"] + last["gpp_size"])
    assert last["gpp_size"] == pytest.approx(5.0 * float(np.mean(pen.val_obs_pred["GPP_pred"] ** 2)), rel=1e-5)
    assert np.mean(pen.val_obs_pred["GPP_pred"] ** 2) < 0.8 * np.mean(plain.val_obs_pred["GPP_pred"] ** 2)       # the penalty on the size of GPP shrinks GPP
    assert set(pen.val_history[-1]["mse"]) == {"NEE", "GPP", "sum"}


# ----------------------------------------------------------------------------------------------
# a target with NO valid sample inside a batch that has some (missing 1): pinned
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hidden", [(16, 8), (100, 40), (160, 48, 24)])          # per-wave kernel, row-split kernel, layer-wise form
def test_target_without_a_valid_sample_adds_nothing_to_the_gradient(hidden):
    """The reference takes `mean(abs2, yhat[mask] .- y[mask])` over the empty selection (src/losses/loss_fn.jl:61-63): NaN for the VALUE of
    compute_loss, nothing for its GRADIENT (the pullback of a mean over an empty selection scatters nothing back), and only the batch whose
    masks are ALL empty is skipped (src/training/epoch.jl:17-19,35-37).  The engine: the same gradient -- that of the other target alone --,
    the sum of the other targets as the value by default, the reference's NaN with the `empty_target_nan` option at the objective seam; a
    training step is a step on the other target's gradient (the reference discards the step's loss value, epoch.jl:20)."""
    B = 1200
    spec, theta, X, f, y = _flux(B, hidden, 2)
    y["GPP"] = np.full(B, np.nan, np.float32)
    eng = util.load_engine(spec, theta, X, f, y)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
    assert nv0[1] == 0 and nv == nv0[0]
    assert loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL and np.isfinite(grad).all()
    # ... which is the one-target problem on NEE
    spec1 = ho.HybridSpec(spec.n_pred, list(hidden), "fluxpart", dict(spec.parameters), ["RUE", "Rb"], ["Q10"], ["NEE"], "tanh", True)
    l1, g1, _ = ho.loss_and_grad(spec1, theta.astype(np.float64), X, f, {"NEE": y["NEE"]})
    assert l0 == pytest.approx(l1, rel=1e-12) and np.allclose(g0, g1, rtol=1e-12, atol=0)
    # the reference's value on request; the gradient does not change
    ln, gn, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, empty_target="nan")
    assert np.isnan(ln) and np.array_equal(gn, g0)
    eng.set_option("empty_target_nan", 1)
    l2, g2, _ = eng.loss_and_grad()
    assert np.isnan(l2) and np.array_equal(g2, grad)
    eng.set_option("empty_target_nan", 0)
    # a training step moves along the other target's gradient (Descent: theta - lr g)
    eng.opt_init("Descent", 0.1)
    eng.train_step(0, B, want_loss=False)
    step = (theta.astype(np.float64) - eng.get_params().astype(np.float64)) / 0.1
    assert util.relerr(step, g0) <= 3e-5
    eng.close()


def test_target_without_a_valid_sample_under_the_data_parallel_seam():
    """the same batch split over two members of the local group: the GLOBAL count of the empty target is zero, its weight is zero on every
    member, and the group's step is the one-engine step"""
    from tests.test_gpu_comm import _shard_engines
    B = 2048
    spec, theta, X, f, y = _flux(B, (16, 8), 2)
    y["GPP"] = np.full(B, np.nan, np.float32)
    engs = _shard_engines(spec, theta, X, f, y, 2, opt=("Descent", 0.1))
    shift = [float(np.nanmean(y["NEE"])), 0.0]
    for e in engs:
        e.set_target_shift(shift)
    HybridEngine.comm_init_local(engs)
    loss = HybridEngine.dp_train_step_group(engs, [0, 0], B // 2, want_loss=True)
    l0, g0, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
    assert loss == pytest.approx(l0, rel=1e-4)
    step = (theta.astype(np.float64) - engs[0].get_params().astype(np.float64)) / 0.1
    assert util.relerr(step, g0) <= 3e-5 and np.array_equal(engs[0].get_params(), engs[1].get_params())
    for e in engs:
        e.close()


# ---- entries of the extra loss over SEVERAL predictions and over predictions x global parameters (compute_loss.jl:31-34) -------------------
def _mixed_case(targets, fn, hidden, seed=3, B=1300):
    """the three-output flux closure as the model; `fn(yhat, ps)` recorded into its program (program.trace_extra_loss_mixed).  The oracle gets
    the extended program as plain data + a closure that computes the model's outputs AND the entries' per-sample expressions by running
    `fn`'s pieces on NumPy arrays (independent forward values), and identity entries on the new outputs."""
    from easyhybrid_jl_amd.program import trace_extra_loss_mixed, _identity_entry_program
    from tests import closures as cl
    util.register_closure("flux_closure", cl.flux_closure, list(cl.FLUX_TABLE), ["sw", "ta", "vpd"], targets)
    spec = ho.HybridSpec(5, list(hidden), "flux_closure", dict(cl.FLUX_TABLE), ["alpha", "rref", "gmax"], ["e0", "k"], list(targets), "tanh", True)
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (5, B)).astype(np.float32)
    f = {"sw": rng.uniform(0, 800, B).astype(np.float32), "ta": rng.uniform(-5, 30, B).astype(np.float32), "vpd": rng.uniform(0, 30, B).astype(np.float32)}
    theta = ho.init_theta(spec, seed + 1, np.float32)
    truth = ho.forward(spec, ho.init_theta(spec, seed + 2, np.float32).astype(np.float64), X, f)
    y = {}
    for t in targets:
        v = truth[t] * (1.0 + 0.05 * rng.normal(size=B)); v[rng.uniform(size=B) < 0.1] = np.nan
        y[t] = v.astype(np.float32)
    bounds = {g: (cl.FLUX_TABLE[g][1], cl.FLUX_TABLE[g][2]) for g in ("e0", "k")}
    prog, entries = trace_extra_loss_mixed(cl.flux_closure, fn, list(cl.FLUX_TABLE), ["sw", "ta", "vpd"], list(targets), ["e0", "k"], bounds)
    name_x = f"flux_closure_x{id(fn)}"
    ho.program_mech(name_x, prog.as_dict(), None)
    spec_x = ho.HybridSpec(5, list(hidden), name_x, dict(cl.FLUX_TABLE), ["alpha", "rref", "gmax"], ["e0", "k"], list(targets), "tanh", True)
    extra = []
    for i, (nm, out, red, pg) in enumerate(entries):
        kname = f"xlm_{id(fn)}_{i}"
        ho.loss_program(kname, _identity_entry_program().as_dict(), None)
        extra.append((out, kname, red))
    return spec, spec_x, extra, theta, X, f, y


def _raw_globals(spec, theta):
    """ps.<global> as the reference's extra_loss sees it: the raw one-element vectors"""
    _, raw = ho.unpack(spec, np.asarray(theta, np.float64))
    class Ps:
        pass
    ps = Ps()
    for j, g in enumerate(spec.glob):
        setattr(ps, g, np.asarray(raw[j], np.float64).reshape(1))
    ps.__class__.__getitem__ = lambda self, k: getattr(self, k)
    return ps


@pytest.mark.parametrize("hidden", [(16, 8), (100, 40), (160, 48, 24)])          # per-wave kernel, row-split kernel, layer-wise form
@pytest.mark.parametrize("agg", ["sum", "mean"])
def test_extra_loss_entries_over_two_predictions_and_global_parameters(hidden, agg):
    """`extra_loss(yhat, ps) = (; balance = 0.02 mean((yhat.gpp - yhat.reco)^2 + yhat.nee yhat.gpp), scaled = mean(yhat.gpp) ps.k[1] + sum(abs2, ps.e0) 1e-4)`:
    an entry over three outputs of the model (two of them no targets) and one over a prediction and the RAW global parameters -- the
    loss against the value computed from the closure's own outputs on NumPy arrays, loss and gradient against the oracle (reverse sweep over
    the recorded program), the oracle's gradient against central differences in fp64."""
    fn = lambda yhat, ps: {"balance": 0.02 * np.mean((yhat["gpp"] - yhat["reco"]) ** 2 + yhat["nee"] * yhat["gpp"]),
                           "scaled": np.mean(yhat["gpp"]) * ps.k[0] + 1e-4 * np.sum(ps.e0 ** 2)}
    spec, spec_x, extra, theta, X, f, y = _mixed_case(["nee"], fn, hidden)
    assert [e[2] for e in extra] == ["mean", "mean"]
    model = util.model_from_spec(spec)
    eng = model.engine(0, extra_fn=fn)
    assert eng.n_pseudo == 2
    eng.set_option("aot_spec", 0)
    eng.set_data(eh.EH_SPLIT_TRAIN, X, [f[k] for k in model.forcing], [y["nee"]])      # (the model's forcings: the order the recorder met them in)
    eng.set_params(theta)
    eng.set_agg(agg, eng.n_pseudo)
    loss, grad, nv = eng.loss_and_grad()
    th64 = theta.astype(np.float64)
    l0, g0, nv0 = ho.loss_and_grad(spec_x, th64, X, f, y, agg=agg, extra=extra)
    # the value, from the closure's own outputs
    lm, _, _ = ho.loss_and_grad(spec, th64, X, f, y)
    res = ho.forward(spec, th64, X, f)
    ent = fn({k: np.asarray(v, np.float64) for k, v in _all_outputs(spec, th64, X, f).items()}, _raw_globals(spec, th64))
    want = lm + sum(float(v) for v in ent.values()) if agg == "sum" else (lm + sum(float(v) for v in ent.values())) / 3
    assert l0 == pytest.approx(want, rel=2e-6), (l0, want)          # (the recorded program holds its constants -- 0.02, the bounds -- in fp32)
    assert loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL, (loss, l0, util.relerr(grad, g0))
    # the oracle's gradient against central differences (fp64): the last NN bias, both global parameters, two weights
    for j in (0, 7, theta.size - 3, theta.size - 2, theta.size - 1):
        h = 1e-6 * max(1.0, abs(th64[j]))
        tp, tm = th64.copy(), th64.copy(); tp[j] += h; tm[j] -= h
        fd = (ho.loss_and_grad(spec_x, tp, X, f, y, agg=agg, extra=extra)[0] - ho.loss_and_grad(spec_x, tm, X, f, y, agg=agg, extra=extra)[0]) / (2 * h)
        assert fd == pytest.approx(g0[j], rel=2e-5, abs=1e-9), (j, fd, g0[j])
    lm_, gm_, _ = ho.loss_and_grad(spec, th64, X, f, y, agg=agg)
    assert util.relerr(g0, gm_) > 1e-3                                # the entries are not negligible here
    m, yh = eng.eval(eh.EH_SPLIT_TRAIN, predictions=True)             # the caller sees the data target only
    assert len(m) == 1 and set(yh) == {"nee"}
    eng.close()


def _all_outputs(spec, th64, X, f):
    """every output of the flux closure at the oracle's parameters (the reference's yhat holds all of them)"""
    from tests import closures as cl
    res = ho.forward(spec, th64, X, f, keep=True)
    par = res["_tape"]["par"] if "par" in res["_tape"] else None
    if par is None:
        par = {p: res["parameters"][p] for p in cl.FLUX_TABLE}
    return cl.flux_closure(**{k: np.asarray(v, np.float64) for k, v in f.items()}, **{p: np.asarray(par[p], np.float64) for p in cl.FLUX_TABLE})


def test_train_front_door_with_an_entry_over_two_targets_and_a_global_parameter():
    """TrainConfig.extra_loss = f(yhat, ps) mixing both targets and ps.k: the model's closure is recorded again with the entry as an output;
    the history carries the entry (host side, from the predictions and the raw parameter), training moves it"""
    from tests import closures as cl
    cols_rng = np.random.default_rng(5)
    n = 3000
    cols = {f"x{i}": cols_rng.uniform(-1, 1, n).astype(np.float32) for i in range(5)}
    cols.update(sw=cols_rng.uniform(0, 800, n).astype(np.float32), ta=cols_rng.uniform(-5, 30, n).astype(np.float32), vpd=cols_rng.uniform(0, 30, n).astype(np.float32))
    out = cl.flux_closure(sw=cols["sw"], ta=cols["ta"], vpd=cols["vpd"], alpha=0.05, gmax=20.0, rref=3.0, e0=150.0, k=0.05)
    cols["nee"] = (out["nee"] + 0.1 * cols_rng.normal(size=n)).astype(np.float32); cols["gpp"] = (out["gpp"] + 0.1 * cols_rng.normal(size=n)).astype(np.float32)
    model = eh.constructHybridModel([f"x{i}" for i in range(5)], ["sw", "ta", "vpd"], ["nee", "gpp"], cl.flux_closure, dict(cl.FLUX_TABLE), ["alpha", "rref", "gmax"], ["e0", "k"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    kw = dict(nepochs=3, batchsize=512, opt=eh.Adam(0.01), loss_types=["mse"], random_seed=3)
    fn = lambda yhat, ps: {"coupling": 0.05 * np.mean(yhat["nee"] * yhat["gpp"]) * ps.k[0] + 0.5 * np.mean((yhat["nee"] + yhat["gpp"]) ** 2)}
    plain = eh.train(model, cols, **kw)
    pen = eh.train(model, cols, extra_loss=fn, **kw)
    last = pen.val_history[-1]["extra_loss"]
    assert set(last) == {"coupling", "sum"} and np.isfinite(last["coupling"]) and last["coupling"] == pytest.approx(last["sum"])
    assert not np.allclose(pen.ps, plain.ps)
    first = pen.val_history[0]["extra_loss"]["coupling"]
    assert last["coupling"] < first                                   # the penalised quantity went down
    with pytest.raises(NotImplementedError, match="closure"):
        m2 = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, {"rb": (3.0, 0.0, 13.0), "Q10": (2.0, 1.0, 4.0)}, ["rb"], ["Q10"], hidden_layers=[8])
        m2.engine(0, extra_fn=lambda yhat, ps: np.mean(yhat["reco"]) * ps.Q10[0])
