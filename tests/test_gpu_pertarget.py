"""-m gpu: PerTarget training losses (src/losses/compute_loss.jl:128-145), non-MSE losses on multi-target models, and the tuple forms
(f, args) / (f, kwargs) of a custom training loss (src/losses/loss_fn.jl:92-107), on every kernel family, against the fp64 oracle."""
import numpy as np
import pytest

import easyhybrid_jl_amd as eh
from oracle import hybrid_oracle as ho
from tests import util

pytestmark = pytest.mark.gpu
PARS = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}


def _flux_case(hidden, B=900, seed=8):
    rng = np.random.default_rng(seed)
    spec = ho.HybridSpec(6, list(hidden), "fluxpart", dict(PARS), ["RUE", "Rb"], ["Q10"], ["NEE", "GPP"], "tanh", True)
    X = rng.standard_normal((6, B)).astype(np.float32)
    f = {"SW_IN": (rng.random(B) * 400).astype(np.float32), "TA": (rng.random(B) * 30).astype(np.float32)}
    y = {"NEE": rng.standard_normal(B).astype(np.float32), "GPP": (rng.random(B) * 3).astype(np.float32)}
    y["NEE"][rng.random(B) < 0.2] = np.nan; y["GPP"][rng.random(B) < 0.1] = np.nan
    return spec, ho.init_theta(spec, 3, np.float32), X, f, y


@pytest.mark.parametrize("hidden", [(16, 16), (48, 48), (128, 96), (160, 96, 48, 24)])      # per-wave, per-wave 64-wide family, row-split, layer-wise
@pytest.mark.parametrize("kinds", [("mse", "mae"), ("nseLoss", "mse"), ("mae", "mae"), ("nseLoss", "nseLoss")])
def test_per_target_losses(hidden, kinds):
    spec, theta, X, f, y = _flux_case(hidden)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_training_loss(eh.PerTarget(kinds) if kinds[0] != kinds[1] else kinds[0])
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=kinds)
    assert nv == sum(nv0) and abs(loss - l0) <= 1e-5 * abs(l0) and util.relerr(grad, g0) <= 1e-5, (loss, l0, util.relerr(grad, g0))
    eng.opt_init("Adam", 0.01)
    batches = [(0, 300), (300, 300), (600, 300)]
    losses = [eng.train_step(a, n) for a, n in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32, kind=kinds)
    assert np.allclose(losses, l_ref, rtol=5e-5)
    d = np.abs(eng.get_params() - th_ref)
    assert np.mean(d <= 3e-5) >= 0.995
    eng.close()


def test_train_front_door_with_per_target_losses():
    rng = np.random.default_rng(3)
    n = 1200
    cols = {f"x{i}": rng.standard_normal(n).astype(np.float32) for i in range(6)}
    cols["SW_IN"] = (rng.random(n) * 400).astype(np.float32); cols["TA"] = (rng.random(n) * 30).astype(np.float32)
    cols["NEE"] = rng.standard_normal(n).astype(np.float32); cols["GPP"] = (rng.random(n) * 3).astype(np.float32)
    model = eh.constructHybridModel([f"x{i}" for i in range(6)], ["SW_IN", "TA"], ["NEE", "GPP"], eh.FluxPartModelQ10, dict(PARS), ["RUE", "Rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    out = eh.train(model, cols, nepochs=3, batchsize=200, training_loss=eh.PerTarget(("mae", "mse")), random_seed=1)
    assert len(out.val_history) == 4 and np.isfinite(out.best_loss)
    with pytest.raises(AssertionError):
        eh.train(model, cols, nepochs=1, batchsize=200, training_loss=eh.PerTarget(("mae",)), random_seed=1)
    out = eh.train(model, cols, nepochs=2, batchsize=200, training_loss=eh.PerTarget(("kgeLoss", "rmse")), random_seed=1)      # two-pass losses per target
    assert len(out.val_history) == 3 and np.isfinite(out.best_loss)
    with pytest.raises(NotImplementedError):
        eh.train(model, cols, nepochs=1, batchsize=200, training_loss=eh.PerTarget(("kgeLoss", "no_such_loss")), random_seed=1)


def test_tuple_forms_of_a_custom_training_loss():
    """training_loss = (f, args) / (f, kwargs): f(yhat[mask], y[mask], args...; kwargs...)   (loss_fn.jl:92-107)"""
    def huber(yh, y, delta):
        r = yh - y
        return np.mean(np.where(np.abs(r) <= delta, 0.5 * r * r, delta * (np.abs(r) - 0.5 * delta)))
    spec, theta, X, f, y = util.rbq10_case(800, "tanh", True, 0.1)
    name = util.register_loss("huber_0p7", lambda yh, y: huber(yh, y, 0.7)) and "huber_0p7"
    l0, g0, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=name)
    for form in ((huber, (0.7,)), (huber, {"delta": 0.7}), (huber, (), {"delta": 0.7})):
        eng = util.load_engine(spec, theta, X, f, y)
        eng.set_training_loss(form)
        loss, grad, _ = eng.loss_and_grad()
        assert abs(loss - l0) <= 1e-5 * abs(l0) and util.relerr(grad, g0) <= 2e-5
        eng.close()


# ---- losses that need batch statistics of the predictions, per target on multi-target models: rmse (its scale 1 / (n rmse) has to be
# known inside the one pass that serves all targets) and the moment losses pearsonLoss / kgeLoss / pbkgeLoss.  The reference applies
# whatever training_loss it is given to every target (src/losses/compute_loss.jl:115-145, loss_fn.jl:58-174). ----------------------------
@pytest.mark.parametrize("hidden", [(16, 16), (48, 48), (128, 96), (160, 96, 48, 24)])      # per-wave, per-wave 64-wide family, row-split, layer-wise
@pytest.mark.parametrize("kinds", [("rmse", "rmse"), ("rmse", "mae"), ("mse", "rmse"), ("kgeLoss", "mse"), ("pearsonLoss", "nseLoss"), ("pbkgeLoss", "rmse"), ("kgeLoss", "kgeLoss")])
def test_two_pass_losses_per_target(hidden, kinds):
    spec, theta, X, f, y = _flux_case(hidden)
    # (targets the model can correlate with: a correlation of noise is ill-conditioned, tests/test_gpu_fuzz.py)
    ref = ho.forward(spec, (theta + np.float32(0.05) * np.random.default_rng(5).standard_normal(theta.size).astype(np.float32)).astype(np.float64), X, f)
    rng = np.random.default_rng(6)
    for t in spec.targets:
        m = np.isnan(y[t])
        y[t] = (ref[t] * 0.8 + 0.3 + 0.2 * np.std(ref[t]) * rng.standard_normal(X.shape[1])).astype(np.float32)
        y[t][m] = np.nan
    moment = any(k in ("kgeLoss", "pearsonLoss", "pbkgeLoss") for k in kinds)
    tol = 1e-4 if moment else 1e-5                           # (differences of moments amplify fp32 rounding: DESIGN section 3.4)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_training_loss(eh.PerTarget(kinds) if kinds[0] != kinds[1] else kinds[0])
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=kinds)
    assert nv == sum(nv0) and abs(loss - l0) <= tol * abs(l0) and util.relerr(grad, g0) <= tol, (loss, l0, util.relerr(grad, g0))
    idx = np.random.default_rng(2).permutation(X.shape[1])[:500].astype(np.int32)
    loss, grad, nv = eng.loss_and_grad(idx=idx)
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X[:, idx], {k: v[idx] for k, v in f.items()}, {k: v[idx] for k, v in y.items()}, kind=kinds)
    assert nv == sum(nv0) and abs(loss - l0) <= tol * abs(l0) and util.relerr(grad, g0) <= tol
    eng.opt_init("Descent", 0.02)
    batches = [(0, 300), (300, 300), (600, 300)]
    losses = [eng.train_step(a, n) for a, n in batches]
    th = theta.astype(np.float32).copy()
    for (a, n), lg in zip(batches, losses):
        sl = slice(a, a + n)
        l_, g_, _ = ho.loss_and_grad(spec, th.astype(np.float64), X[:, sl], {k: v[sl] for k, v in f.items()}, {k: v[sl] for k, v in y.items()}, kind=kinds)
        assert abs(lg - l_) <= 5 * tol * abs(l_)
        th = (th - np.float32(0.02) * g_.astype(np.float32)).astype(np.float32)
    assert np.max(np.abs(eng.get_params() - th)) <= 20 * tol * max(1.0, float(np.max(np.abs(th))))
    with pytest.raises(NotImplementedError):
        eng.set_option("fused_update", 1)                    # forward passes ahead of the step: no one-kernel mode
    eng.close()


def test_one_target_without_a_valid_sample_adds_nothing():
    spec, theta, X, f, y = _flux_case((16, 16), B=400)
    y["GPP"][:] = np.nan
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_training_loss(eh.PerTarget(("mse", "rmse")))
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=("mse", "rmse"))
    assert nv == sum(nv0) and abs(loss - l0) <= 1e-5 * abs(l0) and util.relerr(grad, g0) <= 1e-5
    eng.close()


@pytest.mark.parametrize("hidden", [(16, 16), (128, 96)])
def test_recorded_loss_function_on_a_multi_target_model(hidden):
    """training_loss::Function is applied to every target (loss_fn.jl:92-94 through compute_loss.jl:115-126); PerTarget mixes it with
    the named losses"""
    def logcosh(yh, y):                                         # (pseudo-Huber, the smooth cousin: sqrt(1 + r^2) - 1)
        r = yh - y
        return np.mean(np.sqrt(1.0 + r * r) - 1.0)
    spec, theta, X, f, y = _flux_case(hidden)
    name = util.register_loss("pseudo_huber_mt", logcosh) and "pseudo_huber_mt"
    for spec_kinds, dev in (((name, name), logcosh), ((name, "mse"), eh.PerTarget((logcosh, "mse"))), (("mae", name), eh.PerTarget(("mae", logcosh)))):
        eng = util.load_engine(spec, theta, X, f, y)
        eng.set_training_loss(dev)
        loss, grad, nv = eng.loss_and_grad()
        l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=spec_kinds)
        assert nv == sum(nv0) and abs(loss - l0) <= 1e-5 * abs(l0) and util.relerr(grad, g0) <= 2e-5, (spec_kinds, loss, l0, util.relerr(grad, g0))
        eng.close()


def test_per_target_custom_losses_as_in_the_reference_test():
    """test/test_compute_loss.jl:44-62 on the device: PerTarget((:mse, custom_loss)) = mse(target 1) + custom(target 2), and
    PerTarget(((weighted_loss, (0.5,)), (scaled_loss, (scale = 2.0,)))) = 0.5 mse(target 1) + 2 mse(target 2) -- two DIFFERENT
    functions, each recorded into its own program (eh_set_target_loss_program)"""
    def custom_loss(yh, y):
        return np.mean((yh - y) ** 2)

    def weighted_loss(yh, y, w):
        return w * np.mean((yh - y) ** 2)

    def scaled_loss(yh, y, scale=1.0):
        return scale * np.mean((yh - y) ** 2)
    spec, theta, X, f, y = _flux_case((16, 16))
    eng = util.load_engine(spec, theta, X, f, y)
    # the per-target mse values from the oracle: l_1, l_2 with loss = l_1 + l_2
    l_sum, g_sum, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=("mse", "mse"))
    ynan = dict(y); ynan["GPP"] = np.full_like(y["GPP"], np.nan)
    l1, g1, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, ynan, kind=("mse", "mse"))          # target 1 alone
    l2, g2 = l_sum - l1, g_sum - g1
    eng.set_training_loss(eh.PerTarget(("mse", custom_loss)))
    loss, grad, _ = eng.loss_and_grad()
    assert abs(loss - (l1 + l2)) <= 1e-5 * abs(l1 + l2) and util.relerr(grad, g1 + g2) <= 2e-5
    eng.set_training_loss(eh.PerTarget(((weighted_loss, (0.5,)), (scaled_loss, {"scale": 2.0}))))
    loss, grad, _ = eng.loss_and_grad()
    want_l, want_g = 0.5 * l1 + 2.0 * l2, 0.5 * g1 + 2.0 * g2
    assert abs(loss - want_l) <= 1e-5 * abs(want_l) and util.relerr(grad, want_g) <= 2e-5, (loss, want_l, util.relerr(grad, want_g))
    with pytest.raises(AssertionError):
        eng.set_training_loss(eh.PerTarget(("mse",)))             # mismatched number of losses and targets
    eng.close()
