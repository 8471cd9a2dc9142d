"""Pins the CPU oracle on every numeric known-answer the reference's own tests hold for the hot
path.  Paths are relative to /root/reference (not read at run time)."""
import numpy as np
import pytest

from oracle import hybrid_oracle as ho


def test_scale_single_param_known_answers():
    # test/test_generic_hybrid_model.jl:109-117: bounds (0,2) -> 1.0, bounds (1,3) -> 2.0 at raw = 0
    assert ho.scale_single_param(np.float32([0.0]), np.float32(0), np.float32(2))[0] == pytest.approx(1.0)
    assert ho.scale_single_param(np.float32([0.0]), np.float32(1), np.float32(3))[0] == pytest.approx(2.0)


def test_scale_single_param_minmax_midrange_is_zero():
    # test/test_generic_hybrid_model.jl:119-126: default at mid-range -> inv_sigmoid(0.5) = 0
    assert ho.scale_single_param_minmax(np.float32(1), np.float32(0), np.float32(2)) == pytest.approx(0.0)
    assert ho.scale_single_param_minmax(np.float32(2), np.float32(1), np.float32(3)) == pytest.approx(0.0)


def test_rbq10_params_initial_global():
    # RbQ10_PARAMS Q10 = (2, 1, 4)  (test/test_split_data_train.jl:42-45): raw0 = log(1/3 / (2/3)) = log 0.5 ; Q10(raw0) = 2
    raw = ho.scale_single_param_minmax(np.float32(2), np.float32(1), np.float32(4))
    assert raw == pytest.approx(np.log(0.5), rel=1e-6)
    assert ho.scale_single_param(np.float32([raw]), np.float32(1), np.float32(4))[0] == pytest.approx(2.0, rel=1e-6)


YH = np.array([1.0, 2.0, 3.0, 4.0])
Y = np.array([1.1, 1.9, 3.2, 3.8])


def test_mse_known_answers():
    # test/test_loss_fn.jl:6-8,20: mean(abs2, yh - y) = 0.025 ; masked [T,T,F,T] (:90-96) = 0.02
    assert ho.loss_fn(YH, Y, np.ones(4, bool), "mse") == pytest.approx(0.025)
    assert ho.loss_fn(YH, Y, np.array([True, True, False, True]), "mse") == pytest.approx(0.02)


@pytest.mark.parametrize("mask", [np.ones(4, bool), np.array([True, True, False, True])])
def test_metric_closed_forms(mask):
    # test/test_loss_fn.jl:17-74 and :90-145: every metric equals its closed form on yh[mask], y[mask]
    a, b = YH[mask], Y[mask]
    r = np.corrcoef(a, b)[0, 1]
    al, be = np.std(a, ddof=1) / np.std(b, ddof=1), a.mean() / b.mean()
    nse = 1 - np.sum((a - b) ** 2) / np.sum((b - b.mean()) ** 2)
    exp = {"rmse": np.sqrt(np.mean((a - b) ** 2)), "mae": np.mean(np.abs(a - b)), "pearson": r, "nse": nse, "r2": nse,
           "pearsonLoss": 1 - r, "nseLoss": 1 - nse, "kgeLoss": np.sqrt((r - 1) ** 2 + (al - 1) ** 2 + (be - 1) ** 2),
           "β": be, "α": al, "pbkgeLoss": np.sqrt((r - 1) ** 2 + (be - 1) ** 2)}
    exp["kge"] = 1 - exp["kgeLoss"]
    exp["pbkge"] = 1 - exp["pbkgeLoss"]
    for k, v in exp.items():
        assert ho.loss_fn(YH, Y, mask, k) == pytest.approx(v, rel=1e-12), k


def test_multi_target_loss_is_sum_of_mean_squares():
    # test/test_compute_loss.jl:69-79: _compute_loss(:mse, sum) = sum_t mean(abs2, yh_t - y_t); masked stays finite (:90-93)
    yh = {"var1": np.array([1.0, 2.0, 3.0]), "var2": np.array([2.0, 3.0, 4.0])}
    y = {"var1": np.array([1.1, 1.9, 3.2]), "var2": np.array([1.8, 3.1, 3.9])}
    tot = sum(ho.loss_fn(yh[k], y[k], np.ones(3, bool), "mse") for k in yh)
    assert tot == pytest.approx(sum(np.mean((yh[k] - y[k]) ** 2) for k in yh))
    m = np.array([True, False, True])
    assert np.isfinite(sum(ho.loss_fn(yh[k], y[k], m, "mse") for k in yh))


def test_analytic_forward_theta_zero():
    # theta = 0, scale_nn_outputs = true, RbQ10_PARAMS, Q10_raw = log 0.5:
    # rb = 6.5, Q10 = 2 -> yhat(ta = 25) = 13.0, yhat(ta = 15) = 6.5 ; dL/dW3 = 0 for tanh
    spec = ho.rbq10_spec((16, 16), "tanh", True)
    theta = np.zeros(spec.n_theta)
    theta[-1] = np.log(0.5)
    X = np.zeros((2, 2))
    out = ho.forward(spec, theta, X, {"ta": np.array([25.0, 15.0])})
    assert out["reco"] == pytest.approx([13.0, 6.5])
    y = {"reco": np.array([10.0, 7.0])}
    loss, grad, nv = ho.loss_and_grad(spec, theta, X, {"ta": np.array([25.0, 15.0])}, y)
    assert loss == pytest.approx(((13 - 10) ** 2 + (6.5 - 7) ** 2) / 2)
    (Ws,), graw = ho.unpack(spec, grad)
    assert np.all(Ws[2][0] == 0)                                       # dW3 = 0 (h2 = tanh(0) = 0)
    dy = 2 * np.array([3.0, -0.5]) / 2
    p = np.array([2.0, 1.0])
    assert Ws[2][1][0] == pytest.approx(np.sum(dy * p) * 13 * 0.25)   # db3 = sum dy p (hi-lo) s(1-s)


def test_flat_theta_layout():
    # ComponentArray order: layer weights column-major (out,in) then bias, then global raws (GenericHybridModel.jl:236-256)
    spec = ho.rbq10_spec((16, 16))
    assert spec.n_theta == 338 and spec.layer_dims == [(16, 2), (16, 16), (1, 16)]
    theta = np.arange(338, dtype=np.float64)
    (Ws,), raw = ho.unpack(spec, theta)
    assert Ws[0][0][3, 1] == 3 + 16 * 1 and Ws[0][1][0] == 32 and Ws[1][0][0, 0] == 48 and raw[0] == 337


def test_adam_first_step_is_lr_sign():
    # Optimisers.Adam: after the first step the update is eta * g/(|g| + eps') ~ eta * sign(g)
    th = np.array([1.0, -2.0, 0.5], np.float32)
    g = np.array([0.3, -4.0, 1e-3], np.float32)
    st = ho.adam_init(3)
    th2 = ho.adam_step(th, g, st, lr=0.01)
    assert np.allclose(th - th2, 0.01 * np.sign(g), rtol=1e-4)
    assert st["t"] == 1


def test_weight_l2_relations():
    # test/test_extract_weights.jl:21-39 on a Dense(2 => 16, tanh), Dense(16 => 1) chain: weight_l2(ps) = sum of the squared
    # WEIGHT matrices; its gradient is nowhere zero on the weights; biases are not regularised; normalize = true divides by the
    # number of weights (the BatchNorm layer of the reference's chain has no :weight leaf with affine = false and adds nothing)
    spec = ho.HybridSpec(2, [16], "rbq10", dict(ho.RBQ10_PARAMS), ["rb"], ["Q10"], ["reco"], "tanh", False)
    theta = ho.init_theta(spec, 11, np.float64)
    (Ws,), raw = ho.unpack(spec, theta)
    manual = sum(np.sum(W ** 2) for W, _ in Ws)
    val, grad = ho.weight_l2(spec, theta, 1.0)
    assert val == pytest.approx(manual, rel=1e-14)
    lam = 1e-3
    val_l, grad_l = ho.weight_l2(spec, theta, lam)
    assert val_l == pytest.approx(lam * manual, rel=1e-14)
    m = ho.weight_mask(spec)
    assert np.all(grad_l[m] != 0) and np.all(grad_l[~m] == 0)            # every weight gets a gradient, no bias / global does
    assert np.allclose(grad_l[m], 2 * lam * theta[m])
    n_weights = sum(W.size for W, _ in Ws)
    assert m.sum() == n_weights == 2 * 16 + 16
    val_n, _ = ho.weight_l2(spec, theta, 1.0, normalize=True)
    assert val_n == pytest.approx(manual / n_weights, rel=1e-14)


def test_extra_loss_is_added_through_agg_sum():
    # test/test_compute_loss.jl:257-285: compute_loss(train mode, extra_loss) = sum([main_loss, extra...]); :287-308: without
    # extra_loss it is the main loss, which is _compute_loss(HM(x), y, mask, targets, :mse, sum)
    spec = ho.rbq10_spec((16, 16), "tanh", True)
    X, f, y = ho.make_synth_rbq10(64, 3, 0.2)
    X = X / 50
    theta = ho.init_theta(spec, 5, np.float64)
    main = ho.compute_loss(spec, theta, X, f, y)
    res = ho.forward(spec, theta, X, f)
    yv = np.asarray(y["reco"], np.float64)
    msk = ho.valid_mask(yv)
    assert main == pytest.approx(ho.loss_fn(res["reco"], yv, msk, "mse"), rel=1e-14)
    l_plain, g_plain, _ = ho.loss_and_grad(spec, theta, X, f, y)
    assert l_plain == pytest.approx(main, rel=1e-13)
    lam = 0.05
    extra, gextra = ho.weight_l2(spec, theta, lam)
    l_extra, g_extra, _ = ho.loss_and_grad(spec, theta, X, f, y, l2=(lam, False))
    assert l_extra == pytest.approx(sum([main, extra]), rel=1e-13)
    assert np.allclose(g_extra, g_plain + gextra, rtol=1e-13, atol=0)
