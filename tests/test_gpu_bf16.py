"""BASELINE.json configs[4] as stated: MLP [32,128,128,6] + the three-forcing RbQ10-family model, bf16 forward / fp32 accumulate
(csrc/eh_wide_bf16.hpp), against the bf16-emulating oracle (oracle/hybrid_oracle.py, HybridSpec.precision = "bf16_fwd"), through
the C ABI.

Tolerances.  The device rounds the fp32 value of an activation, the oracle the fp64 one; the two differ by ~1e-7 relative, so about
one rounded operand in 2^-8 / 1e-7 ~ 4e4 lands on the other side of a bf16 rounding boundary and then differs by one bf16 unit
(0.4 %).  A sample has 32 + 128 + 128 rounded operands: roughly 1 % of the samples see one flip and their prediction moves by
~1e-4 relative; sums over a batch average that out.  Hence: loss within 2e-5, gradient within 5e-5 of its largest entry (both far
below the 4e-3 quantisation step and the ~1e-2 distance between the bf16 and the fp32 model, which the tests also check), and per
sample 98 % of the predictions within 1e-5, all within 2e-3."""
import numpy as np
import pytest

import easyhybrid_jl_amd as eh
from oracle import hybrid_oracle as ho
from tests import util

pytestmark = pytest.mark.gpu


def _case(B, act="tanh", hidden=(128, 128), n_pred=32, nan=0.1, seed=11):
    spec = ho.c5_spec(hidden, act, "bf16_fwd", n_pred)
    X, f, y = ho.make_synth_c5(B, seed, nan, n_pred)
    return spec, ho.init_theta(spec, 3, np.float32), X, f, y


def _check(eng, spec, theta, X, f, y, ltol=2e-5, gtol=5e-5):
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
    assert nv == sum(nv0)
    assert abs(loss - l0) <= ltol * abs(l0), (loss, l0)
    assert util.relerr(grad, g0) <= gtol, util.relerr(grad, g0)
    return loss, grad, l0, g0


@pytest.mark.parametrize("B", [33, 1000, 4096])
def test_config5_loss_and_gradient_match_the_bf16_oracle(B):
    spec, theta, X, f, y = _case(B)
    eng = util.load_engine(spec, theta, X, f, y)
    loss, grad, l0, g0 = _check(eng, spec, theta, X, f, y)
    # the same engine in fp32 mode computes a measurably different function (the test would not notice a kernel that ignored the option)
    eng.set_option("precision", 0)
    l32, g32, _ = eng.loss_and_grad()
    spec32 = ho.c5_spec(precision="f32")
    l32o, g32o, _ = ho.loss_and_grad(spec32, theta.astype(np.float64), X, f, y)
    assert abs(l32 - l32o) <= 1e-5 * abs(l32o) and util.relerr(g32, g32o) <= 1e-5
    assert abs(l32 - loss) > 1e-5 * abs(loss)
    eng.set_option("precision", 1)
    l2, g2, _ = eng.loss_and_grad()
    assert l2 == loss and np.array_equal(g2, grad)           # deterministic, and the switch goes both ways
    eng.close()


@pytest.mark.parametrize("act", ["sigmoid", "relu", "identity"])
def test_other_activations(act):
    spec, theta, X, f, y = _case(777, act)
    eng = util.load_engine(spec, theta, X, f, y)
    _check(eng, spec, theta, X, f, y)
    eng.close()


@pytest.mark.parametrize("hidden,n_pred", [((128,), 32), ((100, 70), 20), ((64, 64), 8), ((40, 33, 50), 5), ((128, 128), 3)])
def test_other_shapes(hidden, n_pred):
    """one hidden layer, ragged widths (zero padded), the 64-wide family (four waves), three layers, fewer predictors than one MFMA k-step"""
    spec, theta, X, f, y = _case(600, "tanh", hidden, n_pred)
    eng = util.load_engine(spec, theta, X, f, y)
    _check(eng, spec, theta, X, f, y)
    eng.close()


def test_forward_and_metrics():
    spec, theta, X, f, y = _case(3000)
    eng = util.load_engine(spec, theta, X, f, y)
    m, yhat = eng.eval(eh.EH_SPLIT_TRAIN, predictions=True)
    par = eng.forward(eh.EH_SPLIT_TRAIN)["parameters"]
    res = ho.forward(spec, theta.astype(np.float64), X, f)
    err = np.abs(yhat["R_soil"] - res["R_soil"]) / np.abs(res["R_soil"])
    assert np.mean(err <= 1e-5) >= 0.98 and err.max() <= 2e-3, (np.mean(err <= 1e-5), err.max())
    ev, _ = ho.evaluate(spec, theta.astype(np.float64), X, f, y, ("mse", "r2"))
    assert abs(m[0]["mse"] - ev["mse"]["R_soil"]) <= 2e-5 * ev["mse"]["R_soil"]
    assert abs(m[0]["r2"] - ev["r2"]["R_soil"]) <= 2e-5
    for n in spec.parameters:
        e = np.abs(par[n] - res["parameters"][n]) / np.abs(res["parameters"][n])
        assert np.mean(e <= 1e-5) >= 0.98 and e.max() <= 2e-3
    eng.close()


def test_adam_trajectory():
    spec, theta, X, f, y = _case(2048, nan=0.05)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.opt_init("Adam", 0.01)
    windows = [(0, 1024), (1024, 1024), (512, 1024), (0, 2048)]
    losses = [eng.train_step(a, n) for a, n in windows]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, windows, dtype=np.float32)
    # Adam's first steps are sign-like (|update| = lr whatever the gradient's size): an entry whose gradient is within rounding of
    # zero can take the other sign, so the bar is on the bulk, as in test_gpu_parity.py
    d = np.abs(eng.get_params() - th_ref)
    assert np.mean(d <= 2e-5) >= 0.999 and np.max(d) <= 2.5e-2 * 0.01 * len(windows) * 40, (np.mean(d <= 2e-5), np.max(d))
    assert np.allclose(losses, l_ref, rtol=5e-5)
    eng.close()


def test_all_missing_targets_skip_the_step():
    spec, theta, X, f, y = _case(128)
    y = {"R_soil": np.full(128, np.nan, np.float32)}
    eng = util.load_engine(spec, theta, X, f, y)
    loss, grad, nv = eng.loss_and_grad()
    assert nv == 0 and np.isnan(loss) and not np.any(grad)
    eng.opt_init("Adam", 0.01)
    eng.train_step(0, 128, want_loss=False)
    assert np.array_equal(eng.get_params(), theta)
    eng.close()


def test_gathered_minibatch_matches_the_contiguous_one():
    spec, theta, X, f, y = _case(1500)
    eng = util.load_engine(spec, theta, X, f, y)
    idx = np.random.default_rng(1).permutation(1500)[:700].astype(np.int32)
    loss, grad, nv = eng.loss_and_grad(idx=idx)
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X[:, idx], {k: v[idx] for k, v in f.items()}, {k: v[idx] for k, v in y.items()})
    assert nv == sum(nv0) and abs(loss - l0) <= 2e-5 * abs(l0) and util.relerr(grad, g0) <= 5e-5
    eng.close()


def test_full_size_batch_is_the_count_weighted_sum_of_its_quarters():
    """BASELINE batch 65 536 (the oracle would take minutes): sums are additive over samples"""
    B = 65536
    spec, theta, X, f, y = _case(B, nan=0.03)
    eng = util.load_engine(spec, theta, X, f, y)
    loss, grad, nv = eng.loss_and_grad()
    acc_l, acc_g, acc_n = 0.0, np.zeros_like(grad, np.float64), 0
    for q in range(4):
        l, g, n = eng.loss_and_grad(first=q * B // 4, count=B // 4)
        acc_l += l * n; acc_g += g.astype(np.float64) * n; acc_n += n
    assert acc_n == nv
    assert abs(acc_l / nv - loss) <= 2e-6 * abs(loss) and util.relerr(acc_g / nv, grad) <= 2e-6
    eng.close()


def test_refusals():
    spec = ho.c5_spec(activation="tanh", precision="f32")
    model = util.model_from_spec(ho.HybridSpec(32, [128, 128], "rs_components3f", dict(ho.RS6_PARAMS), list(ho.RS6_PARAMS), [], ["R_soil"], "swish", True))
    eng = model.engine()
    with pytest.raises(NotImplementedError):
        eng.set_option("precision", 1)                      # swish needs the pre-activation
    eng.close()
    narrow = util.model_from_spec(ho.rbq10_spec((16, 16), "tanh", True)).engine()
    with pytest.raises(NotImplementedError):
        narrow.set_option("precision", 1)                   # no row-split kernel for 16-wide nets
    narrow.close()
