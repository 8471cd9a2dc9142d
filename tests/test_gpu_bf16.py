"""BASELINE.json configs[4] as stated: MLP [32,128,128,6] + the three-forcing RbQ10-family model, bf16 forward / fp32 accumulate
(csrc/eh_wide_bf16.hpp), against the bf16-emulating oracle (oracle/hybrid_oracle.py, HybridSpec.precision = "bf16_fwd"), through
the C ABI.

Tolerances.  The device rounds the fp32 value of an activation, the oracle the fp64 one; the two differ by ~1e-7 relative, so about
one rounded operand in 2^-8 / 1e-7 ~ 4e4 lands on the other side of a bf16 rounding boundary and then differs by one bf16 unit
(0.4 %).  A sample has 32 + 128 + 128 rounded operands: roughly 1 % of the samples see one flip and their prediction moves by
~1e-4 relative; sums over a batch average that out.  Hence: loss within 2e-5, gradient within 5e-5 of its largest entry (both far
below the 4e-3 quantisation step and the ~1e-2 distance between the bf16 and the fp32 model, which the tests also check), and per
sample 98 % of the predictions within 1e-5, all within 2e-3.

precision = "bf16" (bf16 operands in BOTH passes, every backward delta rounded to bfloat16 once; oracle `HybridSpec.precision =
"bf16"`): the forward is the same function, so the loss bar stays 2e-5.  The deltas are rounded in the scale the step carries them --
un-normalised for a one-target model (the division by n comes after the pass) -- and the oracle rounds in that same scale; device
(fp32) and oracle (fp64) deltas then differ by ~1e-7 relative before rounding, one in ~4e4 rounds the other way, and the gradient
agrees to 3e-7 .. 6e-7 of its largest entry (measured; bar 2e-5), against 1e-4 .. 1e-3 between the "bf16" and the "bf16_fwd"
gradients (rounding noise that averages out over the batch), which the tests also check."""
import numpy as np
import pytest

import easyhybrid_jl_amd as eh
from oracle import hybrid_oracle as ho
from tests import util

pytestmark = pytest.mark.gpu


GTOL_BF16 = 2e-5        # gradient bar of precision = "bf16" (see the module docstring)


def _case(B, act="tanh", hidden=(128, 128), n_pred=32, nan=0.1, seed=11, precision="bf16_fwd"):
    spec = ho.c5_spec(hidden, act, precision, n_pred)
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


@pytest.mark.parametrize("B", [33, 1000, 4096])
def test_config5_bf16_operands_in_both_passes_match_the_oracle(B):
    """precision = "bf16": the same forward as "bf16_fwd", the backward products (bias sums included) on once-rounded deltas"""
    spec, theta, X, f, y = _case(B, precision="bf16")
    eng = util.load_engine(spec, theta, X, f, y)
    loss, grad, l0, g0 = _check(eng, spec, theta, X, f, y, gtol=GTOL_BF16)
    print(f"bf16 both passes, B = {B}: gradient relerr {util.relerr(grad, g0):.2e}")
    eng.set_option("precision", 1)                         # the exact-backward mode on the same engine: same loss, a measurably different gradient
    l1, g1, _ = eng.loss_and_grad()
    specf = ho.c5_spec(precision="bf16_fwd")
    _, gf, _ = ho.loss_and_grad(specf, theta.astype(np.float64), X, f, y)
    # (the same forward FUNCTION; since round 5 the two modes run different kernels -- "bf16" the sample-owned one, csrc/eh_bf16_sample.hpp --
    #  so the loss sums agree to the order of summation, not bit for bit)
    assert abs(l1 - loss) <= 1e-6 * abs(loss) and util.relerr(g1, gf) <= 5e-5
    assert util.relerr(grad, g1) > 5e-5, util.relerr(grad, g1)      # (a test that would not notice a kernel that ignored the option)
    # ... and the oracle sees the same distance between the two modes
    assert abs(util.relerr(g0, gf) - util.relerr(grad, g1)) <= 0.1 * util.relerr(g0, gf) + GTOL_BF16
    eng.set_option("precision", 2)
    l2, g2, _ = eng.loss_and_grad()
    assert l2 == loss and np.array_equal(g2, grad)           # deterministic, and the switch goes both ways
    eng.close()


@pytest.mark.parametrize("act", ["sigmoid", "relu", "identity"])
def test_other_activations_bf16_both_passes(act):
    spec, theta, X, f, y = _case(777, act, precision="bf16")
    eng = util.load_engine(spec, theta, X, f, y)
    _check(eng, spec, theta, X, f, y, gtol=GTOL_BF16)
    eng.close()


@pytest.mark.parametrize("hidden,n_pred", [((128,), 32), ((100, 70), 20), ((64, 64), 8), ((40, 33, 50), 5)])
def test_other_shapes_bf16_both_passes(hidden, n_pred):
    spec, theta, X, f, y = _case(600, "tanh", hidden, n_pred, precision="bf16")
    eng = util.load_engine(spec, theta, X, f, y)
    _check(eng, spec, theta, X, f, y, gtol=GTOL_BF16)
    eng.close()


def test_adam_trajectory_bf16_both_passes():
    """Four Adam steps.  The one-step gradient agrees with the oracle to 5e-7 (above); a trajectory cannot be held to that: the once-
    rounded deltas give every gradient entry a rounding noise of 1e-4 .. 1e-3 relative that ANY perturbation re-draws -- one Adam
    sign flip of a near-zero entry in step 1 (|update| = lr whatever the size) moves a weight by 2 lr, every delta of the next step
    rounds afresh, and the updates differ by ~lr x 1e-3 from then on.  The oracle's own fp32 and fp64 trajectories part the same way
    (82 % of theta within 2e-5 after these four steps).  Bar: the first step exact up to such flips, afterwards 99 % of theta within
    4 x lr x 5e-3 and the losses within 2e-3."""
    spec, theta, X, f, y = _case(2048, nan=0.05, precision="bf16")
    eng = util.load_engine(spec, theta, X, f, y)
    eng.opt_init("Adam", 0.01)
    windows = [(0, 1024), (1024, 1024), (512, 1024), (0, 2048)]
    l1 = eng.train_step(*windows[0])
    th1, lr1 = ho.train_steps(spec, theta, X, f, y, windows[:1], dtype=np.float32)
    d1 = np.abs(eng.get_params() - th1)
    assert np.mean(d1 <= 1e-6) >= 0.999 and np.max(d1) <= 2.01 * 0.01 and abs(l1 - lr1[0]) <= 2e-5 * abs(lr1[0]), (np.mean(d1 <= 1e-6), np.max(d1))
    losses = [l1] + [eng.train_step(a, n) for a, n in windows[1:]]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, windows, dtype=np.float32)
    d = np.abs(eng.get_params() - th_ref)
    assert np.mean(d <= 2e-4) >= 0.99 and np.max(d) <= 2.5e-2 * 0.01 * len(windows) * 40, (np.mean(d <= 2e-4), np.max(d))
    assert np.allclose(losses, l_ref, rtol=2e-3)
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


# ---- config 5 with its 1e7 samples resident (BASELINE configs[4]: N = 1e7, 32 covariates, 3 forcings) ------------------------------
N_C5 = 10_000_000


def _big_c5(n, seed=5):
    """make_synth_c5's distributions drawn in float32 directly (3.6e8 values: the float64 detour of the small-case generator
    would take half a minute and 4 GB); 3 % of the targets missing"""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((32, n), dtype=np.float32)
    X *= np.float32(0.5)
    ta = rng.standard_normal(n, dtype=np.float32) * np.float32(10) + np.float32(10)
    sw = rng.random(n, dtype=np.float32) + np.float32(0.2)
    vpd = rng.random(n, dtype=np.float32) + np.float32(0.2)
    e = np.float32(0.1) * (ta - np.float32(15))
    w = [np.float32(1), sw, vpd]
    y = np.zeros(n, np.float32)
    for c in range(3):
        y += w[c] * (np.float32(1) + np.float32(0.8) * np.tanh(X[3 * c] + np.float32(0.5) * X[3 * c + 1])) * np.power(np.float32(1.6 + 0.4 * c), e)
    y *= np.float32(1) + np.float32(0.05) * rng.standard_normal(n, dtype=np.float32)
    y[rng.random(n, dtype=np.float32) < 0.03] = np.nan
    return X, {"ta": ta, "sw_in": sw, "vpd": vpd}, {"R_soil": y}


@pytest.fixture(scope="module")
def c5_resident():
    spec = ho.c5_spec()
    theta = ho.init_theta(spec, 3, np.float32)
    X, f, y = _big_c5(N_C5)
    eng = util.load_engine(spec, theta, X, f, y)               # 1.44 GB of 36-float records in HBM
    yield spec, theta, X, f, y, eng
    eng.close()


PREC_OPT = {"f32": 0, "bf16_fwd": 1, "bf16": 2}
PREC_TOL = {"f32": (1e-5, 1e-5), "bf16_fwd": (2e-5, 5e-5), "bf16": (2e-5, GTOL_BF16)}


@pytest.mark.parametrize("precision", ["bf16_fwd", "bf16", "f32"])
def test_windows_and_gathered_minibatches_at_the_end_of_1e7_resident_samples(c5_resident, precision):
    """record offsets near the end of a 1.44 GB array (360 M floats: 32-bit element offsets would still hold, 32-bit BYTE offsets
    would not): the last window, a window straddling nothing but the last records, and minibatches gathered from the last 1 %"""
    spec, theta, X, f, y, eng = c5_resident
    eng.set_option("precision", PREC_OPT[precision])
    sp = ho.c5_spec(precision=precision)
    ltol, gtol = PREC_TOL[precision]
    N = N_C5

    def oracle(ix):
        return ho.loss_and_grad(sp, theta.astype(np.float64), X[:, ix], {k: v[ix] for k, v in f.items()}, {k: v[ix] for k, v in y.items()})
    for first, count in ((N - 4096, 4096), (N - 1001, 1001), (N // 2 + 3, 2000)):
        loss, grad, nv = eng.loss_and_grad(eh.EH_SPLIT_TRAIN, first, count)
        l0, g0, nv0 = oracle(np.arange(first, first + count))
        assert nv == sum(nv0) and abs(loss - l0) <= ltol * abs(l0) and util.relerr(grad, g0) <= gtol, (first, loss, l0, util.relerr(grad, g0))
    rng = np.random.default_rng(17)
    idx = rng.choice(np.arange(N - N // 100, N), 3000, replace=False).astype(np.int32)
    idx[:3] = (N - 1, N - N // 100, N - 2)                      # the very last record is in
    loss, grad, nv = eng.loss_and_grad(idx=idx)
    l0, g0, nv0 = oracle(idx)
    assert nv == sum(nv0) and abs(loss - l0) <= ltol * abs(l0) and util.relerr(grad, g0) <= gtol
    with pytest.raises(ValueError):
        eng.loss_and_grad(eh.EH_SPLIT_TRAIN, N - 100, 101)      # one past the end


@pytest.mark.parametrize("precision,aot", [("bf16_fwd", 0), ("bf16", 0), ("f32", 0), ("bf16", 1), ("bf16_fwd", 1)])
def test_training_steps_on_full_size_minibatches_from_the_last_percent(c5_resident, precision, aot):
    """B = 65 536 gathered from the last 1 % (what a shuffled epoch's last steps read), as training steps: the step's loss equals
    the loss_and_grad of the same indices (HIP vs HIP, bit for bit the same pass), loss AND gradient equal the count-weighted sums
    over oracle-checked eighths, and plain descent moves theta by exactly -lr x that gradient.  aot = 1: the
    kernels specialised ahead of time for BASELINE configs[4] -- eh_spec_ns_6::eh_bfs_kernel for "bf16", what tools/bench_config.py c5
    times -- meet the oracle here in the mode and at the size that is benchmarked, with the 1e7 samples resident."""
    spec, theta, X, f, y, eng = c5_resident
    eng.set_option("aot_spec", aot)
    eng.set_option("precision", PREC_OPT[precision])
    eng.set_params(theta)
    sp = ho.c5_spec(precision=precision)
    N, B = N_C5, 65536
    rng = np.random.default_rng(23)
    idx = rng.choice(np.arange(N - N // 100, N), B, replace=False).astype(np.int32)
    loss, grad, nv = eng.loss_and_grad(idx=idx)
    if aot:
        assert eng.jit_status()[1].startswith("ahead-of-time"), eng.jit_status()[1][:200]
    acc_l, acc_n, acc_g = 0.0, 0, 0.0
    for q in range(0, B, 8192):
        ix = idx[q:q + 8192]
        l0, g0, nv0 = ho.loss_and_grad(sp, theta.astype(np.float64), X[:, ix], {k: v[ix] for k, v in f.items()}, {k: v[ix] for k, v in y.items()})
        acc_l += l0 * sum(nv0); acc_n += sum(nv0); acc_g = acc_g + g0 * sum(nv0)
    assert nv == acc_n and abs(loss - acc_l / acc_n) <= PREC_TOL[precision][0] * abs(loss)
    # (a one-target model carries its deltas un-normalised -- also where the "bf16" mode rounds them -- so the gradient of the whole
    #  minibatch IS the count-weighted sum of the eighths', at the mode's own gradient tolerance)
    assert util.relerr(grad, acc_g / acc_n) <= PREC_TOL[precision][1], util.relerr(grad, acc_g / acc_n)
    eng.opt_init("Descent", 0.05)
    step_loss = eng.train_step(0, B, idx=idx)
    assert step_loss == pytest.approx(loss, rel=1e-6)
    moved = (theta.astype(np.float64) - eng.get_params().astype(np.float64)) / 0.05
    assert util.relerr(moved, grad) <= 1e-5
    eng.set_params(theta)
    eng.set_option("aot_spec", 0)
