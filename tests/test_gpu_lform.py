"""-m gpu: the layer-wise execution form (csrc/eh_lform.hpp) -- networks no fused kernel holds: hidden widths above 128 or more
than three hidden layers, first of all the reference's own GPU tutorial net hidden_layers = [1024, 512, 256, 128, 64], sigmoid,
scale_nn_outputs, input_batchnorm (docs/literate/tutorials/synthetic_respiration_gpu.jl:79-92) -- through the C ABI against the
fp64 oracle at the north_star's 1e-5."""
import numpy as np
import pytest

import easyhybrid_jl_amd as eh
from oracle import hybrid_oracle as ho
from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-5
TUTORIAL = (1024, 512, 256, 128, 64)


def _check(spec, theta, X, f, y, eng=None, bn_state=None, **kw):
    own = eng is None
    eng = eng or util.load_engine(spec, theta, X, f, y)
    loss, grad, nv = eng.loss_and_grad(**kw)
    if "idx" in kw:
        ix = kw["idx"]
        X, f, y = X[:, ix], {k: v[ix] for k, v in f.items()}, {k: v[ix] for k, v in y.items()}
    l0, g0, nv0 = ho.loss_and_grad(spec, np.asarray(theta, np.float64), X, f, y, bn_state=bn_state)
    assert nv == sum(nv0)
    assert abs(loss - l0) <= TOL * abs(l0), (loss, l0)
    assert util.relerr(grad, g0) <= TOL, util.relerr(grad, g0)
    assert util.elem_relerr(grad, g0, 1e-3) <= 5e-4          # entry by entry, down to a thousandth of the largest one
    if own:
        eng.close()


@pytest.mark.parametrize("B", [64, 300, 2049])
def test_tutorial_network_loss_and_gradient(B):
    _check(*util.rbq10_case(B, "sigmoid", True, 0.1, hidden=TUTORIAL))


def test_tutorial_network_with_input_batchnorm_and_rmsprop():
    spec, theta, X, f, y = util.rbq10_case(512, "sigmoid", True, 0.05, hidden=TUTORIAL)
    X = (X * np.float32(50)).astype(np.float32)            # raw predictor scale: what the BatchNorm layer is there for
    spec.input_batchnorm = True
    eng = util.load_engine(spec, theta, X, f, y)
    _check(spec, theta, X, f, y, eng=eng, bn_state=ho.bn_init(spec))
    eng.opt_init("RMSProp", 0.001)                           # the tutorial's optimiser family (RMSProp(0.01) there)
    bn = ho.bn_init(spec)
    batches = [(0, 256), (256, 256), (100, 300)]
    losses = [eng.train_step(a, n) for a, n in batches]
    th, mm = theta.copy(), None
    # Optimisers.RMSProp(eta, rho = 0.9, eps = 1e-8): v = rho v + (1 - rho) g^2 ; theta -= eta g / (sqrt(v) + eps)
    v = np.zeros_like(th)
    for (a, n), l_dev in zip(batches, losses):
        sl = slice(a, a + n)
        l0, g0, _ = ho.loss_and_grad(spec, th.astype(np.float64), X[:, sl], {k: q[sl] for k, q in f.items()}, {k: q[sl] for k, q in y.items()}, bn_state=bn)
        _, new = ho.batchnorm_input(np.asarray(X[:, sl], np.float64), bn, True, np.dtype(np.float64)); bn.update(new)
        assert abs(l_dev - l0) <= 2e-5 * abs(l0)
        g = g0.astype(np.float32)
        v = np.float32(0.9) * v + np.float32(0.1) * g * g
        th = th - g * (np.float32(0.001) / (np.sqrt(v) + np.float32(1e-8)))
    d = np.abs(eng.get_params() - th)
    assert np.mean(d <= 2e-5) >= 0.999 and d.max() <= 2.5e-3, (np.mean(d <= 2e-5), d.max())      # first RMSProp steps are sign-like: see test_gpu_parity.py
    rm, rv = eng.get_bn_state()
    assert np.allclose(rm, bn["mean"], rtol=1e-5, atol=1e-6) and np.allclose(rv, bn["var"], rtol=1e-5, atol=1e-6)
    eng.close()


@pytest.mark.parametrize("act,hidden,n_pred", [("tanh", (40, 30, 20, 10), 8), ("relu", (200, 150), 12), ("identity", (129,), 3), ("tanh", (300,), 40)])
def test_other_deep_and_wide_shapes(act, hidden, n_pred):
    """four hidden layers of narrow width; two wide layers; one layer just past the fused kernels' 128; more predictors than the fused kernels take"""
    rng = np.random.default_rng(5)
    spec = ho.HybridSpec(n_pred, list(hidden), "expo2pool", dict(ho.EXPO2POOL_PARAMS), ["R0a", "ka", "R0b", "kb"], [], ["Resp_obs"], act, True)
    B = 700
    X = rng.random((n_pred, B)).astype(np.float32)
    f = {"T": (rng.random(B) * 40 - 10).astype(np.float32)}
    yv = (1.0 + rng.random(B)).astype(np.float32); yv[rng.random(B) < 0.1] = np.nan
    _check(spec, ho.init_theta(spec, 9, np.float32), X, f, {"Resp_obs": yv})


def test_large_batch_takes_the_128_tile_vector_load_gemms():
    """B = 16 384 on [8, 512, 256, 4]: every middle product has >= 256 tiles of 128 x 128 and aligned operands -- the 16-byte-load,
    double-buffered main loop in all three operand layouts (forward NN, delta NT, weight gradient TN); smaller batches run the
    64 x 64 tiles, unaligned shapes the 4-byte form, the first / last layer's weight gradients the streaming kernel"""
    rng = np.random.default_rng(15)
    B = 16384
    spec = ho.HybridSpec(8, [512, 256], "expo2pool", dict(ho.EXPO2POOL_PARAMS), ["R0a", "ka", "R0b", "kb"], [], ["Resp_obs"], "tanh", True)
    X = rng.random((8, B)).astype(np.float32)
    f = {"T": (rng.random(B) * 40 - 10).astype(np.float32)}
    yv = (1.0 + rng.random(B)).astype(np.float32); yv[rng.random(B) < 0.1] = np.nan
    theta = ho.init_theta(spec, 4, np.float32)
    eng = util.load_engine(spec, theta, X, f, {"Resp_obs": yv})
    _check(spec, theta, X, f, {"Resp_obs": yv}, eng=eng)
    # same answer from the 4-byte form (EH_GEMM_NOVEC is read once per process: compare against the oracle-checked gradient instead)
    l1, g1, _ = eng.loss_and_grad(first=0, count=2048)        # 64 x 64 tiles
    l0, g0, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X[:, :2048], {"T": f["T"][:2048]}, {"Resp_obs": yv[:2048]})
    assert abs(l1 - l0) <= TOL * abs(l0) and util.relerr(g1, g0) <= TOL
    eng.close()


def test_two_targets_global_and_fixed_parameters():
    rng = np.random.default_rng(8)
    B = 500
    pars = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}
    spec = ho.HybridSpec(6, [160, 96, 48, 24], "fluxpart", pars, ["RUE", "Rb"], ["Q10"], ["NEE", "GPP"], "tanh", True)
    X = rng.standard_normal((6, B)).astype(np.float32)
    f = {"SW_IN": (rng.random(B) * 400).astype(np.float32), "TA": (rng.random(B) * 30).astype(np.float32)}
    y = {"NEE": rng.standard_normal(B).astype(np.float32), "GPP": (rng.random(B) * 3).astype(np.float32)}
    y["NEE"][rng.random(B) < 0.2] = np.nan; y["GPP"][rng.random(B) < 0.1] = np.nan
    _check(spec, ho.init_theta(spec, 3, np.float32), X, f, y)


def test_forward_metrics_and_gathered_minibatch():
    spec, theta, X, f, y = util.rbq10_case(3000, "sigmoid", True, 0.1, hidden=(256, 192, 64, 32))
    eng = util.load_engine(spec, theta, X, f, y)
    out = eng.forward(eh.EH_SPLIT_TRAIN)
    ref = ho.forward(spec, theta.astype(np.float64), X, f)
    assert util.relerr(out["reco"], ref["reco"]) <= TOL and util.relerr(out["parameters"]["rb"], ref["parameters"]["rb"]) <= TOL
    m, _ = eng.eval(eh.EH_SPLIT_TRAIN)
    ev, _ = ho.evaluate(spec, theta.astype(np.float64), X, f, y, ("mse", "r2"))
    assert m[0]["mse"] == pytest.approx(ev["mse"]["reco"], rel=2e-5) and m[0]["r2"] == pytest.approx(ev["r2"]["reco"], abs=2e-5)
    idx = np.random.default_rng(2).permutation(3000)[:1100].astype(np.int32)
    _check(spec, theta, X, f, y, eng=eng, idx=idx)
    eng.close()


def test_adam_trajectory_and_epoch_driver():
    spec, theta, X, f, y = util.rbq10_case(1024, "tanh", True, 0.1, hidden=(160, 80, 40, 20))
    eng = util.load_engine(spec, theta, X, f, y)
    eng.opt_init("Adam", 0.01)
    batches = [(i * 256, 256) for i in range(4)]
    losses = [eng.train_step(a, n) for a, n in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32)
    assert np.allclose(losses, l_ref, rtol=2e-5)
    d = np.abs(eng.get_params() - th_ref)
    assert np.mean(d <= 2e-5) >= 0.999 and d.max() <= 4 * 0.01 * 1.01
    mean_loss, nsteps = eng.train_epoch(300, seed=5, shuffle=True)
    assert nsteps == 4 and np.isfinite(mean_loss)
    eng.close()


def test_weight_l2_and_data_parallel_seam():
    import torch
    spec, theta, X, f, y = util.rbq10_case(2048, "tanh", True, 0.1, hidden=(144, 72, 36, 18))
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_weight_l2(0.01, False)
    loss, grad, _ = eng.loss_and_grad()
    l0, g0, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, l2=(0.01, False))
    assert abs(loss - l0) <= TOL * abs(l0) and util.relerr(grad, g0) <= TOL
    eng.set_weight_l2(0.0, False)
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01); eng.opt_init("Adam", 0.01)
    l_ref = ref.train_step(0, 2048)
    ptr, n = eng.device_buffer(eh._lib.EH_BUF_GRAD)
    buf = torch.as_tensor(eh.dp._DevArray(ptr, n), device="cuda")
    acc = torch.zeros_like(buf)
    for k in range(4):                                    # four "ranks": the sum of their raw partial vectors stands in for the all-reduce
        eng.dp_grad(k * 512, 512); eng.synchronize(); acc += buf
    buf.copy_(acc); torch.cuda.synchronize()
    assert eng.dp_apply(want_loss=True) == pytest.approx(l_ref, rel=1e-5)
    # (the 512-sample shards and the 2 048-sample step run different product kernels -- other summation orders -- and the first Adam step
    #  is lr * g / (|g| + eps): a gradient entry near zero turns 1e-8 of rounding into 1e-5 of the 0.01 step)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 2e-5
    eng.close(); ref.close()


def test_train_front_door_on_the_tutorial_network():
    cols = eh.synthetic.make_synth_rbq10(1500, seed=4, nan_frac=0.05)
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=list(TUTORIAL), activation="sigmoid", scale_nn_outputs=True, input_batchnorm=True)
    out = eh.train(model, cols, nepochs=3, batchsize=64, opt=eh.RMSProp(0.001), loss_types=["mse", "nse"], random_seed=7, keep_history=True)
    assert len(out.val_history) == 4 and all(np.isfinite(h["mse"]["sum"]) for h in out.val_history)
    assert out.val_history[-1]["mse"]["sum"] < out.val_history[0]["mse"]["sum"]
    assert out.ps.size == 2 * 1024 + 1024 + 1024 * 512 + 512 + 512 * 256 + 256 + 256 * 128 + 128 + 128 * 64 + 64 + 64 + 1 + 1


def test_refusals():
    eng = util.model_from_spec(ho.rbq10_spec((256, 256), "tanh", True)).engine()
    with pytest.raises(NotImplementedError):
        eng.set_option("fused_update", 1)
    eng.close()


def test_graph_capture_needs_a_warm_up_step_and_says_so():
    """hipGraph capture (eh_graph_begin / _end / _launch) of steps of a layer-wise model: the form sizes its buffers at the first step of
    a batch size, which cannot happen inside a recording -- a clear error, the recording dropped, the engine still usable; after one
    warm-up step the same recording works and replays to the same parameters as eager steps"""
    spec, theta, X, f, y = util.rbq10_case(1024, "tanh", True, 0.1, hidden=(160, 64, 32, 16))
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01)
    eng.graph_begin()
    with pytest.raises(eh.EngineError, match="before eh_graph_begin"):
        eng.train_step(0, 256, want_loss=False)
    eng.train_step(0, 256, want_loss=False)                       # (eager: sizes the buffers; the engine survived the dropped recording)
    eng.set_params(theta); eng.opt_init("Adam", 0.01)
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01)
    eng.graph_begin()
    for s_ in range(4):
        eng.train_step(s_ * 256, 256, want_loss=False)
    g = eng.graph_end()
    eng.graph_launch(g); eng.synchronize()
    for s_ in range(4):
        ref.train_step(s_ * 256, 256, want_loss=False)
    assert np.array_equal(eng.get_params(), ref.get_params())
    eng.close(); ref.close()


def _huber(yh, y, delta=0.7):
    r = np.abs(yh - y)
    return np.mean(np.where(r <= delta, 0.5 * r * r, delta * (r - 0.5 * delta)))


@pytest.mark.parametrize("B", [64, 700, 3000])
def test_recorded_loss_function_on_a_network_the_fused_kernels_do_not_hold(B):
    """training_loss::Function (src/losses/loss_fn.jl:92-107) on the tutorial network's kind: no kernel is compiled at run time in the
    layer-wise form, the mechanistic kernel interprets the recorded tape (forward, then the reverse sweep from d l / d l = 1) -- loss and
    gradient against the oracle running the function itself, an Adam trajectory, and the named loss coming back afterwards"""
    spec, theta, X, f, y = util.rbq10_case(B, "sigmoid", True, 0.1, hidden=(160, 64, 32, 16))
    util.register_loss("huber_lform", _huber)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_training_loss(_huber)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind="huber_lform")
    lm, gm, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
    assert abs(l0 - lm) > 1e-3 * abs(lm)                       # (not mse in disguise)
    assert nv == sum(nv0) and abs(loss - l0) <= TOL * abs(l0) and util.relerr(grad, g0) <= TOL, (loss, l0, util.relerr(grad, g0))
    eng.opt_init("Adam", 0.01)
    n2 = B // 2
    batches = [(0, n2), (n2, B - n2)]
    losses = [eng.train_step(a, n) for a, n in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32, kind="huber_lform")
    assert np.allclose(losses, l_ref, rtol=2e-5)
    d = np.abs(eng.get_params() - th_ref)
    assert np.mean(d <= 2e-5) >= 0.999
    eng.set_params(theta)
    eng.set_training_loss("mse")
    l1, g1, _ = eng.loss_and_grad()
    assert abs(l1 - lm) <= TOL * abs(lm) and util.relerr(g1, gm) <= TOL
    eng.close()


def test_train_front_door_with_a_recorded_loss_on_a_deep_network():
    cols = eh.synthetic.make_synth_rbq10(1500, seed=4, nan_frac=0.05)
    cols = dict(cols); cols["sw_pot"] = cols["sw_pot"] / 50; cols["dsw_pot"] = cols["dsw_pot"] / 50
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=[160, 64, 32, 16], activation="sigmoid", scale_nn_outputs=True)
    out = eh.train(model, cols, nepochs=4, batchsize=128, opt=eh.Adam(0.003), training_loss=_huber, loss_types=["mse", "mae"], random_seed=7, keep_history=True)
    assert all(np.isfinite(h["mse"]["sum"]) for h in out.val_history)
    assert out.val_history[-1]["mae"]["sum"] < out.val_history[0]["mae"]["sum"]


def test_recorded_losses_per_target_in_the_layer_wise_form():
    """PerTarget((f, :mse)) and PerTarget((:mae, f)) on a two-target model that runs layer by layer (compute_loss.jl:128-145)"""
    def pseudo_huber(yh, y):
        r = yh - y
        return np.mean(np.sqrt(1.0 + r * r) - 1.0)
    rng = np.random.default_rng(8)
    B = 500
    pars = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}
    spec = ho.HybridSpec(6, [160, 96, 48, 24], "fluxpart", pars, ["RUE", "Rb"], ["Q10"], ["NEE", "GPP"], "tanh", True)
    X = rng.standard_normal((6, B)).astype(np.float32)
    f = {"SW_IN": (rng.random(B) * 400).astype(np.float32), "TA": (rng.random(B) * 30).astype(np.float32)}
    y = {"NEE": rng.standard_normal(B).astype(np.float32), "GPP": (rng.random(B) * 3).astype(np.float32)}
    y["NEE"][rng.random(B) < 0.2] = np.nan; y["GPP"][rng.random(B) < 0.1] = np.nan
    theta = ho.init_theta(spec, 3, np.float32)
    name = util.register_loss("pseudo_huber_lform", pseudo_huber) and "pseudo_huber_lform"
    for kinds, dev in (((name, name), pseudo_huber), ((name, "mse"), eh.PerTarget((pseudo_huber, "mse"))), (("mae", name), eh.PerTarget(("mae", pseudo_huber)))):
        eng = util.load_engine(spec, theta, X, f, y)
        eng.set_training_loss(dev)
        loss, grad, nv = eng.loss_and_grad()
        l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=kinds)
        assert nv == sum(nv0) and abs(loss - l0) <= TOL * abs(l0) and util.relerr(grad, g0) <= 2e-5, (kinds, loss, l0, util.relerr(grad, g0))
        eng.close()


# ---- what the layer-wise form did not hold in round 2: swish, MultiNNHybridModel, the moment losses ----------------------------------
@pytest.mark.parametrize("hidden", [TUTORIAL, (200, 40), (96, 80, 64, 48)])
def test_swish_networks(hidden):
    """swish (README.md:186) on networks no fused kernel holds: act' needs the PRE-activation, which the layer-wise form now keeps next
    to the activation for swish layers"""
    B = 700
    spec, theta, X, f, y = util.rbq10_case(B, "swish", True, 0.1, hidden=hidden)
    _check(spec, theta, X, f, y)
    if hidden == TUTORIAL:
        return
    eng = util.load_engine(spec, theta, X, f, y)
    eng.opt_init("Adam", 0.01)
    batches = [(0, 350), (350, 350)]
    losses = [eng.train_step(a, n) for a, n in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32)
    assert np.allclose(losses, l_ref, rtol=2e-5)
    d = np.abs(eng.get_params() - th_ref)
    assert np.mean(d <= 2e-5) >= 0.999
    out = eng.forward(eh.EH_SPLIT_TRAIN)
    assert util.relerr(out["reco"], ho.forward(spec, eng.get_params().astype(np.float64), X, f)["reco"]) <= TOL
    eng.close()


MULTI = [
    # (nets: (predictor rows, hidden widths) per neural parameter, activations or None, globals)
    ([([0, 1], [160, 40]), ([2, 3, 4], [96, 24])], None, ["Q10"]),                                  # wider than any fused envelope
    ([([0], [40, 30, 20, 10]), ([1, 2], [24, 24, 24, 24])], None, []),                              # four hidden layers
    ([([0, 1], [140, 30]), ([2], [20]), ([3, 4, 5], [64, 64, 16])], ["tanh", "swish", "relu"], ["Q10_het", "Q10_root", "Q10_myc"]),   # depths 2 / 1 / 3, own activations
]


@pytest.mark.parametrize("case", range(len(MULTI)))
def test_multinn_models(case):
    """MultiNNHybridModel (src/models/GenericHybridModel.jl:142-206,458-530) beyond the block-diagonal envelope of the fused
    kernels: every network runs as its own chain of products on its own predictor columns"""
    nets, acts, glob = MULTI[case]
    rng = np.random.default_rng(20 + case)
    B = 900
    if len(nets) == 2:
        pars = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}
        neural = ["RUE", "Rb"]
        spec = ho.HybridSpec(max(max(r) for r, _ in nets) + 1, [], "fluxpart", pars, neural, glob, ["NEE", "GPP"], "sigmoid", True, nets=nets, net_activations=acts)
        f = {"SW_IN": (rng.random(B) * 400).astype(np.float32), "TA": (rng.random(B) * 30).astype(np.float32)}
        y = {"NEE": rng.standard_normal(B).astype(np.float32), "GPP": (rng.random(B) * 3).astype(np.float32)}
        y["NEE"][rng.random(B) < 0.2] = np.nan
    else:
        neural = ["Rb_het", "Rb_root", "Rb_myc"]
        spec = ho.HybridSpec(6, [], "rs_components", dict(ho.RS6_PARAMS), neural, glob, ["R_soil"], "tanh", True, nets=nets, net_activations=acts)
        f = {"ta": (10 + 10 * rng.standard_normal(B)).astype(np.float32)}
        y = {"R_soil": (rng.random(B) * 5 + 0.5).astype(np.float32)}
        y["R_soil"][rng.random(B) < 0.1] = np.nan
    X = rng.standard_normal((spec.n_pred, B)).astype(np.float32)
    theta = ho.init_theta(spec, 5, np.float32)
    eng = util.load_engine(spec, theta, X, f, y)
    _check(spec, theta, X, f, y, eng=eng)
    idx = rng.permutation(B)[:333].astype(np.int32)
    _check(spec, theta, X, f, y, eng=eng, idx=idx)
    eng.opt_init("Adam", 0.01)
    batches = [(0, 450), (450, 450)]
    losses = [eng.train_step(a, n) for a, n in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32)
    assert np.allclose(losses, l_ref, rtol=2e-5)
    d = np.abs(eng.get_params() - th_ref)
    assert np.mean(d <= 2e-5) >= 0.998
    eng.close()


@pytest.mark.parametrize("B", [100, 511, 512, 1000, 1024, 1025])
def test_few_rows_products_and_grouped_weight_gradients(B):
    """Up to 1 024 rows the forward / delta products with an aligned k >= 64 run one output tile per workgroup with k split over its
    waves (16 x 16 tiles below 512 rows, 32 x 32 from there on; output widths that are not multiples of the tile: 80, 144), and up to
    4 096 rows every layer's delta is kept and the weight gradients follow as grouped launches -- here nine tiled ones (more than one
    group holds) and six thin ones; every boundary against the fp64 oracle, contiguous and gathered."""
    nets = [([0, 1], [64, 64, 64, 32]), ([2, 3], [64, 64, 64, 32]), ([4, 5], [128, 144, 80, 16])]
    rng = np.random.default_rng(40 + B)
    spec = ho.HybridSpec(6, [], "rs_components", dict(ho.RS6_PARAMS), ["Rb_het", "Rb_root", "Rb_myc"], ["Q10_het", "Q10_root", "Q10_myc"], ["R_soil"], "tanh", True,
                         nets=nets, net_activations=["tanh", "sigmoid", "swish"])
    n = B + 300
    X = rng.standard_normal((6, n)).astype(np.float32)
    f = {"ta": (10 + 10 * rng.standard_normal(n)).astype(np.float32)}
    y = {"R_soil": (rng.random(n) * 5 + 0.5).astype(np.float32)}
    y["R_soil"][rng.random(n) < 0.1] = np.nan
    theta = ho.init_theta(spec, 5, np.float32)
    eng = util.load_engine(spec, theta, X, f, y)
    sl = slice(100, 100 + B)
    loss, grad, nv = eng.loss_and_grad(eh.EH_SPLIT_TRAIN, 100, B)
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X[:, sl], {k: v[sl] for k, v in f.items()}, {k: v[sl] for k, v in y.items()})
    assert nv == sum(nv0) and abs(loss - l0) <= TOL * abs(l0) and util.relerr(grad, g0) <= TOL, (loss, l0, util.relerr(grad, g0))
    assert util.elem_relerr(grad, g0, 1e-3) <= 5e-4
    idx = rng.permutation(n)[:B].astype(np.int32)
    _check(spec, theta, X, f, y, eng=eng, idx=idx)
    eng.close()


@pytest.mark.parametrize("kind", ["kgeLoss", "pearsonLoss", "pbkgeLoss"])
def test_moment_losses(kind):
    spec, theta, X, f, y = util.rbq10_case(1500, "tanh", True, 0.1, hidden=(192, 64, 32, 16))
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_training_loss(kind)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=kind)
    assert nv == sum(nv0) and abs(loss - l0) <= 1e-4 * abs(l0) and util.relerr(grad, g0) <= 1e-4, (loss, l0, util.relerr(grad, g0))
    eng.close()


@pytest.mark.parametrize("B", [1, 7, 64, 200, 256, 257])
@pytest.mark.parametrize("hidden", [(256, 128, 144), (300, 130), (512, 64, 32, 16)])
def test_small_minibatches_take_the_split_k_and_streaming_kernels(B, hidden):
    """M <= 256 rows: products with K >= 128 (a multiple of 16, aligned operands) are split over k into partial products plus a combine
    pass; products with a degenerate dimension are streaming kernels; the others stay on the tiled kernel.  Every batch size around the
    boundaries, against the fp64 oracle; and the bits must not depend on which way a product went beyond the summation order (1e-5)."""
    spec, theta, X, f, y = util.rbq10_case(max(B, 8), "tanh", True, 0.1 if B > 8 else 0.0, hidden=hidden)
    eng = util.load_engine(spec, theta, X, f, y)
    loss, grad, nv = eng.loss_and_grad(eh.EH_SPLIT_TRAIN, 0, B)
    sl = slice(0, B)
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X[:, sl], {k: v[sl] for k, v in f.items()}, {k: v[sl] for k, v in y.items()})
    assert nv == sum(nv0) and abs(loss - l0) <= TOL * abs(l0) and util.relerr(grad, g0) <= TOL, (loss, l0, util.relerr(grad, g0))
    eng.close()


# ---- round 6: few rows -- the narrow end of a one-network model as ONE launch (eh_lform_tailchain_kernel) ---------------------------
def _expo2pool_case(n_pred, hidden, act, B, seed=5, nan=0.1):
    rng = np.random.default_rng(seed)
    spec = ho.HybridSpec(n_pred, list(hidden), "expo2pool", dict(ho.EXPO2POOL_PARAMS), ["R0a", "ka", "R0b", "kb"], [], ["Resp_obs"], act, True)
    X = rng.random((n_pred, B)).astype(np.float32)
    f = {"T": (rng.random(B) * 40 - 10).astype(np.float32)}
    yv = (1.0 + rng.random(B)).astype(np.float32); yv[rng.random(B) < nan] = np.nan
    return spec, ho.init_theta(spec, 9, np.float32), X, f, {"Resp_obs": yv}


@pytest.mark.parametrize("B", [1, 37, 64, 65, 130, 256])
def test_few_rows_tutorial_network(B):
    """one row per workgroup up to 64 rows, four from there to 256 (partial last workgroup, a single sample, the largest minibatch the
    path takes); 257 rows and more run the products launch by launch as before"""
    _check(*util.rbq10_case(B, "sigmoid", True, 0.1, hidden=TUTORIAL))


@pytest.mark.parametrize("act,hidden,n_pred,B", [
    ("tanh", (40, 30, 20, 10), 8, 64),          # every layer narrow: the whole network is the suffix (its first layer reads the minibatch matrix); out = 30, 10: 4-byte loads
    ("swish", (96, 80, 64, 48), 8, 200),        # pre-activations kept for act'
    ("relu", (300, 150), 12, 100),              # 12 -> 300 stays a product of its own (300 > 256); the suffix starts at a layer whose input it reads back
    ("swish", (512, 200, 36), 5, 50),           # the layer BELOW the suffix is swish: its pre-activation is read for the delta that leaves the chain
    ("identity", (129,), 3, 64),
])
def test_few_rows_other_shapes(act, hidden, n_pred, B):
    _check(*_expo2pool_case(n_pred, hidden, act, B))


def test_few_rows_gathered_minibatch_two_targets_and_training():
    rng = np.random.default_rng(8)
    N = 500
    pars = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}
    spec = ho.HybridSpec(6, [160, 96, 48, 24], "fluxpart", pars, ["RUE", "Rb"], ["Q10"], ["NEE", "GPP"], "tanh", True)
    X = rng.standard_normal((6, N)).astype(np.float32)
    f = {"SW_IN": (rng.random(N) * 400).astype(np.float32), "TA": (rng.random(N) * 30).astype(np.float32)}
    y = {"NEE": rng.standard_normal(N).astype(np.float32), "GPP": (rng.random(N) * 3).astype(np.float32)}
    y["NEE"][rng.random(N) < 0.2] = np.nan; y["GPP"][rng.random(N) < 0.1] = np.nan
    theta = ho.init_theta(spec, 3, np.float32)
    eng = util.load_engine(spec, theta, X, f, y)
    for nb in (64, 150):
        idx = rng.permutation(N)[:nb].astype(np.int32)
        _check(spec, theta, X, f, y, eng=eng, idx=idx)
    eng.close()
    # an Adam trajectory over minibatches of 64 and the epoch driver, the reference's default batch size (src/config/TrainingConfig.jl:14)
    spec, theta, X, f, y = util.rbq10_case(1024, "tanh", True, 0.1, hidden=(160, 80, 40, 20))
    eng = util.load_engine(spec, theta, X, f, y)
    eng.opt_init("Adam", 0.01)
    batches = [(i * 64, 64) for i in range(6)] + [(400, 200)]
    losses = [eng.train_step(a, n) for a, n in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32)
    assert np.allclose(losses, l_ref, rtol=2e-5)
    d = np.abs(eng.get_params() - th_ref)
    assert np.mean(d <= 2e-5) >= 0.999 and d.max() <= 8 * 0.01 * 1.01
    mean_loss, nsteps = eng.train_epoch(64, seed=5, shuffle=True)
    assert nsteps == 16 and np.isfinite(mean_loss)
    eng.close()


def test_few_rows_recorded_closure_and_recorded_loss():
    """a recorded mechanistic closure and a recorded loss function at the tutorial's batch size: the chain's mechanistic stage interprets both tapes"""
    from tests import closures
    spec, theta, X, f, y = util.rbq10_case(64, "sigmoid", True, 0.1, hidden=(160, 64, 32, 16))
    util.register_loss("huber_lform", _huber)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_training_loss(_huber)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind="huber_lform")
    assert nv == sum(nv0) and abs(loss - l0) <= TOL * abs(l0) and util.relerr(grad, g0) <= TOL, (loss, l0, util.relerr(grad, g0))
    eng.close()
    # the three-output flux closure (tests/closures.py) behind a network only the layer-wise form holds, two of its outputs as targets
    util.register_closure("flux_closure", closures.flux_closure, list(closures.FLUX_TABLE), ["sw", "ta", "vpd"], ["nee", "gpp"])
    spec = ho.HybridSpec(3, [144, 72, 36, 18], "flux_closure", dict(closures.FLUX_TABLE), ["alpha", "gmax", "rref"], ["e0", "k"], ["nee", "gpp"], "tanh", True)
    rng = np.random.default_rng(31)
    B = 100
    X = rng.uniform(-1, 1, (3, B)).astype(np.float32)
    f = {"sw": rng.uniform(0, 800, B).astype(np.float32), "ta": rng.uniform(-5, 30, B).astype(np.float32), "vpd": rng.uniform(0, 30, B).astype(np.float32)}
    truth = ho.forward(spec, ho.init_theta(spec, 33, np.float32).astype(np.float64), X, f)
    y = {t: (truth[t] * (1.0 + 0.05 * rng.normal(size=B))).astype(np.float32) for t in spec.targets}
    y["nee"][rng.uniform(size=B) < 0.1] = np.nan
    _check(spec, ho.init_theta(spec, 32, np.float32), X, f, y)


@pytest.mark.parametrize("rule", ["AdamW", "Descent", "Adam"])
@pytest.mark.parametrize("B", [64, 40])
def test_few_rows_step_with_every_optimiser_rule_in_the_weight_gradient_launch(rule, B):
    """the optimiser in the epilogues of the weight-gradient launch (eh_dw_apply64_kernel: tiles, thin products, bias sums, global parameters)
    for the rules the tutorial does not use: four steps at the tutorial's batch size (and a smaller one: partial tiles) against the oracle's
    fp32 trajectory, Optimisers.jl's rules op for op"""
    spec, theta, X, f, y = util.rbq10_case(4 * B, "tanh", True, 0.1, hidden=TUTORIAL)
    eng = util.load_engine(spec, theta, X, f, y)
    lr, wd = (0.003, 0.05) if rule == "AdamW" else ((0.05, 0.0) if rule == "Descent" else (0.003, 0.0))
    eng.opt_init(rule, lr, weight_decay=wd)
    batches = [(i * B, B) for i in range(4)]
    losses = [eng.train_step(a, n) for a, n in batches]
    th = theta.copy(); st = ho.adam_init(theta.size, np.float32); l_ref = []
    for a, n in batches:
        sl = slice(a, a + n)
        l_, g_, _ = ho.loss_and_grad(spec, th, X[:, sl], {k: v[sl] for k, v in f.items()}, {k: v[sl] for k, v in y.items()}, dtype=np.float32)
        l_ref.append(float(l_))
        g = g_.astype(np.float32)
        th = (th - np.float32(lr) * g).astype(np.float32) if rule == "Descent" else ho.adam_step(th, g, st, lr, weight_decay=wd)
    assert np.allclose(losses, l_ref, rtol=5e-5), (losses, l_ref)
    d = np.abs(eng.get_params() - th)
    assert np.mean(d <= 2e-5) >= 0.999 and d.max() <= 4 * lr * 1.01 + 1e-4, (np.mean(d <= 2e-5), d.max())
    eng.close()
