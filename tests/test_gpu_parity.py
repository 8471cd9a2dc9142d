"""-m gpu parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes --
through size-independent properties.  Tolerance: north_star = 1e-5 relative (fp32); the oracle is
evaluated in fp64 so its own rounding does not eat the budget."""
import glob
import json
import os

import numpy as np
import pytest

import easyhybrid_jl_amd as eh
from oracle import hybrid_oracle as ho
from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _check_grad(spec, theta, X, f, y, eng=None, **kw):
    own = eng is None
    eng = eng or util.load_engine(spec, theta, X, f, y)
    loss, grad, nv = eng.loss_and_grad(**kw)
    if "idx" in kw:
        ix = kw["idx"]
        X, f, y = X[:, ix], {k: v[ix] for k, v in f.items()}, {k: v[ix] for k, v in y.items()}
    elif "first" in kw:
        sl = slice(kw["first"], kw["first"] + kw["count"])
        X, f, y = X[:, sl], {k: v[sl] for k, v in f.items()}, {k: v[sl] for k, v in y.items()}
    l0, g0, nv0 = ho.loss_and_grad(spec, np.asarray(theta, np.float64), X, f, y)
    assert nv == sum(nv0)
    assert abs(loss - l0) <= TOL * abs(l0)
    assert util.relerr(grad, g0) <= TOL
    if own:
        eng.close()


# ----------------------------------------------------------------------------------------------
# loss + gradient: activations x scaling x mask patterns x ragged sizes (RbQ10 = BASELINE configs 0/1)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("act", ["tanh", "sigmoid", "relu", "swish"])
@pytest.mark.parametrize("scale", [False, True])
@pytest.mark.parametrize("B,nan_frac", [(12, 0.0), (64, 0.2), (1024, 0.2), (1000, 0.05), (1, 0.0), (65, 0.5)])
def test_rbq10_loss_and_grad(act, scale, B, nan_frac):
    _check_grad(*util.rbq10_case(B, act, scale, nan_frac))


@pytest.mark.parametrize("hidden", [(16,), (32, 32), (16, 16, 16), (24, 8), (64, 64), (48, 33, 16), (64,), (40, 64, 24)])
def test_other_mlp_shapes(hidden):
    _check_grad(*util.rbq10_case(300, "tanh", True, 0.1, hidden=hidden))


def test_identity_activation():
    spec, theta, X, f, y = util.rbq10_case(200, "tanh", False, 0.1)
    spec.activation = "identity"
    _check_grad(spec, theta, X, f, y)


def test_expo2pool_config3_shape():
    # BASELINE.json configs[2]: MLP [8,64,64,4], four neural parameters, build-defined two-pool Expo model
    spec = ho.expo2pool_spec((64, 64), "tanh", True)
    X, f, y = ho.make_synth_expo2pool(1000, 5, 0.1)
    _check_grad(spec, ho.init_theta(spec, 2, np.float32), X, f, y)


def test_reference_expo_model_one_predictor():
    # projects/ExpoHybrid/ExpoHybridEstim.jl: Resp0 neural, k global, sigmoid activation (without its BatchNorm)
    spec = ho.HybridSpec(1, [16, 16], "expo", dict(ho.EXPO_PARAMS), ["Resp0"], ["k"], ["Resp_obs"], "sigmoid", False)
    rng = np.random.default_rng(0)
    T = (rng.random(500) * 40 - 10).astype(np.float32)
    SM = (rng.random(500) * 0.8 + 0.1).astype(np.float32)
    resp = (1.1 * np.exp(-8.0 * (SM - 0.6) ** 2) * np.exp(0.07 * T)).astype(np.float32)
    _check_grad(spec, ho.init_theta(spec, 3, np.float32), SM[None], {"T": T}, {"Resp_obs": resp})


@pytest.mark.parametrize("mech,neural,glob", [
    ("linear", ["alpha"], ["beta"]),
    ("linear", ["alpha", "beta"], []),
    ("rs_components", ["Rb_het", "Rb_root", "Rb_myc"], ["Q10_het", "Q10_root", "Q10_myc"]),
    ("rs_components", ["Rb_het", "Rb_root", "Rb_myc"], ["Q10_het"]),          # two Q10s fixed at their defaults
    ("rbq10", ["rb", "Q10"], []),                                             # Q10 predicted by the network too
    ("expo2pool", ["R0a", "R0b"], ["ka", "kb"]),
])
def test_parameter_sources_neural_global_fixed(mech, neural, glob):
    mm = ho.MECH[mech][0]
    tabs = {"linear": {"alpha": (1.0, -2.0, 3.0), "beta": (0.5, -1.0, 2.0)}, "rbq10": dict(ho.RBQ10_PARAMS), "expo2pool": dict(ho.EXPO2POOL_PARAMS),
            "rs_components": {**{f"Rb_{c}": (1.0, 0.0, 5.0) for c in ("het", "root", "myc")}, **{f"Q10_{c}": (2.0 + i * 0.3, 1.0, 4.0) for i, c in enumerate(("het", "root", "myc"))}}}
    spec = ho.HybridSpec(5, [32, 16], mech, tabs[mech], neural, glob, [mm.outputs[0]], "tanh", True)
    rng = np.random.default_rng(1)
    B = 400
    X = rng.standard_normal((5, B)).astype(np.float32)
    frc = {mm.forcings[0]: rng.uniform(-5, 25, B).astype(np.float32)}
    yv = rng.uniform(0.5, 4, B).astype(np.float32)
    yv[rng.random(B) < 0.1] = np.nan
    _check_grad(spec, ho.init_theta(spec, 4, np.float32), X, frc, {mm.outputs[0]: yv})


def test_wide_input_two_blocks():
    # P = 20 > 16 exercises the second input block and the scalar (C % 4 != 0) record path: C = 20 + 1 + 1
    spec = ho.HybridSpec(20, [32, 32], "rbq10", dict(ho.RBQ10_PARAMS), ["rb"], ["Q10"], ["reco"], "tanh", True)
    rng = np.random.default_rng(2)
    B = 333
    X = rng.standard_normal((20, B)).astype(np.float32) * 0.3
    _check_grad(spec, ho.init_theta(spec, 5, np.float32), X, {"ta": rng.uniform(0, 30, B).astype(np.float32)},
                {"reco": rng.uniform(1, 9, B).astype(np.float32)})


def test_window_and_gather_indices():
    spec, theta, X, f, y = util.rbq10_case(2000, "tanh", True, 0.1)
    eng = util.load_engine(spec, theta, X, f, y)
    _check_grad(spec, theta, X, f, y, eng, first=137, count=701)
    idx = np.random.default_rng(3).permutation(2000)[:555].astype(np.int32)
    _check_grad(spec, theta, X, f, y, eng, idx=idx)
    with pytest.raises(ValueError):
        eng.loss_and_grad(first=1900, count=200)
    with pytest.raises(ValueError):
        eng.loss_and_grad(idx=np.array([0, 2000], np.int32))
    eng.close()


def test_all_masked_batch():
    spec, theta, X, f, y = util.rbq10_case(128, "tanh", True)
    y = {"reco": np.full(128, np.nan, np.float32)}
    eng = util.load_engine(spec, theta, X, f, y)
    loss, grad, nv = eng.loss_and_grad()
    assert np.isnan(loss) and nv == 0 and not grad.any()
    eng.opt_init("Adam", 0.01)
    l = eng.train_step(0, 128)
    assert np.isnan(l) and np.array_equal(eng.get_params(), theta)            # skipped: src/training/epoch.jl:17-19
    m, v, bt = eng.get_opt_state()
    assert not m.any() and not v.any() and bt.tolist() == [np.float32(0.9), np.float32(0.999)]
    eng.close()


# ----------------------------------------------------------------------------------------------
# forward / eval
# ----------------------------------------------------------------------------------------------
def test_forward_matches_oracle():
    spec, theta, X, f, y = util.rbq10_case(300, "tanh", True, 0.1)
    eng = util.load_engine(spec, theta, X, f, y)
    out = eng.forward(0)
    ref = ho.forward(spec, theta.astype(np.float64), X, f)
    assert util.relerr(out["reco"], ref["reco"]) <= TOL
    assert util.relerr(out["parameters"]["rb"], ref["parameters"]["rb"]) <= TOL
    assert util.relerr(out["parameters"]["Q10"], np.broadcast_to(ref["parameters"]["Q10"], (300,))) <= TOL
    part = eng.forward(0, 100, 50, params=False)
    assert np.array_equal(part["reco"], out["reco"][100:150])
    eng.close()


def test_eval_metrics_match_loss_fn():
    spec, theta, X, f, y = util.rbq10_case(5000, "tanh", True, 0.15)
    eng = util.load_engine(spec, theta, X, f, y, split=1)
    metrics, pred = eng.eval(1, predictions=True)
    ref = ho.forward(spec, theta.astype(np.float64), X, f)["reco"]
    yy = y["reco"].astype(np.float64)
    mask = ~np.isnan(yy)
    assert metrics[0]["n"] == mask.sum()
    for k in ("mse", "rmse", "mae", "r2", "nse", "pearson", "kge", "pbkge"):
        assert metrics[0][k] == pytest.approx(ho.loss_fn(ref, yy, mask, k), rel=2e-5, abs=2e-6), k
    assert metrics[0]["beta"] == pytest.approx(ho.loss_fn(ref, yy, mask, "β"), rel=2e-5)
    assert metrics[0]["alpha"] == pytest.approx(ho.loss_fn(ref, yy, mask, "α"), rel=2e-5)
    assert util.relerr(pred["reco"], ref) <= TOL
    eng.close()


# ----------------------------------------------------------------------------------------------
# optimiser trajectory
# ----------------------------------------------------------------------------------------------
def test_adam_trajectory_matches_oracle():
    spec, theta, X, f, y = util.rbq10_case(1024, "tanh", True, 0.1)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.opt_init("Adam", 0.01)
    batches = [(i * 128, 128) for i in range(8)]
    losses = [eng.train_step(a, b) for a, b in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32)
    assert np.allclose(losses, l_ref, rtol=1e-4)
    # Adam's first steps are sign-like (m/sqrt(v) ~ +-1): rounding in g flips nothing but shows up at ~1e-6 absolute
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 2e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    m, v, bt = eng.get_opt_state()
    assert bt[0] == pytest.approx(np.float32(0.9) ** 9, rel=1e-6) and bt[1] == pytest.approx(np.float32(0.999) ** 9, rel=1e-6)
    eng.close()


@pytest.mark.parametrize("rule,kw", [("Descent", dict(lr=0.05)), ("RMSProp", dict(lr=0.003, beta1=0.9)), ("AdamW", dict(lr=0.01, weight_decay=0.1))])
def test_other_optimiser_rules(rule, kw):
    spec, theta, X, f, y = util.rbq10_case(512, "tanh", True, 0.1)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.opt_init(rule, **kw)
    th = theta.astype(np.float32).copy()
    st = ho.adam_init(th.size)
    vq = np.zeros_like(th)
    for a in range(0, 512, 128):
        eng.train_step(a, 128)
        sl = slice(a, a + 128)
        _, g, _ = ho.loss_and_grad(spec, th, X[:, sl], {k: v[sl] for k, v in f.items()}, {k: v[sl] for k, v in y.items()}, np.float32)
        g = g.astype(np.float32)
        if rule == "Descent":
            th = th - np.float32(kw["lr"]) * g
        elif rule == "RMSProp":
            vq = np.float32(0.9) * vq + np.float32(0.1) * g * g
            th = th - g * (np.float32(kw["lr"]) / (np.sqrt(vq) + np.float32(1e-8)))
        else:
            th = ho.adam_step(th, g, st, lr=kw["lr"], weight_decay=kw["weight_decay"])
    assert np.max(np.abs(eng.get_params() - th)) <= 3e-5 * max(1.0, float(np.max(np.abs(th))))
    eng.close()


def test_set_get_opt_state_round_trip_resumes_identically():
    spec, theta, X, f, y = util.rbq10_case(512, "tanh", True)
    a = util.load_engine(spec, theta, X, f, y); a.opt_init("Adam", 0.01)
    for i in range(3):
        a.train_step(i * 128, 128)
    b = util.load_engine(spec, a.get_params(), X, f, y); b.opt_init("Adam", 0.01)
    b.set_opt_state(*a.get_opt_state())
    a.train_step(384, 128); b.train_step(384, 128)
    assert np.array_equal(a.get_params(), b.get_params())
    a.close(); b.close()


# ----------------------------------------------------------------------------------------------
# epoch driver
# ----------------------------------------------------------------------------------------------
def test_epoch_unshuffled_equals_manual_steps_with_partial_last_batch():
    spec, theta, X, f, y = util.rbq10_case(1000, "tanh", True, 0.1)
    a = util.load_engine(spec, theta, X, f, y); a.opt_init("Adam", 0.01)
    b = util.load_engine(spec, theta, X, f, y); b.opt_init("Adam", 0.01)
    mean_loss, ns = a.train_epoch(300, shuffle=False)
    losses = [b.train_step(s, min(300, 1000 - s)) for s in range(0, 1000, 300)]
    assert ns == 4 and np.array_equal(a.get_params(), b.get_params())
    assert mean_loss == pytest.approx(np.mean(losses), rel=1e-6)
    a.close(); b.close()


def test_epoch_shuffle_is_a_permutation_and_trains():
    # with Descent(lr=0) nothing moves, so the mean over the shuffled full-batch epoch must equal the
    # unshuffled one: every sample is visited exactly once whatever the permutation
    spec, theta, X, f, y = util.rbq10_case(4096, "tanh", True, 0.1)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.opt_init("Descent", 0.0)
    l_plain, _ = eng.train_epoch(4096, shuffle=False)
    l_shuf, _ = eng.train_epoch(4096, seed=7, shuffle=True)
    assert l_shuf == pytest.approx(l_plain, rel=1e-5)
    eng.opt_init("Adam", 0.01)
    first, _ = eng.train_epoch(256, seed=1, shuffle=True)
    for e in range(2, 12):
        last, _ = eng.train_epoch(256, seed=e, shuffle=True)
    assert last < 0.7 * first
    eng.close()


def test_train_front_door_runs_and_improves():
    cols = eh.synthetic.make_synth_rbq10(4000, seed=3, nan_frac=0.05)
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    out = eh.train(model, cols, nepochs=6, batchsize=256, opt=eh.Adam(0.01), random_seed=1)
    assert len(out.train_history) == 7 and len(out.val_history) == 7          # initial snapshot + one per epoch (history.jl)
    assert set(out.val_history[0]) == {"mse", "r2"} and set(out.val_history[0]["mse"]) == {"reco", "sum"}
    assert out.val_history[-1]["mse"]["sum"] < out.val_history[0]["mse"]["sum"]
    assert out.best_epoch >= 1 and out.ps.size == 338 and "reco_pred" in out.val_obs_pred
    out2 = eh.train(model, cols, nepochs=2, batchsize=256, keep_history=False, random_seed=1)
    assert len(out2.train_history) == 1                                        # test_split_data_train.jl:165-166


def test_train_front_door_with_a_chain_of_dense_layers():
    """train() on a model whose hidden layers are given as a Lux-style Chain with activations of their own (NNModels.jl:145-219): the
    epoch loop on the kernels compiled at run time for it (batch 64: every minibatch is one workgroup's), against the oracle's trajectory
    on the same shuffled batches via the engine's own epoch call"""
    cols = eh.synthetic.make_synth_rbq10(3000, seed=4, nan_frac=0.05)
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=eh.Chain(eh.Dense(16, 16, "relu"), eh.Dense(16, 8, "sigmoid")), activation="tanh", scale_nn_outputs=True)
    assert model.layer_activations == ["tanh", "relu", "sigmoid"]
    out = eh.train(model, cols, nepochs=5, batchsize=64, opt=eh.Adam(0.01), random_seed=1)
    assert len(out.val_history) == 6 and out.val_history[-1]["mse"]["sum"] < 0.8 * out.val_history[0]["mse"]["sum"]
    assert out.ps.size == model.n_theta == (2 * 16 + 16) + (16 * 16 + 16) + (16 * 8 + 8) + (8 + 1) + 1
    # the trained parameters reproduce the reported validation loss through the oracle's forward with the per-layer activations
    spec = ho.HybridSpec(2, [16, 16, 8], "rbq10", dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"], ["reco"], "tanh", True,
                         layer_activations=["tanh", "relu", "sigmoid"])
    yo, yp = out.val_obs_pred["reco"], out.val_obs_pred["reco_pred"]
    ok = ~np.isnan(yo)
    assert float(np.mean((yp[ok] - yo[ok]) ** 2)) == pytest.approx(out.best_loss, rel=1e-4)
    assert spec.n_theta == out.ps.size


# ----------------------------------------------------------------------------------------------
# committed golden fixtures
# ----------------------------------------------------------------------------------------------
def _load_spec(d):
    s = json.loads(str(d["spec"]))
    if s["mech"] not in ho.MECH:                           # a fixture of a user closure: tests/closures.py holds the function
        from tests import closures as cl
        fn, table, forc = cl.CLOSURES[s["mech"]]
        util.register_closure(s["mech"], fn, list(table), forc, s["targets"])
    return ho.HybridSpec(s["n_pred"], s["hidden"], s["mech"], {k: tuple(v) for k, v in s["parameters"].items()}, s["neural"],
                         s["glob"], s["targets"], s["activation"], s["scale_nn_outputs"],
                         nets=[(r, h) for r, h in s["nets"]] if s.get("nets") else None, net_activations=s.get("net_activations"))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_golden_fixture(path):
    d = np.load(path)
    spec = _load_spec(d)
    f = {k[8:]: d[k] for k in d.files if k.startswith("forcing_")}
    y = {k[7:]: d[k] for k in d.files if k.startswith("target_")}
    eng = util.load_engine(spec, d["theta"], d["X"], f, y)
    loss, grad, nv = eng.loss_and_grad()
    assert nv == int(d["n_valid"].sum())
    assert abs(loss - float(d["loss"])) <= TOL * abs(float(d["loss"]))
    assert util.relerr(grad, d["grad"]) <= TOL
    for t in spec.targets:
        assert util.relerr(eng.forward(0, params=False)[t], d["yhat_" + t]) <= TOL
    n, batch = d["X"].shape[1], int(d["batch"])
    batches = [(i, min(batch, n - i)) for i in range(0, n, batch)]
    eng.opt_init("Adam", 0.01)
    eng.train_step(*batches[0])
    scale = max(1.0, float(np.max(np.abs(d["theta_after_1"]))))
    assert np.max(np.abs(eng.get_params() - d["theta_after_1"])) <= 2e-5 * scale
    for b in ((batches * 10)[1:10]):
        eng.train_step(*b)
    assert np.max(np.abs(eng.get_params() - d["theta_after_10"])) <= 2e-4 * scale     # 10 sign-like Adam steps of 1e-2
    eng.close()


# ----------------------------------------------------------------------------------------------
# full BASELINE sizes: size-independent properties
# ----------------------------------------------------------------------------------------------
def test_full_size_gradient_is_count_weighted_sum_of_shards():
    # B = 65 536 (configs[1]): grad(full) * n = sum_k grad(shard_k) * n_k   (linearity of the sums; the DP protocol)
    B = 65536
    spec, theta, X, f, y = util.rbq10_case(B, "tanh", True, 0.05)
    eng = util.load_engine(spec, theta, X, f, y)
    l, g, n = eng.loss_and_grad()
    acc_g, acc_l, acc_n = np.zeros_like(g, dtype=np.float64), 0.0, 0
    for k in range(8):
        lk, gk, nk = eng.loss_and_grad(first=k * B // 8, count=B // 8)
        acc_g += gk.astype(np.float64) * nk; acc_l += lk * nk; acc_n += nk
    assert acc_n == n and acc_l / n == pytest.approx(l, rel=1e-5)
    assert util.relerr(acc_g / n, g) <= 2e-5
    # and against the CPU oracle on a 4096-sample slice of the same resident data
    _check_grad(spec, theta, X, f, y, eng, first=12345, count=4096)
    eng.close()


def test_full_size_variants_and_grids_agree():
    B = 65536
    spec, theta, X, f, y = util.rbq10_case(B, "tanh", True, 0.05)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_option("variant", 0)
    l0, g0, n0 = eng.loss_and_grad()
    for var, mb in ((1, 256), (2, 256), (3, 256), (0, 64), (1, 100)):
        eng.set_option("variant", var); eng.set_option("max_blocks", mb)
        l, g, n = eng.loss_and_grad()
        assert n == n0 and l == pytest.approx(l0, rel=2e-6) and util.relerr(g, g0) <= 5e-6
    eng.set_option("variant", 0); eng.set_option("max_blocks", 256)
    l, g, n = eng.loss_and_grad()
    assert l == l0 and np.array_equal(g, g0)                                   # same launch -> bitwise (deterministic reduction)
    eng.close()


def test_config3_full_batch_262144_runs_and_matches_slice():
    spec = ho.expo2pool_spec((64, 64), "tanh", True)
    X, f, y = ho.make_synth_expo2pool(262144, 11, 0.05)
    theta = ho.init_theta(spec, 2, np.float32)
    eng = util.load_engine(spec, theta, X, f, y)
    l, g, n = eng.loss_and_grad()
    assert n == int((~np.isnan(y["Resp_obs"])).sum()) and np.isfinite(l) and np.isfinite(g).all()
    _check_grad(spec, theta, X, f, y, eng, first=100000, count=2048)
    eng.close()


# ----------------------------------------------------------------------------------------------
# data-parallel seam on one GPU ("virtual shards") and device-resident inputs
# ----------------------------------------------------------------------------------------------
def test_dp_seam_virtual_shards_equal_single_step():
    import torch
    spec, theta, X, f, y = util.rbq10_case(2048, "tanh", True, 0.1)
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01)
    l_ref = ref.train_step(0, 2048)
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01)
    ptr, n = eng.device_buffer(eh._lib.EH_BUF_GRAD)
    buf = torch.as_tensor(eh.dp._DevArray(ptr, n), device="cuda")
    acc = torch.zeros_like(buf)
    for k in range(4):                                    # 4 "ranks", each its quarter; the sum stands in for the all-reduce
        eng.dp_grad(k * 512, 512)
        eng.synchronize()
        acc += buf
    buf.copy_(acc)
    torch.cuda.synchronize()
    l = eng.dp_apply(want_loss=True)
    assert l == pytest.approx(l_ref, rel=1e-5)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 2e-6
    ref.close(); eng.close()


def test_config4_size_eight_virtual_shards_of_65536():
    """BASELINE.json configs[3]: RbQ10, global batch 524 288 = 8 shards x 65 536 (one per GPU there; eight "ranks" on the one GPU
    here, the sum of their raw partial vectors standing in for the RCCL all-reduce).  Five steps; the replica must land where a
    single engine training on the whole 524 288-sample batches lands, and the loss is the mean over the GLOBAL valid count."""
    import torch
    W, b, steps = 8, 65536, 5
    spec, theta, X, f, y = util.rbq10_case(W * b * 2, "tanh", True, 0.05)      # two distinct global batches, visited alternately
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01)
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01)
    ptr, n = eng.device_buffer(eh._lib.EH_BUF_GRAD)
    buf = torch.as_tensor(eh.dp._DevArray(ptr, n), device="cuda")
    for s_ in range(steps):
        g0 = (s_ % 2) * W * b
        l_ref = ref.train_step(g0, W * b)
        acc = torch.zeros_like(buf)
        for k in range(W):
            eng.dp_grad(g0 + k * b, b)
            eng.synchronize()
            acc += buf
        buf.copy_(acc)
        torch.cuda.synchronize()
        l = eng.dp_apply(want_loss=True)
        assert l == pytest.approx(l_ref, rel=2e-6)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 5e-6
    ref.close(); eng.close()


@pytest.mark.parametrize("kinds", ["mse", ("mae", "nseLoss")])
def test_dp_seam_multi_target_virtual_shards(kinds):
    """multi-target models under data parallelism: eh_dp_counts -> all-reduce of the 12 per-target sums -> eh_dp_grad with the
    weights of the GLOBAL batch -> all-reduce -> eh_dp_apply; four "ranks" on the one GPU, sums standing in for the collectives;
    shards with very different numbers of valid targets, so per-shard normalisers would be visibly wrong"""
    import torch
    rng = np.random.default_rng(21)
    B = 2048
    pars = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}
    spec = ho.HybridSpec(4, [16, 8], "fluxpart", pars, ["RUE", "Rb"], ["Q10"], ["NEE", "GPP"], "tanh", True)
    X = rng.standard_normal((4, B)).astype(np.float32)
    f = {"SW_IN": (rng.random(B) * 400).astype(np.float32), "TA": (rng.random(B) * 30).astype(np.float32)}
    y = {"NEE": (5 + rng.standard_normal(B)).astype(np.float32), "GPP": (rng.random(B) * 3).astype(np.float32)}
    y["NEE"][:512][rng.random(512) < 0.8] = np.nan            # shard 0: few NEE values
    y["GPP"][1536:] = np.nan                                   # shard 3: no GPP at all
    theta = ho.init_theta(spec, 6, np.float32)
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01); ref.set_training_loss(kinds)
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01); eng.set_training_loss(kinds)
    gptr, gn = eng.device_buffer(eh._lib.EH_BUF_GRAD); cptr, cn = eng.device_buffer(eh._lib.EH_BUF_TCOUNT)
    gbuf = torch.as_tensor(eh.dp._DevArray(gptr, gn), device="cuda"); cbuf = torch.as_tensor(eh.dp._DevArray(cptr, cn), device="cuda")
    with pytest.raises(RuntimeError, match="eh_dp_counts"):
        eng.dp_grad(0, 512)
    for step in range(3):
        l_ref = ref.train_step(0, B)
        cacc = torch.zeros_like(cbuf)
        for k in range(4):
            eng.dp_counts(k * 512, 512); eng.synchronize(); cacc += cbuf
        gacc = torch.zeros_like(gbuf)
        for k in range(4):
            eng.dp_counts(k * 512, 512); eng.synchronize()      # "rank" k's own call (arms eh_dp_grad) ...
            cbuf.copy_(cacc); torch.cuda.synchronize()          # ... and the all-reduced sums in its buffer
            eng.dp_grad(k * 512, 512); eng.synchronize(); gacc += gbuf
        gbuf.copy_(gacc); torch.cuda.synchronize()
        l = eng.dp_apply(want_loss=True)
        assert l == pytest.approx(l_ref, rel=2e-5), (step, l, l_ref)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 5e-6
    l0, g0, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=kinds)      # and the first step's loss against the oracle
    ref.close(); eng.close()
    chk = util.load_engine(spec, theta, X, f, y); chk.set_training_loss(kinds)
    l1, g1, _ = chk.loss_and_grad()
    assert abs(l1 - l0) <= 1e-5 * abs(l0) and util.relerr(g1, g0) <= 1e-5
    chk.close()


def test_dp_seam_and_front_door_with_per_network_activations_and_depths():
    """the reference's MultiNN constructor case (hidden_layers = (a = [16, 8], d = [8]), activation = (a = tanh, d = sigmoid),
    test/test_generic_hybrid_model.jl:346-347) through the data-parallel seam and through `train`: kernels compiled at run time"""
    import torch
    nets = [([0, 1], [16, 8]), ([2], [8])]
    spec = ho.HybridSpec(3, [1], "rbq10", dict(ho.RBQ10_PARAMS), ["rb", "Q10"], [], ["reco"], "tanh", True, nets=nets, net_activations=["tanh", "sigmoid"])
    rng = np.random.default_rng(12)
    B = 2048
    X = rng.standard_normal((3, B)).astype(np.float32); f = {"ta": rng.uniform(0, 30, B).astype(np.float32)}
    yv = rng.uniform(1, 9, B).astype(np.float32); yv[rng.random(B) < 0.1] = np.nan
    y = {"reco": yv}
    theta = ho.init_theta(spec, 13, np.float32)
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01)
    l_ref = ref.train_step(0, B)
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01)
    ptr, n = eng.device_buffer(eh._lib.EH_BUF_GRAD)
    buf = torch.as_tensor(eh.dp._DevArray(ptr, n), device="cuda")
    acc = torch.zeros_like(buf)
    for k in range(4):
        eng.dp_grad(k * 512, 512); eng.synchronize(); acc += buf
    buf.copy_(acc); torch.cuda.synchronize()
    assert eng.dp_apply(want_loss=True) == pytest.approx(l_ref, rel=1e-5)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 2e-6
    th_ref, _ = ho.train_steps(spec, theta, X, f, y, [(0, B)], dtype=np.float32)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 2e-5
    # epoch driver (eh_train_epoch) on the same kernels, against step-by-step
    a = util.load_engine(spec, theta, X, f, y); a.opt_init("Adam", 0.01)
    la, na = a.train_epoch(256, shuffle=False)
    th_ep, l_ep = ho.train_steps(spec, theta, X, f, y, [(i * 256, 256) for i in range(8)], dtype=np.float32)
    assert na == 8 and la == pytest.approx(float(np.mean(l_ep)), rel=1e-4)
    assert np.max(np.abs(a.get_params() - th_ep)) <= 1e-4
    ref.close(); eng.close(); a.close()
    # the front door: train(model, data) on a table of columns
    cols = {"x0": X[0], "x1": X[1], "x2": X[2], "ta": f["ta"], "reco": yv}
    out = eh.train(util.model_from_spec(spec), cols, nepochs=4, batchsize=256, random_seed=2)
    assert len(out.val_history) == 5 and out.val_history[-1]["mse"]["sum"] < out.val_history[0]["mse"]["sum"]


# ----------------------------------------------------------------------------------------------
# the mechanistic stage on its own (eh_mech_loss_vjp): NN outputs in, d loss / d o out, all on the device
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mech,B,scale", [("rbq10", 4096, True), ("rbq10", 1001, False), ("expo2pool", 2048, True), ("rs_components", 777, True),
                                          ("fluxpart", 4000, True), ("linear", 64, False)])
def test_mech_stage_alone(mech, B, scale):
    import torch
    tabs = {"rbq10": (dict(ho.RBQ10_PARAMS), ["rb"], ["Q10"], None),
            "expo2pool": (dict(ho.EXPO2POOL_PARAMS), ["R0a", "R0b"], ["ka", "kb"], None),
            "rs_components": ({**{f"Rb_{c}": (1.0, 0.0, 5.0) for c in ("het", "root", "myc")}, **{f"Q10_{c}": (2.0, 1.0, 4.0) for c in ("het", "root", "myc")}},
                              ["Rb_het", "Rb_root", "Rb_myc"], ["Q10_het", "Q10_myc"], None),       # Q10_root fixed at its default
            "fluxpart": (dict(FLUX_PARAMS), ["RUE", "Rb"], ["Q10"], ["NEE", "RECO", "GPP"]),
            "linear": ({"alpha": (1.0, -2.0, 3.0), "beta": (0.5, -1.0, 2.0)}, ["alpha", "beta"], [], None)}
    tab, neural, glob, targets = tabs[mech]
    mm = ho.MECH[mech][0]
    targets = targets or [mm.outputs[0]]
    spec = ho.HybridSpec(3, [8], mech, tab, neural, glob, targets, "tanh", scale)
    rng = np.random.default_rng(17)
    K = len(neural)
    o = (rng.standard_normal((K, B)) * (1.0 if scale else 0.3) + (0.0 if scale else 1.5)).astype(np.float32)
    f = {k: rng.uniform(1, 25, B).astype(np.float32) for k in mm.forcings}
    y = {t: rng.uniform(0.5, 6, B).astype(np.float32) for t in targets}
    for v in y.values():
        v[rng.random(B) < 0.15] = np.nan
    theta = ho.init_theta(spec, 18, np.float32)
    theta[spec.n_nn:] += rng.uniform(-0.5, 0.5, len(glob)).astype(np.float32)        # globals away from their defaults
    eng = util.model_from_spec(spec).engine()
    eng.set_params(theta)
    od = torch.from_numpy(o).cuda(); dod = torch.full_like(od, float("nan")); yh = torch.empty((len(targets), B), device="cuda")
    fd = [torch.from_numpy(f[k]).cuda() for k in mm.forcings]; yd = [torch.from_numpy(y[t]).cuda() for t in targets]
    loss, gg, nv = eng.mech_loss_vjp(B, od.data_ptr(), [t.data_ptr() for t in fd], [t.data_ptr() for t in yd], dod.data_ptr(), yh.data_ptr())
    l0, do0, gg0, nv0, yh0 = ho.mech_loss_vjp(spec, theta.astype(np.float64), o, f, y)
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL)
    assert util.relerr(dod.cpu().numpy(), do0) <= TOL
    if glob:
        assert util.relerr(gg, gg0) <= TOL
    for i, t in enumerate(targets):
        assert util.relerr(yh[i].cpu().numpy(), yh0[t]) <= TOL
    # the caller knows the masks' counts (they belong to the data set): same result without the counting pass
    loss2, gg2, nv2 = eng.mech_loss_vjp(B, od.data_ptr(), [t.data_ptr() for t in fd], [t.data_ptr() for t in yd], dod.data_ptr(), n_valid_in=nv0)
    assert loss2 == loss and nv2 == nv and np.array_equal(gg2, gg)
    # a window inside wider planes (ld > count) at an odd offset: the scalar-access kernel
    if B > 100:
        c = B - 37
        dod.fill_(float("nan"))
        l3, _, nv3 = eng.mech_loss_vjp(c, od.data_ptr() + 4 * 5, [t.data_ptr() + 4 * 5 for t in fd], [t.data_ptr() + 4 * 5 for t in yd], dod.data_ptr() + 4 * 5, ld=B)
        l30, do30, _, nv30, _ = ho.mech_loss_vjp(spec, theta.astype(np.float64), o[:, 5:5 + c], {k: v[5:5 + c] for k, v in f.items()}, {k: v[5:5 + c] for k, v in y.items()})
        assert nv3 == sum(nv30) and l3 == pytest.approx(l30, rel=TOL) and util.relerr(dod.cpu().numpy()[:, 5:5 + c], do30) <= TOL
        assert np.isnan(dod.cpu().numpy()[:, :5]).all() and np.isnan(dod.cpu().numpy()[:, 5 + c:]).all()      # nothing outside the window is written
    # mean absolute error as the training loss (loss_fn.jl:64-66); the other losses need batch statistics first: refused
    if len(targets) == 1:                                  # (the engine takes losses other than mse for single-target models)
        eng.set_training_loss("mae")
        l5, g5, nv5 = eng.mech_loss_vjp(B, od.data_ptr(), [t.data_ptr() for t in fd], [t.data_ptr() for t in yd], dod.data_ptr())
        l50, do50, g50, _, _ = ho.mech_loss_vjp(spec, theta.astype(np.float64), o, f, y, kind="mae")
        assert nv5 == nv and l5 == pytest.approx(l50, rel=TOL) and util.relerr(dod.cpu().numpy(), do50) <= TOL
        if glob:
            assert util.relerr(g5, g50) <= TOL
        eng.set_training_loss("rmse")
        with pytest.raises(NotImplementedError):
            eng.mech_loss_vjp(B, od.data_ptr(), [t.data_ptr() for t in fd], [t.data_ptr() for t in yd], dod.data_ptr())
        eng.set_training_loss("mse")
    # all targets missing: skipped batch (epoch.jl:17-19)
    ynan = [torch.full((B,), float("nan"), device="cuda") for _ in targets]
    l4, _, nv4 = eng.mech_loss_vjp(B, od.data_ptr(), [t.data_ptr() for t in fd], [t.data_ptr() for t in ynan], dod.data_ptr())
    assert nv4 == 0 and np.isnan(l4) and not dod.cpu().numpy().any()
    eng.close()


@pytest.mark.parametrize("name,neural,glob,targets,ranges", [
    ("flux_closure", ["alpha", "rref"], ["gmax", "e0"], ["nee", "gpp"], {"sw": (0, 800), "ta": (-5, 30), "vpd": (0, 30)}),
    ("rbq10_closure", ["rb"], ["Q10"], ["reco"], {"ta": (0, 30)}),
])
def test_mech_stage_alone_with_a_recorded_closure(name, neural, glob, targets, ranges):
    """any closure f(; forcing..., params...) (GenericHybridModel.jl:420-425) behind a network that lives outside the library: the
    recorded program runs in the stand-alone stage too (interpreted, one sample per lane)"""
    import torch
    from tests import closures as cl
    fn, table, forc = cl.CLOSURES[name]
    util.register_closure(name, fn, list(table), forc, targets)
    spec = ho.HybridSpec(3, [8], name, dict(table), neural, glob, targets, "tanh", True)
    rng = np.random.default_rng(23)
    B = 1500
    o = rng.standard_normal((len(neural), B)).astype(np.float32)
    f = {k: rng.uniform(lo, hi, B).astype(np.float32) for k, (lo, hi) in ranges.items()}
    truth = ho.mech_loss_vjp(spec, ho.init_theta(spec, 1, np.float64), o * 0.5, f, {t: np.zeros(B) for t in targets})[4]
    y = {}
    for t in targets:
        v = (truth[t] * (1 + 0.1 * rng.standard_normal(B))).astype(np.float32); v[rng.random(B) < 0.1] = np.nan
        y[t] = v
    theta = ho.init_theta(spec, 24, np.float32)
    eng = util.model_from_spec(spec).engine(); eng.set_params(theta)
    od = torch.from_numpy(o).cuda(); dod = torch.empty_like(od); yh = torch.empty((len(targets), B), device="cuda")
    fd = [torch.from_numpy(f[k]).cuda() for k in ho.MECH[name][0].forcings]         # the model's forcing order (eh_set_data order)
    yd = [torch.from_numpy(y[t]).cuda() for t in targets]
    loss, gg, nv = eng.mech_loss_vjp(B, od.data_ptr(), [t.data_ptr() for t in fd], [t.data_ptr() for t in yd], dod.data_ptr(), yh.data_ptr())
    l0, do0, gg0, nv0, yh0 = ho.mech_loss_vjp(spec, theta.astype(np.float64), o, f, y)
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL)
    assert util.relerr(dod.cpu().numpy(), do0) <= TOL and util.relerr(gg, gg0) <= TOL
    for i, t in enumerate(targets):
        assert util.relerr(yh[i].cpu().numpy(), yh0[t]) <= TOL
    eng.close()


def test_mech_stage_trains_an_external_torch_network_like_the_fused_engine():
    """The seam end to end: a torch MLP (autograd) on the GPU + eh_mech_loss_vjp for everything after the network, five plain
    gradient-descent steps, against the fused engine training the same model from the same theta (one kernel does it all there)."""
    import torch
    spec, theta, X, f, y = util.rbq10_case(2048, "tanh", True, 0.1)
    B, lr = 2048, 0.01
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Descent", lr)
    for _ in range(5):
        ref.train_step(0, B, want_loss=False)
    th_ref = ref.get_params(); ref.close()
    (layers,), raw = ho.unpack(spec, theta.astype(np.float64))
    Ws = [torch.tensor(W, dtype=torch.float32, device="cuda", requires_grad=True) for W, _ in layers]
    bs = [torch.tensor(b, dtype=torch.float32, device="cuda", requires_grad=True) for _, b in layers]
    q_raw = np.float32(raw[0])
    xd = torch.from_numpy(X).cuda(); tad = torch.from_numpy(f["ta"]).cuda(); yd = torch.from_numpy(y["reco"]).cuda()
    eng = util.model_from_spec(spec).engine()
    th = theta.copy()
    for _ in range(5):
        eng.set_params(th)                                   # the engine holds the global parameter (Q10); the network lives in torch
        h = xd
        for li, (W, b) in enumerate(zip(Ws, bs)):
            h = W @ h + b[:, None]
            if li < len(Ws) - 1:
                h = torch.tanh(h)
        o = h.contiguous()                                   # (1, B) raw NN output
        d_o = torch.empty_like(o)
        torch.cuda.synchronize()                             # (the engine runs on its own stream)
        loss, gg, nv = eng.mech_loss_vjp(B, o.data_ptr(), [tad.data_ptr()], [yd.data_ptr()], d_o.data_ptr())
        o.backward(d_o)                                      # the network's pullback, by autograd
        with torch.no_grad():
            for p_ in Ws + bs:
                p_ -= lr * p_.grad; p_.grad = None
        q_raw = np.float32(q_raw - np.float32(lr) * gg[0])
        nets = [[(W.detach().cpu().numpy().astype(np.float64), b.detach().cpu().numpy().astype(np.float64)) for W, b in zip(Ws, bs)]]
        th = ho.pack(spec, nets, [np.array([q_raw], np.float64)], np.float64).astype(np.float32)
    # torch.tanh is the exact function, the engine's NN uses NNlib's tanh_fast rational like LuxLib: 1e-6 apart
    assert np.max(np.abs(th - th_ref)) <= 2e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    eng.close()


def test_mech_stage_alone_full_size_properties():
    """B = 4 194 304 (64 batches of BASELINE configs[1]): the loss is the count-weighted mean of the losses of its quarters, and
    d loss / d o of the whole is the quarters' scaled by n_q / n (size-independent properties; no oracle run at this size)"""
    import torch
    spec = ho.rbq10_spec((16, 16), "tanh", True)
    B = 1 << 22
    g = torch.Generator(device="cuda").manual_seed(5)
    od = torch.randn((1, B), device="cuda", generator=g)
    ta = torch.rand(B, device="cuda", generator=g) * 30
    yv = torch.rand(B, device="cuda", generator=g) * 8 + 1
    yv[torch.rand(B, device="cuda", generator=g) < 0.05] = float("nan")
    eng = util.model_from_spec(spec).engine(); eng.set_params(ho.init_theta(spec, 1, np.float32))
    dod = torch.empty_like(od)
    loss, gg, nv = eng.mech_loss_vjp(B, od.data_ptr(), [ta.data_ptr()], [yv.data_ptr()], dod.data_ptr())
    assert nv == int((~torch.isnan(yv)).sum())
    q = B // 4
    acc, gacc = 0.0, 0.0
    dq = torch.empty((1, q), device="cuda")
    for i in range(4):
        lq, gq, nq = eng.mech_loss_vjp(q, od.data_ptr() + 4 * i * q, [ta.data_ptr() + 4 * i * q], [yv.data_ptr() + 4 * i * q], dq.data_ptr())
        acc += lq * nq; gacc += float(gq[0]) * nq
        assert torch.allclose(dod[:, i * q:(i + 1) * q], dq * (nq / nv), rtol=2e-6, atol=1e-12)
    assert loss == pytest.approx(acc / nv, rel=2e-6) and float(gg[0]) == pytest.approx(gacc / nv, rel=2e-5)
    eng.close()


def test_set_data_from_device_pointers():
    import torch
    spec, theta, X, f, y = util.rbq10_case(777, "tanh", True, 0.1)
    eng = util.model_from_spec(spec).engine()
    xd = torch.from_numpy(np.asfortranarray(X).T.copy()).cuda()          # (N, P) row-major == (P x N) column-major
    fd = torch.from_numpy(f["ta"]).cuda(); yd = torch.from_numpy(y["reco"]).cuda()
    eng.set_data_device(0, 777, xd.data_ptr(), [fd.data_ptr()], [yd.data_ptr()])
    eng.set_params(theta)
    _check_grad(spec, theta, X, f, y, eng)
    eng.close()


def test_set_data_layouts_agree():
    """eh_set_data takes the predictors as the reference holds them -- (P x N) column-major = N records of P -- or as P arrays of N
    (EH_DATA_X_PLANES: what a NumPy host holds), from the host or from the device: four ways in, one data set"""
    import ctypes as C
    import torch
    spec, theta, X, f, y = util.rbq10_case(70001, "tanh", True, 0.1)      # (above 65 536: the host path packs in two threads)
    outs = []
    for where, planes in (("host", True), ("host", False), ("device", True), ("device", False), ("rows", None)):
        eng = util.model_from_spec(spec).engine()
        xh = np.ascontiguousarray(X) if planes else np.ascontiguousarray(X.T)
        if where == "rows":          # EH_DATA_X_ROWS: the predictor columns as separate arrays, through the front door (engine.set_data with a list)
            cols = [X[p].copy() for p in range(X.shape[0])]
            eng.set_data(0, cols, [f["ta"]], [y["reco"]])
        elif where == "host":
            fp = (C.c_void_p * 1)(f["ta"].ctypes.data); tp = (C.c_void_p * 1)(y["reco"].ctypes.data)
            eng._chk(eng._lib.eh_set_data(eng._h, 0, 70001, C.c_void_p(xh.ctypes.data), fp, tp, 2 if planes else 0))
            eng.n_samples[0] = 70001
        else:
            xd = torch.from_numpy(xh).cuda(); fd = torch.from_numpy(f["ta"]).cuda(); yd = torch.from_numpy(y["reco"]).cuda()
            eng.set_data_device(0, 70001, xd.data_ptr(), [fd.data_ptr()], [yd.data_ptr()], planes=planes)
        eng.set_params(theta)
        outs.append(eng.loss_and_grad())
        eng.close()
    l0, g0, n0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
    assert outs[0][0] == pytest.approx(l0, rel=TOL) and util.relerr(outs[0][1], g0) <= TOL
    for l, g, n in outs[1:]:
        assert l == outs[0][0] and np.array_equal(g, outs[0][1]) and n == outs[0][2]
    eng = util.model_from_spec(spec).engine()
    with pytest.raises(ValueError, match="flags 8"):
        eng._chk(eng._lib.eh_set_data(eng._h, 0, 10, C.c_void_p(X.ctypes.data), (C.c_void_p * 1)(f["ta"].ctypes.data), (C.c_void_p * 1)(y["reco"].ctypes.data), 8))
    with pytest.raises(ValueError, match="EH_DATA_X_ROWS"):          # rows of pointers are host arrays, and not planes at the same time
        eng._chk(eng._lib.eh_set_data(eng._h, 0, 10, C.c_void_p(X.ctypes.data), (C.c_void_p * 1)(f["ta"].ctypes.data), (C.c_void_p * 1)(y["reco"].ctypes.data), 4 | 2))
    eng.close()


def test_error_paths_on_device():
    spec, theta, X, f, y = util.rbq10_case(64)
    eng = util.model_from_spec(spec).engine()
    with pytest.raises(eh.EngineError):
        eng.loss_and_grad(count=10)                       # no data yet
    with pytest.raises(eh.EngineError):
        eng.train_step(0, 10)                             # no optimiser yet
    with pytest.raises(ValueError):
        eng.set_params(np.zeros(5, np.float32))
    with pytest.raises(NotImplementedError):
        eng.opt_init("Lion")
    eng.close()
    wide = ho.HybridSpec(40, [256, 16], "rbq10", dict(ho.RBQ10_PARAMS), ["rb"], ["Q10"], ["reco"], "tanh", True, input_batchnorm=True)
    with pytest.raises(NotImplementedError, match="no kernel for"):      # no fused kernel is that wide, and the normalisation block of the layer-wise form holds 32 predictors
        util.model_from_spec(wide).engine()


# ----------------------------------------------------------------------------------------------
# fused-update mode (one kernel per step; float-atomic accumulation -> tolerance, not bitwise)
# ----------------------------------------------------------------------------------------------
def test_fused_update_mode_matches_two_kernel_mode():
    spec, theta, X, f, y = util.rbq10_case(4096, "tanh", True, 0.1)
    a = util.load_engine(spec, theta, X, f, y); a.opt_init("Adam", 0.01)
    b = util.load_engine(spec, theta, X, f, y); b.opt_init("Adam", 0.01); b.set_option("fused_update", 1)
    la, na = a.train_epoch(512, shuffle=False)
    lb, nb = b.train_epoch(512, shuffle=False)
    assert na == nb == 8 and lb == pytest.approx(la, rel=1e-5)
    assert np.max(np.abs(a.get_params() - b.get_params())) <= 2e-5
    assert a.train_step(0, 1000) == pytest.approx(b.train_step(0, 1000), rel=1e-5)      # loss of the step itself
    for i in range(5):
        a.train_step(i * 512, 512, want_loss=False); b.train_step(i * 512, 512, want_loss=False)
    assert np.max(np.abs(a.get_params() - b.get_params())) <= 3e-5
    ma, va, bta = a.get_opt_state(); mb, vb, btb = b.get_opt_state()
    assert np.array_equal(bta, btb) and util.relerr(mb, ma) <= 1e-4 and util.relerr(vb, va) <= 1e-4
    # and against the oracle's trajectory from the start
    c = util.load_engine(spec, theta, X, f, y); c.opt_init("Adam", 0.01); c.set_option("fused_update", 1)
    batches = [(i * 256, 256) for i in range(6)]
    for bb in batches:
        c.train_step(*bb, want_loss=False)
    th_ref, _ = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32)
    assert np.max(np.abs(c.get_params() - th_ref)) <= 2e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    lg = c.loss_and_grad()                                     # a non-fused call in between flushes and stays consistent
    l0, g0, _ = ho.loss_and_grad(spec, c.get_params().astype(np.float64), X, f, y)
    assert lg[0] == pytest.approx(l0, rel=1e-5) and util.relerr(lg[1], g0) <= 1e-5
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("aot", [0, 1])
def test_fused_update_trajectory_does_not_depend_on_when_the_host_drains(aot):
    """advisor r05: a step's eight accumulator shards are folded in ONE order (eh_fold8) by the next step's prologue AND by the flush
    kernel behind every drain (synchronize, get_params, a loss read): the same steps with a drain after every one and drained once
    at the end are the same bits.  Minibatches of 2 048 = eight workgroups of the headline kernel, one per shard: every shard is
    non-zero and takes exactly one add, so the float atomics themselves have no order to differ in."""
    spec, theta, X, f, y = util.rbq10_case(8 * 2048, "tanh", True, 0.05)
    out = []
    for drain in (False, True):
        e = util.load_engine(spec, theta, X, f, y); e.set_option("aot_spec", aot)
        e.opt_init("Adam", 0.01); e.set_option("fused_update", 1)
        for i in range(24):
            e.train_step((i % 8) * 2048, 2048, want_loss=False)
            if drain:
                e.synchronize()
                if i % 5 == 0: e.get_params()
        out.append((e.get_params().copy(), [np.asarray(v).copy() for v in e.get_opt_state()]))
        e.close()
    (th0, st0), (th1, st1) = out
    assert np.array_equal(th0, th1)
    assert all(np.array_equal(u, v) for u, v in zip(st0, st1))


def test_fused_update_skips_all_masked_batch_and_reports_losses():
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "rbq10_allmasked_batch.npz"))
    spec = _load_spec(d)
    f = {"ta": d["forcing_ta"]}; y = {"reco": d["target_reco"]}
    eng = util.load_engine(spec, d["theta"], d["X"], f, y); eng.opt_init("Adam", 0.01); eng.set_option("fused_update", 1)
    ref = util.load_engine(spec, d["theta"], d["X"], f, y); ref.opt_init("Adam", 0.01)
    le, _ = eng.train_epoch(64, shuffle=False)                 # batches: valid, ALL MASKED, valid
    lr, _ = ref.train_epoch(64, shuffle=False)
    assert le == pytest.approx(lr, rel=1e-5)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 2e-5
    _, _, bt = eng.get_opt_state()
    assert bt[0] == pytest.approx(np.float32(0.9) ** 3, rel=1e-6)           # two updates only: beta^(2+1)
    eng.set_option("fused_update", 0)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 2e-5
    eng.close(); ref.close()


@pytest.mark.parametrize("kernels", ["generic", "compiled at run time", "specialised ahead of time"])
@pytest.mark.parametrize("case", ["rbq10", "rbq10_bn", "expo2pool", "relu"])
def test_epochs_of_small_minibatches_run_several_steps_per_launch(case, kernels):
    """minibatches that one workgroup covers (the reference's default batch of 64, src/config/TrainingConfig.jl:14) run up to 256
    fused-update steps per kernel launch with the step-to-step state in LDS (EH_MODE_TRAIN_MULTI): against one launch per step and
    against the oracle -- partial last batch, missing targets, input BatchNorm with its running statistics, shuffled epochs with
    evaluation passes in between (the pending update is flushed and picked up again), the optimiser state afterwards"""
    if case == "expo2pool":
        spec = ho.expo2pool_spec((16, 16), "tanh", True)
        rng = np.random.default_rng(3)
        mm = ho.MECH[spec.mech][0]
        N = 1000
        X = rng.standard_normal((spec.n_pred, N)).astype(np.float32)
        f = {k: (10 + 5 * rng.standard_normal(N)).astype(np.float32) for k in mm.forcings}
        y = {k: np.abs(rng.standard_normal(N)).astype(np.float32) for k in spec.targets}
        y[spec.targets[0]][rng.random(N) < 0.1] = np.nan
        theta = ho.init_theta(spec, 2, np.float32)
    elif case == "rbq10_bn":
        spec, theta, X, f, y = _bn_case(1000, seed=5)
    else:
        spec, theta, X, f, y = util.rbq10_case(1000, "relu" if case == "relu" else "tanh", True, 0.1, seed=5)
    B = 64                                                    # 15 full minibatches + one of 40
    engs = []
    if kernels == "specialised ahead of time" and case not in ("rbq10", "rbq10_bn"):
        pytest.skip("no canonical descriptor")                # (csrc/Makefile SPECDEF_*: RbQ10 [2,16,16,1] tanh; input BatchNorm lives in the image, not in the descriptor)
    for multi in (1, 0):
        eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01); eng.set_option("fused_update", 1); eng.set_option("multi_step", multi)
        eng.set_option("aot_spec", int(kernels == "specialised ahead of time")); eng.set_option("specialize", int(kernels == "compiled at run time"))
        if kernels == "compiled at run time":
            eng.loss_and_grad(count=64)                        # (builds the kernel and checks it against the generic one: only then does the multi-step form take over)
        l1, n1 = eng.train_epoch(B, shuffle=False)
        engs.append((eng, l1, n1, eng.get_params()))
    (e1, l1, n1, t1), (e0, l0, n0, t0) = engs
    assert n1 == n0 == 16 and l1 == pytest.approx(l0, rel=2e-6) and np.max(np.abs(t1 - t0)) <= 2e-6
    if kernels != "generic":
        nj, log = e1.jit_status()
        assert nj >= 1 and log.startswith("ahead-of-time") == (kernels == "specialised ahead of time"), (nj, log[:200])
    st = ho.bn_init(spec) if getattr(spec, "input_batchnorm", False) else None
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, [(i * B, min(B, 1000 - i * B)) for i in range(16)], dtype=np.float32, bn_state=st)
    assert l1 == pytest.approx(float(np.nanmean(l_ref)), rel=1e-4) and np.max(np.abs(t1 - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    if st is not None:
        rm, rv = e1.get_bn_state()
        assert util.relerr(rm, st["mean"]) <= 1e-5 and util.relerr(rv, st["var"]) <= 1e-5
    for ep in range(3):                                       # shuffled epochs, an evaluation pass between them
        for e in (e1, e0):
            e.train_epoch(B, seed=40 + ep, shuffle=True, want_loss=False)
        m1, _ = e1.eval(0); m0, _ = e0.eval(0)
        assert m1[0]["mse"] == pytest.approx(m0[0]["mse"], rel=2e-5)
    assert np.max(np.abs(e1.get_params() - e0.get_params())) <= 2e-5
    (m1_, v1_, b1_), (m0_, v0_, b0_) = e1.get_opt_state(), e0.get_opt_state()
    assert np.max(np.abs(m1_ - m0_)) <= 1e-5 * max(1e-3, float(np.max(np.abs(m0_)))) + 1e-9 and np.allclose(b1_, b0_, rtol=1e-6)
    e1.set_option("fused_update", 0)                          # leaving the mode: the deterministic path continues from the same state
    l_a = e1.train_step(0, 64); l_b = e0.train_step(0, 64)
    assert l_a == pytest.approx(l_b, rel=2e-5)
    e1.close(); e0.close()


def test_data_parallel_driver_world_size_one_nccl():
    # the real DataParallel driver (RCCL through torch.distributed) with a single rank: both the
    # fused one-kernel path and the deterministic three-kernel path must reproduce plain training
    import socket
    import torch
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        spec, theta, X, f, y = util.rbq10_case(4096, "tanh", True, 0.1)
        ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01)
        for i in range(8):
            ref.train_step(i * 512, 512, want_loss=False)
        for fused in (True, False):
            eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01)
            drv = eh.dp.DataParallel(eng, fused=fused)
            for i in range(8):
                drv.step(i * 512, 512)
            eng.synchronize()
            torch.cuda.synchronize()
            assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 3e-5, fused
            eng.close()
        ref.close()
        # a model without a fused-update kernel (hidden width 128): the driver drops to the three-kernel path by itself
        spec, theta, X, f, y = _rs6_case(32, (128, 128), 2048)
        ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.002)
        eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.002)
        drv = eh.dp.DataParallel(eng)
        assert not drv.fused and not drv.p2p
        for i in range(4):
            ref.train_step(i * 512, 512, want_loss=False)
            drv.step(i * 512, 512)
        eng.synchronize(); torch.cuda.synchronize()
        assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 3e-5
        eng.close(); ref.close()
    finally:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# other training losses (loss_fn.jl:58-86)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["rmse", "mae", "nseLoss"])
@pytest.mark.parametrize("fused", [0, 1])
def test_other_training_losses(kind, fused):
    spec, theta, X, f, y = util.rbq10_case(1500, "tanh", True, 0.1)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_training_loss(kind)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=kind)
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    eng.opt_init("Adam", 0.01)
    eng.set_option("fused_update", fused)
    batches = [(i * 300, 300) for i in range(5)]
    losses = [eng.train_step(*b) for b in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32, kind=kind)
    assert np.allclose(losses, l_ref, rtol=1e-4)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    eng.close()


def test_train_with_nse_loss_front_door():
    cols = eh.synthetic.make_synth_rbq10(3000, seed=5, nan_frac=0.05)
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    out = eh.train(model, cols, nepochs=5, batchsize=200, training_loss="nseLoss", loss_types=["nse", "mse"], random_seed=2)
    assert out.val_history[-1]["nse"]["sum"] > out.val_history[0]["nse"]["sum"]         # maximised metric (loss_fn.jl:181-187)


# ----------------------------------------------------------------------------------------------
# input BatchNorm (constructHybridModel(...; input_batchnorm = true), NNModels.jl:89-105,226)
# ----------------------------------------------------------------------------------------------
def _bn_case(B, P_raw=True, seed=11):
    spec = ho.rbq10_spec((16, 16), "tanh", True)
    spec.input_batchnorm = True
    X, f, y = ho.make_synth_rbq10(B, seed, 0.1)          # raw predictors (sw_pot ~ 50 +- 20): exactly why the README turns BN on
    return spec, ho.init_theta(spec, 3, np.float32), X, f, y


def test_input_batchnorm_train_mode_gradient_and_test_mode_forward():
    spec, theta, X, f, y = _bn_case(1500)
    eng = util.load_engine(spec, theta, X, f, y)
    loss, grad, nv = eng.loss_and_grad()                                    # train mode: statistics of this batch
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, bn_state=ho.bn_init(spec))
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    _check = eng.loss_and_grad(first=200, count=333)                        # other window -> other statistics
    sl = slice(200, 533)
    l1, g1, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X[:, sl], {"ta": f["ta"][sl]}, {"reco": y["reco"][sl]})
    assert _check[0] == pytest.approx(l1, rel=TOL) and util.relerr(_check[1], g1) <= TOL
    st = ho.bn_init(spec)                                                   # test mode before any step: running mean 0, var 1
    ref = ho.forward(spec, theta.astype(np.float64), X, f, bn_state=st, train_mode=False)
    assert util.relerr(eng.forward(0, params=False)["reco"], ref["reco"]) <= TOL
    rm, rv = eng.get_bn_state()
    assert not rm.any() and np.all(rv == 1)                                 # loss_and_grad does not touch the state
    eng.close()


@pytest.mark.parametrize("fused", [0, 1])
def test_input_batchnorm_training_trajectory_and_running_statistics(fused):
    spec, theta, X, f, y = _bn_case(2048)
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01); eng.set_option("fused_update", fused)
    batches = [(i * 256, 256) for i in range(8)]
    losses = [eng.train_step(*b) for b in batches]
    st = ho.bn_init(spec)
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32, bn_state=st)
    assert np.allclose(losses, l_ref, rtol=1e-4)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    rm, rv = eng.get_bn_state()
    assert util.relerr(rm, st["mean"]) <= 1e-5 and util.relerr(rv, st["var"]) <= 1e-5
    m, _ = eng.eval(0)                                                      # eval = test mode with the running statistics
    ref = ho.evaluate(spec, eng.get_params().astype(np.float64), X, f, y, bn_state=st)[0]
    assert m[0]["mse"] == pytest.approx(ref["mse"]["reco"], rel=3e-5)
    eng.set_bn_state(np.array([50.0, 0.0], np.float32), np.array([400.0, 700.0], np.float32))
    ref2 = ho.forward(spec, eng.get_params().astype(np.float64), X, f, bn_state={"mean": np.array([50.0, 0.0]), "var": np.array([400.0, 700.0])}, train_mode=False)
    assert util.relerr(eng.forward(0, params=False)["reco"], ref2["reco"]) <= TOL
    eng.close()


@pytest.mark.parametrize("B", [64, 257, 512, 513])
@pytest.mark.parametrize("fused", [0, 1])
def test_input_batchnorm_statistics_inside_the_step_kernel(B, fused):
    """minibatches of up to 512 samples (the reference's tutorial trains on 64: docs/literate/tutorials/synthetic_respiration_gpu.jl:79-104)
    take their BatchNorm statistics inside the per-wave step kernel, larger ones from a launch in front of it: both against the oracle, on
    a window off the start, on gathered indices, through a training trajectory with the running statistics, and against each other"""
    spec, theta, X, f, y = _bn_case(2100, seed=23)
    th64 = theta.astype(np.float64)
    res = []
    for in_kernel in (1, 0):
        eng = util.load_engine(spec, theta, X, f, y); eng.set_option("bn_in_kernel", in_kernel)
        sl = slice(77, 77 + B)
        l, g, nv = eng.loss_and_grad(first=77, count=B)
        l0, g0, _ = ho.loss_and_grad(spec, th64, X[:, sl], {"ta": f["ta"][sl]}, {"reco": y["reco"][sl]})
        assert l == pytest.approx(l0, rel=TOL) and util.relerr(g, g0) <= TOL
        eng.opt_init("Adam", 0.01); eng.set_option("fused_update", fused)
        rng = np.random.default_rng(5)
        ix = [rng.choice(2100, B, replace=False).astype(np.int32) for _ in range(4)]
        losses = [eng.train_step(0, B, idx=i) for i in ix]
        cat = np.concatenate(ix)                                         # the oracle steps over the same gathered minibatches, laid end to end
        st = ho.bn_init(spec)
        th_ref, l_ref = ho.train_steps(spec, theta, X[:, cat], {"ta": f["ta"][cat]}, {"reco": y["reco"][cat]}, [(k * B, B) for k in range(4)],
                                       dtype=np.float32, bn_state=st)
        assert np.allclose(losses, l_ref, rtol=1e-4)
        assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
        rm, rv = eng.get_bn_state()
        assert util.relerr(rm, st["mean"]) <= 1e-5 and util.relerr(rv, st["var"]) <= 1e-5
        eng.synchronize()
        res.append((l, g, np.array(losses), eng.get_params(), eng.get_bn_state()))
        eng.close()
    (l1, g1, ls1, t1, (rm1, rv1)), (l0_, g0_, ls0, t0, (rm0, rv0)) = res
    assert abs(l1 - l0_) <= 2e-6 * abs(l0_) and util.relerr(g1, g0_) <= 2e-6
    assert np.allclose(ls1, ls0, rtol=2e-5) and np.max(np.abs(t1 - t0)) <= 2e-5
    assert util.relerr(rm1, rm0) <= 1e-6 and util.relerr(rv1, rv0) <= 1e-6


def test_readme_quickstart_configuration_trains():
    # README.md:185-201: hidden [16,16], sigmoid/tanh, scale_nn_outputs = true, input_batchnorm = true on RAW predictors
    cols = eh.synthetic.make_synth_rbq10(4000, seed=8, nan_frac=0.05)
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="sigmoid", scale_nn_outputs=True, input_batchnorm=True)
    out = eh.train(model, cols, nepochs=8, batchsize=128, opt=eh.AdamW(0.01, (0.9, 0.999), 0.01), random_seed=3)
    assert out.val_history[-1]["mse"]["sum"] < 0.5 * out.val_history[0]["mse"]["sum"]
    assert out.st["st_nn"]["running_mean"][0] == pytest.approx(50.0, rel=0.1)


# ----------------------------------------------------------------------------------------------
# MultiNNHybridModel: one single-output MLP per neural parameter (GenericHybridModel.jl:142-206,458-530)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nets,glob", [
    ([([0, 1], [16, 16]), ([1, 2], [8, 8])], []),                       # rb and Q10 both neural, overlapping predictor sets
    ([([0, 1, 2], [16, 16])], ["Q10"]),                                 # a single network (ExpoHybridEstim.jl:34 style) + a global
    ([([0], [24, 12]), ([1, 2, 3], [30, 20])], []),                     # unequal widths, 54 units side by side -> the 64-wide kernel
])
def test_multinn_hybrid_model(nets, glob):
    neural = ["rb", "Q10"][: len(nets)]
    spec = ho.HybridSpec(4, [1], "rbq10", dict(ho.RBQ10_PARAMS), neural, glob, ["reco"], "tanh", True, nets=nets)
    rng = np.random.default_rng(5)
    B = 700
    X = rng.standard_normal((4, B)).astype(np.float32)
    f = {"ta": rng.uniform(0, 30, B).astype(np.float32)}
    yv = rng.uniform(1, 9, B).astype(np.float32); yv[rng.random(B) < 0.1] = np.nan
    theta = ho.init_theta(spec, 6, np.float32)
    eng = util.load_engine(spec, theta, X, f, {"reco": yv})
    assert eng.n_theta == spec.n_theta
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, {"reco": yv})
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    eng.opt_init("Adam", 0.01)
    batches = [(i * 100, 100) for i in range(6)]
    for b in batches:
        eng.train_step(*b, want_loss=False)
    th_ref, _ = ho.train_steps(spec, theta, X, f, {"reco": yv}, batches, dtype=np.float32)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    ref = ho.forward(spec, eng.get_params().astype(np.float64), X, f)
    out = eng.forward(0)
    assert util.relerr(out["reco"], ref["reco"]) <= TOL and util.relerr(out["parameters"]["rb"], ref["parameters"]["rb"]) <= TOL
    eng.close()


@pytest.mark.parametrize("hidden,acts,bn", [
    ([16, 16], ["tanh", "relu"], False),                       # per-wave kernel, the headline's shape
    ([16, 32, 8], ["tanh", "relu", "sigmoid"], True),          # three layers of different widths, input BatchNorm in front
    ([24], ["swish"], False),                                  # one hidden layer: the Chain degenerates to the model's activation
    ([64, 48], ["sigmoid", "swish"], False),                   # 64-wide kernel; swish keeps its pre-activation
    ([96, 120], ["relu", "tanh"], False),                      # the row-split kernel
    ([200, 64, 32], ["tanh", "swish", "relu"], False),         # wider than any fused kernel: the layer-wise form
    ([32, 32, 16, 8], ["sigmoid", "tanh", "identity", "relu"], False),      # four hidden layers: layer-wise
])
def test_single_network_with_an_activation_per_layer(hidden, acts, bn):
    """`hidden_layers::Chain` of Dense layers with activations of their own (NNModels.jl:145-219): the reference puts
    Dense(in, first_h, activation) in front and Dense(last_h, out) behind them.  The fused kernels that run it are compiled at run
    time around eh_row_act(layer, row) (descriptor: EH_ACT_PER_NET with n_nets = 0, net_activation[l] = layer l's); the layer-wise
    form takes the layer's activation per product.  Loss, gradient, forward, metrics and an Adam trajectory against the oracle."""
    spec = ho.HybridSpec(3, list(hidden), "rbq10", dict(ho.RBQ10_PARAMS), ["rb"], ["Q10"], ["reco"], acts[0], True, layer_activations=list(acts))
    spec.input_batchnorm = bn
    rng = np.random.default_rng(12)
    B = 1300
    X = rng.standard_normal((3, B)).astype(np.float32)
    f = {"ta": rng.uniform(0, 30, B).astype(np.float32)}
    yv = rng.uniform(1, 9, B).astype(np.float32); yv[rng.random(B) < 0.1] = np.nan
    theta = ho.init_theta(spec, 8, np.float32)
    model = util.model_from_spec(spec)
    assert model.layer_activations == (list(acts) if len(set(acts)) > 1 else None)
    eng = util.load_engine(spec, theta, X, f, {"reco": yv})
    njit, jlog = eng.jit_status()
    fused = len(hidden) <= 3 and max(hidden) <= 128 and not (len(hidden) == 3 and max(hidden) > 64)
    if fused and len(set(acts)) > 1:
        assert njit >= 1, jlog                   # no kernel built ahead of time can run this model
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, {"reco": yv})
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    if not bn:                                   # (with BatchNorm the forward of a fresh engine is in test mode on initial running statistics)
        ref = ho.forward(spec, theta.astype(np.float64), X, f)
        out = eng.forward(0)
        assert util.relerr(out["reco"], ref["reco"]) <= TOL and util.relerr(out["parameters"]["rb"], ref["parameters"]["rb"]) <= TOL
        m, _ = eng.eval(0)
        yy = yv.astype(np.float64)
        assert m[0]["mse"] == pytest.approx(ho.loss_fn(ref["reco"], yy, ~np.isnan(yy), "mse"), rel=3e-5)
        eng.opt_init("Adam", 0.01)
        batches = [(i * 160, 160) for i in range(7)]
        losses = [eng.train_step(*b) for b in batches]
        th_ref, l_ref = ho.train_steps(spec, theta, X, f, {"reco": yv}, batches, dtype=np.float32)
        assert np.allclose(losses, l_ref, rtol=1e-4)
        assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
        if fused and max(hidden) <= 64:
            eng.set_option("fused_update", 1)    # one kernel per step: the same compiled kernel, update in its prologue
            more = [(i * 100, 100) for i in range(3)]
            for b in more:
                eng.train_step(*b, want_loss=False)
            th_ref, _ = ho.train_steps(spec, theta, X, f, {"reco": yv}, batches + more, dtype=np.float32)
            assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    eng.close()


@pytest.mark.parametrize("nets,acts,glob", [
    ([([0, 1], [16, 16]), ([1, 2], [8, 8])], ["swish", "tanh"], []),           # per-wave kernel, 24 units side by side
    ([([0], [24, 12]), ([1, 2, 3], [30, 20])], ["relu", "sigmoid"], []),        # a net boundary inside a 16-row block; 64-wide kernel
    ([([0, 1], [8]), ([2], [5])], ["tanh", "identity"], []),                    # one hidden layer
    ([([0, 1, 2], [40, 60]), ([1, 3], [50, 60])], ["sigmoid", "swish"], []),    # 90 / 120 units: the row-split kernel
    ([([0, 1], [16, 16]), ([2, 3], [16, 16]), ([0, 3], [16, 16])], ["tanh", "relu", "swish"], ["Q10_het", "Q10_root", "Q10_myc"]),
    # nets of different depth: hidden_layers = (a = [16, 8], d = [8]), activation = (a = tanh, d = sigmoid) is the reference's own
    # constructor test (test/test_generic_hybrid_model.jl:346-347); the shallower net rides identity blocks to the output layer
    ([([0, 1], [16, 8]), ([2], [8])], ["tanh", "sigmoid"], []),
    ([([0, 1], [12]), ([1, 2, 3], [20, 10, 6])], None, []),                     # one activation for all, depths 1 and 3
    ([([0, 1, 2], [70]), ([3], [50, 40])], ["relu", "tanh"], []),               # envelope 120 / 110 wide: the row-split kernel
    ([([0], [8, 8]), ([1, 2], [16]), ([3], [4, 12])], ["swish", "tanh", "relu"], ["Q10_het", "Q10_root", "Q10_myc"]),
])
def test_multinn_per_network_activations(nets, acts, glob):
    """activation::NamedTuple of the MultiNN constructor (GenericHybridModel.jl:168-176): net k runs activation[k].  On the device
    the nets are one block-diagonal MLP whose activation depends on the row; that kernel is compiled at run time."""
    if len(nets) == 3:
        tab = {"Rb_het": (1.0, 0.0, 5.0), "Rb_root": (1.0, 0.0, 5.0), "Rb_myc": (0.5, 0.0, 3.0),
               "Q10_het": (2.0, 1.0, 4.0), "Q10_root": (2.0, 1.0, 4.0), "Q10_myc": (2.0, 1.0, 4.0)}
        mech, neural, tname = "rs_components", ["Rb_het", "Rb_root", "Rb_myc"], "R_soil"
    else:
        tab, mech, neural, tname = dict(ho.RBQ10_PARAMS), "rbq10", ["rb", "Q10"], "reco"
    spec = ho.HybridSpec(4, [1], mech, tab, neural, glob, [tname], "swish", True, nets=nets, net_activations=acts)
    rng = np.random.default_rng(9)
    B = 1100
    X = rng.standard_normal((4, B)).astype(np.float32)
    f = {"ta": rng.uniform(0, 30, B).astype(np.float32)}
    yv = rng.uniform(1, 9, B).astype(np.float32); yv[rng.random(B) < 0.1] = np.nan
    theta = ho.init_theta(spec, 8, np.float32)
    eng = util.load_engine(spec, theta, X, f, {tname: yv})
    njit, jlog = eng.jit_status()
    assert njit >= 1, jlog                       # no kernel built ahead of time can run this model
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, {tname: yv})
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    ref = ho.forward(spec, theta.astype(np.float64), X, f)
    out = eng.forward(0)
    assert util.relerr(out[tname], ref[tname]) <= TOL
    for n in neural:
        assert util.relerr(out["parameters"][n], ref["parameters"][n]) <= TOL
    m, _ = eng.eval(0)
    yy = yv.astype(np.float64)
    assert m[0]["mse"] == pytest.approx(ho.loss_fn(ref[tname], yy, ~np.isnan(yy), "mse"), rel=3e-5)
    eng.opt_init("Adam", 0.01)
    batches = [(i * 150, 150) for i in range(7)]
    losses = [eng.train_step(*b) for b in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, {tname: yv}, batches, dtype=np.float32)
    assert np.allclose(losses, l_ref, rtol=1e-4)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    depth = max(len(h) for _, h in nets)
    if max(sum(h[min(l, len(h) - 1)] for _, h in nets) for l in range(depth)) <= 64:
        eng.set_option("fused_update", 1)        # one kernel per step: the same compiled kernel, update in its prologue
        more = [(i * 100, 100) for i in range(3)]
        for b in more:
            eng.train_step(*b, want_loss=False)
        th_ref, _ = ho.train_steps(spec, theta, X, f, {tname: yv}, batches + more, dtype=np.float32)
        assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
        with pytest.raises(NotImplementedError):
            eng.p2p_init(1, 0)                   # no cross-GPU variant of these kernels: the all-reduce seam instead
    eng.close()


# ----------------------------------------------------------------------------------------------
# multi-target losses through the multi-output FluxPart model (compute_loss.jl:50-53,115-126: L = sum_t mean_t)
# ----------------------------------------------------------------------------------------------
FLUX_PARAMS = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}


def _flux_data(B, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((4, B)).astype(np.float32)
    f = {"SW_IN": rng.uniform(0, 800, B).astype(np.float32), "TA": rng.uniform(0, 30, B).astype(np.float32)}
    nee = rng.normal(-3, 4, B).astype(np.float32); reco = rng.uniform(1, 8, B).astype(np.float32); gpp = rng.uniform(0, 12, B).astype(np.float32)
    nee[rng.random(B) < 0.2] = np.nan; reco[rng.random(B) < 0.35] = np.nan          # every target has its own mask and n_t
    return X, f, {"NEE": nee, "RECO": reco, "GPP": gpp}


@pytest.mark.parametrize("targets,nets", [
    (["NEE"], None),
    (["NEE", "RECO"], [([0, 1], [16, 16]), ([2, 3], [16, 16])]),      # the reference's layout: an RUE net and an Rb net (FluxPartModel_Q10_Lux.jl)
    (["RECO", "GPP", "NEE"], None),
])
def test_fluxpart_multi_target(targets, nets):
    B = 900
    X, f, y = _flux_data(B)
    spec = ho.HybridSpec(4, [16, 16], "fluxpart", dict(FLUX_PARAMS), ["RUE", "Rb"], ["Q10"], targets, "tanh", True, nets=nets)
    theta = ho.init_theta(spec, 2, np.float32)
    yt = {t: y[t] for t in targets}
    eng = util.load_engine(spec, theta, X, f, yt)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, yt)
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    out = eng.forward(0, params=False)
    ref = ho.forward(spec, theta.astype(np.float64), X, f)
    for t in targets:
        assert util.relerr(out[t], ref[t]) <= TOL
    m, _ = eng.eval(0)
    for i, t in enumerate(targets):
        yy = yt[t].astype(np.float64)
        assert m[i]["mse"] == pytest.approx(ho.loss_fn(ref[t], yy, ~np.isnan(yy), "mse"), rel=3e-5)
    eng.opt_init("Adam", 0.01)
    batches = [(i * 150, 150) for i in range(6)]
    losses = [eng.train_step(*b) for b in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, yt, batches, dtype=np.float32)
    assert np.allclose(losses, l_ref, rtol=1e-4)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    if len(targets) > 1:
        # one kernel per step on a multi-target model (since round 2: counting pre-pass + fused step), same trajectory
        eng.set_params(theta); eng.opt_init("Adam", 0.01)
        eng.set_option("fused_update", 1)
        lf = [eng.train_step(*b) for b in batches]
        assert np.allclose(lf, l_ref, rtol=1e-4)
        assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
        eng.set_option("fused_update", 0)
        yt2 = dict(yt); yt2[targets[0]] = np.full(B, np.nan, np.float32)          # one target entirely missing: it contributes 0
        eng2 = util.load_engine(spec, theta, X, f, yt2)
        l2, g2, n2 = eng2.loss_and_grad()
        l20, g20, n20 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, yt2)
        assert n2 == sum(n20) and l2 == pytest.approx(l20, rel=TOL) and util.relerr(g2, g20) <= TOL
        eng2.close()
    eng.close()


# ----------------------------------------------------------------------------------------------
# hidden widths 65..128: the row-split kernel (eh_wide.hpp); BASELINE.json configs[4] = [32,128,128,6]
# ----------------------------------------------------------------------------------------------
def _rs6_case(P, hidden, B, act="tanh", seed=11, nan_frac=0.1):
    comps = ("het", "root", "myc")
    tab = {**{f"Rb_{c}": (1.0, 0.0, 5.0) for c in comps}, **{f"Q10_{c}": (2.0 + 0.3 * i, 1.0, 4.0) for i, c in enumerate(comps)}}
    mm = ho.MECH["rs_components"][0]
    spec = ho.HybridSpec(P, list(hidden), "rs_components", tab, list(tab), [], [mm.outputs[0]], act, True)
    rng = np.random.default_rng(seed)
    X = (rng.standard_normal((P, B)) * 0.5).astype(np.float32)
    frc = {mm.forcings[0]: rng.uniform(-5, 25, B).astype(np.float32)}
    yv = rng.uniform(0.5, 6, B).astype(np.float32)
    yv[rng.random(B) < nan_frac] = np.nan
    return spec, ho.init_theta(spec, seed, np.float32), X, frc, {mm.outputs[0]: yv}


@pytest.mark.parametrize("hidden", [(128,), (128, 128), (100, 72), (65, 128), (16, 80)])
@pytest.mark.parametrize("act", ["tanh", "swish", "relu"])
def test_width_128_shapes(hidden, act):
    _check_grad(*util.rbq10_case(300, act, True, 0.1, hidden=hidden))


@pytest.mark.parametrize("B", [1, 31, 32, 33, 1000])
def test_config5_shape_32_128_128_6(B):
    _check_grad(*_rs6_case(32, (128, 128), B))


def test_width_128_twenty_predictors_unscaled_sigmoid():
    # raw (unscaled) network outputs feed the mechanistic model: LinearHM tolerates any sign
    mm = ho.MECH["linear"][0]
    spec = ho.HybridSpec(20, [96, 128], "linear", {"alpha": (1.0, -2.0, 3.0), "beta": (0.5, -1.0, 2.0)}, ["alpha", "beta"], [], [mm.outputs[0]], "sigmoid", False)
    rng = np.random.default_rng(8)
    B = 500
    X = (rng.standard_normal((20, B)) * 0.5).astype(np.float32)
    yv = rng.uniform(-2, 4, B).astype(np.float32)
    yv[rng.random(B) < 0.1] = np.nan
    _check_grad(spec, ho.init_theta(spec, 8, np.float32), X, {mm.forcings[0]: rng.uniform(-1, 2, B).astype(np.float32)}, {mm.outputs[0]: yv})


def test_width_128_forward_eval_and_adam_trajectory():
    spec, theta, X, f, y = _rs6_case(32, (128, 128), 2000)
    eng = util.load_engine(spec, theta, X, f, y)
    out = eng.forward(0)
    ref = ho.forward(spec, theta.astype(np.float64), X, f)
    assert util.relerr(out["R_soil"], ref["R_soil"]) <= TOL
    for name in spec.parameters:
        assert util.relerr(out["parameters"][name], np.broadcast_to(ref["parameters"][name], (2000,))) <= TOL
    metrics, _ = eng.eval(0)
    yy = y["R_soil"].astype(np.float64)
    mask = ~np.isnan(yy)
    assert metrics[0]["n"] == mask.sum()
    for k in ("mse", "mae", "r2", "nse"):
        assert metrics[0][k] == pytest.approx(ho.loss_fn(ref["R_soil"], yy, mask, k), rel=2e-5, abs=2e-6), k
    eng.opt_init("Adam", 0.001)
    batches = [(i * 250, 250) for i in range(8)]
    losses = [eng.train_step(a, b) for a, b in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, lr=0.001, dtype=np.float32)
    assert np.allclose(losses, l_ref, rtol=1e-4)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 2e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    with pytest.raises(NotImplementedError, match="fused_update"):
        eng.set_option("fused_update", 1)
    eng.close()


@pytest.mark.parametrize("kinds", ["mse", ("mse", "mae"), ("nseLoss", "mse")])
def test_fused_update_mode_on_a_multi_target_model(kinds):
    """one kernel per step for T > 1: the counting pre-pass supplies the per-target weights, the next step's prologue applies the update"""
    rng = np.random.default_rng(11)
    B = 1200
    pars = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}
    spec = ho.HybridSpec(5, [16, 16], "fluxpart", pars, ["RUE", "Rb"], ["Q10"], ["NEE", "GPP"], "tanh", True)
    X = rng.standard_normal((5, B)).astype(np.float32)
    f = {"SW_IN": (rng.random(B) * 400).astype(np.float32), "TA": (rng.random(B) * 30).astype(np.float32)}
    y = {"NEE": rng.standard_normal(B).astype(np.float32), "GPP": (rng.random(B) * 3).astype(np.float32)}
    y["NEE"][rng.random(B) < 0.25] = np.nan; y["GPP"][rng.random(B) < 0.1] = np.nan
    y["NEE"][600:800] = np.nan; y["GPP"][600:800] = np.nan                  # one all-masked batch: skipped (epoch.jl:17-19)
    theta = ho.init_theta(spec, 4, np.float32)
    batches = [(i * 200, 200) for i in range(6)]
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01)
    eng.set_option("fused_update", 1)
    eng.set_training_loss(kinds)
    losses = [eng.train_step(a, n) for a, n in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32, kind=kinds)
    ok = ~np.isnan(l_ref)
    assert np.array_equal(np.isnan(losses), ~ok) and np.allclose(np.asarray(losses)[ok], np.asarray(l_ref)[ok], rtol=5e-5)
    d = np.abs(eng.get_params() - th_ref)
    assert np.mean(d <= 3e-5) >= 0.995, float(d.max())
    # and through an epoch of the shuffled driver, against the two-kernel mode
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01); ref.set_training_loss(kinds)
    eng.set_params(theta); eng.opt_init("Adam", 0.01)
    for e_ in (eng, ref):
        e_.train_epoch(128, seed=5, shuffle=True, want_loss=False)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 2e-5
    eng.close(); ref.close()


def test_width_128_three_layers_runs_layer_by_layer():
    # (refused in round 1: no fused kernel holds three 128-wide layers; since round 2 such shapes take the layer-wise form, tests/test_gpu_lform.py)
    _check_grad(*util.rbq10_case(400, "tanh", True, 0.1, hidden=(128, 128, 128)))
    _check_grad(*util.rbq10_case(400, "swish", True, 0.1, hidden=(128, 128, 128)))      # (swish there since round 3: the pre-activations are kept)


def test_width_128_grid_independence_full_batch():
    # 65 536 samples: 2048 tiles over 256 workgroups vs. a single workgroup walking all of them
    spec, theta, X, f, y = _rs6_case(32, (128, 128), 65536, seed=5)
    eng = util.load_engine(spec, theta, X, f, y)
    l1, g1, n1 = eng.loss_and_grad()
    eng.set_option("max_blocks", 7)
    l2, g2, n2 = eng.loss_and_grad()
    assert n1 == n2 and abs(l1 - l2) <= 1e-5 * abs(l1) and util.relerr(g1, g2) <= 1e-5
    sl = slice(0, 4096)
    l0, g0, _ = ho.loss_and_grad(spec, np.asarray(theta, np.float64), X[:, sl], {k: v[sl] for k, v in f.items()}, {k: v[sl] for k, v in y.items()})
    l3, g3, _ = eng.loss_and_grad(first=0, count=4096)
    assert abs(l3 - l0) <= TOL * abs(l0) and util.relerr(g3, g0) <= TOL
    eng.close()


@pytest.mark.parametrize("hidden", [(64, 64), (64,), (64, 48, 64)])
def test_width_64_with_32_predictors_falls_back_to_row_split_kernel(hidden):
    # P > 16 with full 64-wide layers: the per-wave kernel has no LDS room to park one gradient per wave
    _check_grad(*_rs6_case(32, hidden, 700))


@pytest.mark.parametrize("hidden", [(64, 64), (40,), (64, 33, 64)])
def test_row_split_option_agrees_with_per_wave_kernel(hidden):
    spec = ho.expo2pool_spec(hidden, "tanh", True)
    X, f, y = ho.make_synth_expo2pool(3000, 5, 0.1)
    theta = ho.init_theta(spec, 2, np.float32)
    eng = util.load_engine(spec, theta, X, f, y)
    l1, g1, n1 = eng.loss_and_grad()
    eng.set_option("row_split", 1)
    _check_grad(spec, theta, X, f, y, eng=eng)
    l2, g2, n2 = eng.loss_and_grad()
    assert n1 == n2 and abs(l1 - l2) <= 2e-6 * abs(l1) and util.relerr(g1, g2) <= 2e-6
    ya = eng.forward(0)["Resp_obs"]
    eng.set_option("row_split", 0)
    assert util.relerr(eng.forward(0)["Resp_obs"], ya) <= 2e-6
    l3, g3, _ = eng.loss_and_grad()
    assert l3 == l1 and np.array_equal(g3, g1)
    eng.close()
    narrow = util.load_engine(*util.rbq10_case(64))
    with pytest.raises(NotImplementedError, match="row_split"):
        narrow.set_option("row_split", 1)
    narrow.close()


# ----------------------------------------------------------------------------------------------
# input BatchNorm under data parallelism: global batch statistics through a second all-reduce
# ----------------------------------------------------------------------------------------------
def test_sync_batchnorm_virtual_shards_equal_single_step():
    import torch
    spec, theta, X, f, y = _bn_case(2048)
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01)
    l_ref = ref.train_step(0, 2048)
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01)
    with pytest.raises(eh.EngineError, match="eh_dp_bn_stats"):
        eng.dp_grad(0, 512)                                   # BatchNorm without the global statistics is refused
    eng.set_bn_shift(X.mean(axis=1))
    gptr, gn = eng.device_buffer(eh._lib.EH_BUF_GRAD)
    bptr, bn = eng.device_buffer(eh._lib.EH_BUF_BNSTAT)
    gbuf = torch.as_tensor(eh.dp._DevArray(gptr, gn), device="cuda")
    bbuf = torch.as_tensor(eh.dp._DevArray(bptr, bn), device="cuda")
    stat = torch.zeros_like(bbuf)
    for k in range(4):                                        # first all-reduce: [sum (x-c) | sum (x-c)^2 | n] of every shard
        eng.dp_bn_stats(k * 512, 512); eng.synchronize()
        stat += bbuf
    assert float(stat[64]) == 2048
    acc = torch.zeros_like(gbuf)
    for k in range(4):                                        # every "rank" runs its shard with the GLOBAL statistics
        eng.dp_bn_stats(k * 512, 512); eng.synchronize()
        bbuf.copy_(stat); torch.cuda.synchronize()
        eng.dp_grad(k * 512, 512); eng.synchronize()
        acc += gbuf
    gbuf.copy_(acc); torch.cuda.synchronize()
    l = eng.dp_apply(want_loss=True)
    assert l == pytest.approx(l_ref, rel=1e-5)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 2e-6
    ref.close(); eng.close()


def test_sync_batchnorm_data_parallel_driver_world_size_one_nccl():
    import socket
    import torch
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        spec, theta, X, f, y = _bn_case(4096)
        ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01)
        for i in range(8):
            ref.train_step(i * 512, 512, want_loss=False)
        rm0, rv0 = ref.get_bn_state()
        for fused in (True, False):
            eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01)
            drv = eh.dp.DataParallel(eng, fused=fused)
            for i in range(8):
                drv.step(i * 512, 512)
            eng.synchronize()
            torch.cuda.synchronize()
            assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 3e-5, fused
            rm, rv = eng.get_bn_state()
            assert util.relerr(rm, rm0) <= 1e-5 and util.relerr(rv, rv0) <= 1e-5, fused
            eng.close()
        ref.close()
    finally:
        dist.destroy_process_group()


# the row-split kernel has its own record staging, mechanistic stage and epilogue: run the feature matrix through it too
def test_width_128_gather_indices_and_windows():
    spec, theta, X, f, y = _rs6_case(32, (128, 128), 3000)
    eng = util.load_engine(spec, theta, X, f, y)
    _check_grad(spec, theta, X, f, y, eng, first=137, count=701)
    idx = np.random.default_rng(3).permutation(3000)[:1111].astype(np.int32)
    _check_grad(spec, theta, X, f, y, eng, idx=idx)
    _check_grad(spec, theta, X, f, y, eng, idx=idx[:7])
    eng.close()


def test_width_128_all_masked_batch_and_epoch_driver():
    spec, theta, X, f, y = _rs6_case(20, (80, 128), 1000)
    name = list(y)[0]
    eng = util.load_engine(spec, theta, X, f, {name: np.full(1000, np.nan, np.float32)})
    loss, grad, nv = eng.loss_and_grad()
    assert np.isnan(loss) and nv == 0 and not grad.any()
    eng.close()
    a = util.load_engine(spec, theta, X, f, y); a.opt_init("Adam", 0.002)
    b = util.load_engine(spec, theta, X, f, y); b.opt_init("Adam", 0.002)
    mean_loss, ns = a.train_epoch(300, shuffle=False)
    losses = [b.train_step(s, min(300, 1000 - s)) for s in range(0, 1000, 300)]
    assert ns == 4 and np.array_equal(a.get_params(), b.get_params())
    assert mean_loss == pytest.approx(np.mean(losses), rel=1e-6)
    a.opt_init("Descent", 0.0)                              # nothing moves: a shuffled full-batch epoch visits every sample once
    l_plain, _ = a.train_epoch(1000, shuffle=False)
    l_shuf, _ = a.train_epoch(1000, seed=7, shuffle=True)
    assert l_shuf == pytest.approx(l_plain, rel=1e-5)
    a.close(); b.close()


@pytest.mark.parametrize("kind", ["rmse", "mae", "nseLoss"])
def test_width_128_other_training_losses(kind):
    spec, theta, X, f, y = _rs6_case(32, (128, 96), 1200)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_training_loss(kind)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=kind)
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    eng.close()


def test_width_128_input_batchnorm():
    spec, theta, X, f, y = _rs6_case(32, (128, 128), 1500)
    spec.input_batchnorm = True
    X = (X * np.linspace(1, 40, 32)[:, None] + np.linspace(-100, 300, 32)[:, None]).astype(np.float32)      # raw-looking predictors
    eng = util.load_engine(spec, theta, X, f, y)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, bn_state=ho.bn_init(spec))
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    eng.opt_init("Adam", 0.002)
    batches = [(i * 250, 250) for i in range(6)]
    losses = [eng.train_step(*b) for b in batches]
    st = ho.bn_init(spec)
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, lr=0.002, dtype=np.float32, bn_state=st)
    assert np.allclose(losses, l_ref, rtol=1e-4)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    rm, rv = eng.get_bn_state()
    assert util.relerr(rm, st["mean"]) <= 1e-5 and util.relerr(rv, st["var"]) <= 1e-5
    out = eng.forward(0, params=False)                      # test mode: running statistics
    ref = ho.forward(spec, eng.get_params().astype(np.float64), X, f, bn_state=st, train_mode=False)
    assert util.relerr(out[list(y)[0]], ref[list(y)[0]]) <= 2e-5
    eng.close()


@pytest.mark.parametrize("targets,nets", [
    (["RECO", "GPP", "NEE"], None),
    (["NEE", "RECO"], [([0, 1], [64, 48]), ([2, 3], [64, 80])]),        # two nets side by side: block-diagonal layers of width 128
])
def test_width_128_fluxpart_multi_target_and_multinn(targets, nets):
    B = 900
    X, f, y = _flux_data(B)
    spec = ho.HybridSpec(4, [96, 128], "fluxpart", dict(FLUX_PARAMS), ["RUE", "Rb"], ["Q10"], targets, "tanh", True, nets=nets)
    theta = ho.init_theta(spec, 2, np.float32)
    yt = {t: y[t] for t in targets}
    eng = util.load_engine(spec, theta, X, f, yt)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, yt)
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    m, _ = eng.eval(0)
    ref = ho.forward(spec, theta.astype(np.float64), X, f)
    for i, t in enumerate(targets):
        yy = yt[t].astype(np.float64)
        assert m[i]["mse"] == pytest.approx(ho.loss_fn(ref[t], yy, ~np.isnan(yy), "mse"), rel=3e-5)
    eng.opt_init("Adam", 0.003)
    batches = [(i * 150, 150) for i in range(6)]
    losses = [eng.train_step(*b) for b in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, yt, batches, lr=0.003, dtype=np.float32)
    assert np.allclose(losses, l_ref, rtol=1e-4)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    eng.close()


# ----------------------------------------------------------------------------------------------
# train(distributed=True): the epoch loop sharded over ranks (here: one rank, RCCL) must reproduce train()
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bn", [False, True])
def test_train_front_door_distributed_matches_single_process(bn):
    import socket
    import torch
    import torch.distributed as dist
    cols = eh.synthetic.make_synth_rbq10(3000, seed=5, nan_frac=0.05)
    if not bn:
        cols = dict(cols); cols["sw_pot"] = cols["sw_pot"] / 50; cols["dsw_pot"] = cols["dsw_pot"] / 50
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True, input_batchnorm=bn)
    kw = dict(nepochs=6, batchsize=256, opt=eh.Adam(0.01), loss_types=["mse", "r2"], random_seed=11)
    ref = eh.train(model, cols, **kw)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        out = eh.train(model, cols, distributed=True, **kw)
    finally:
        dist.destroy_process_group()
    assert out.best_epoch == ref.best_epoch
    assert np.max(np.abs(out.ps - ref.ps)) <= 5e-5 * max(1.0, float(np.max(np.abs(ref.ps))))
    for a, b in zip(out.val_history, ref.val_history):
        assert a["mse"]["sum"] == pytest.approx(b["mse"]["sum"], rel=2e-4)
    assert util.relerr(out.val_obs_pred["reco_pred"], ref.val_obs_pred["reco_pred"]) <= 1e-4


def test_hipgraph_replay_of_a_step_sequence_equals_plain_steps():
    spec, theta, X, f, y = util.rbq10_case(7 * 512, "tanh", True, 0.1)
    for fused in (0, 1):
        ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01); ref.set_option("fused_update", fused)
        eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01); eng.set_option("fused_update", fused)
        if fused:
            with pytest.raises(eh.EngineError, match="run one training step first"):
                eng.graph_begin()                            # a fused-mode graph has to start with an update pending
        ref.train_step(6 * 512, 512, want_loss=False); eng.train_step(6 * 512, 512, want_loss=False)
        for rep in range(3):
            for i in range(6):
                ref.train_step(i * 512, 512, want_loss=False)
        eng.graph_begin()
        for i in range(6):                                   # recorded, not run: 6 steps = one full rotation of the engine state
            eng.train_step(i * 512, 512, want_loss=False)
        g = eng.graph_end()
        for rep in range(3):
            eng.graph_launch(g)
        assert np.max(np.abs(eng.get_params() - ref.get_params())) <= (2e-6 if fused else 0.0), fused
        if fused:
            with pytest.raises(eh.EngineError, match="not in the state"):
                eng.graph_launch(g)                          # get_params flushed the pending update
        eng.train_step(0, 512, want_loss=False)
        with pytest.raises(eh.EngineError, match="not in the state"):
            eng.graph_launch(g)                              # one step off the recorded rotation
        eng.graph_begin()
        for i in range(5):
            eng.train_step(i * 512, 512, want_loss=False)
        with pytest.raises(ValueError, match="rotation state"):
            eng.graph_end()
        eng.close(); ref.close()


# ----------------------------------------------------------------------------------------------
# moment-based training losses (loss_fn.jl:75-77,105-174): two passes per step.  Their gradient is built from
# differences of batch moments ((r - 1), (alpha - 1), ...), which amplifies fp32 rounding: bound 1e-4, not 1e-5.
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["pearsonLoss", "kgeLoss", "pbkgeLoss"])
@pytest.mark.parametrize("shape", ["rbq10", "rbq10+bn", "wide"])
def test_moment_based_training_losses(kind, shape):
    if shape == "wide":
        spec, theta, X, f, y = _rs6_case(20, (96, 128), 1500)
    elif shape == "rbq10+bn":
        spec, theta, X, f, y = _bn_case(1500)
    else:
        spec, theta, X, f, y = util.rbq10_case(1500, "tanh", True, 0.1)
    bn = ho.bn_init(spec) if shape == "rbq10+bn" else None
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_training_loss(kind)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=kind, bn_state=bn)
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=1e-4) and util.relerr(grad, g0) <= 1e-4
    l1, g1, _ = eng.loss_and_grad(first=100, count=777)
    sl = slice(100, 877)
    l10, g10, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X[:, sl], {k: v[sl] for k, v in f.items()}, {k: v[sl] for k, v in y.items()},
                                   kind=kind, bn_state=ho.bn_init(spec) if bn is not None else None)
    assert l1 == pytest.approx(l10, rel=1e-4) and util.relerr(g1, g10) <= 1e-4
    eng.opt_init("Adam", 0.003)
    batches = [(i * 300, 300) for i in range(5)]
    losses = [eng.train_step(*b) for b in batches]
    st = ho.bn_init(spec) if bn is not None else None
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, lr=0.003, dtype=np.float32, kind=kind, bn_state=st)
    assert np.allclose(losses, l_ref, rtol=5e-4)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 2e-4 * max(1.0, float(np.max(np.abs(th_ref))))
    if shape != "wide":
        with pytest.raises(NotImplementedError, match="forward passes"):
            eng.set_option("fused_update", 1)
    eng.close()


def test_train_front_door_with_kge_loss_improves_kge():
    cols = eh.synthetic.make_synth_rbq10(4000, seed=5, nan_frac=0.05)
    cols = dict(cols); cols["sw_pot"] = cols["sw_pot"] / 50; cols["dsw_pot"] = cols["dsw_pot"] / 50
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    out = eh.train(model, cols, nepochs=15, batchsize=256, opt=eh.Adam(0.01), training_loss="kgeLoss", loss_types=["kge", "mse"], random_seed=3)
    assert out.val_history[-1]["kge"]["sum"] > out.val_history[0]["kge"]["sum"] + 0.2
    assert out.val_history[-1]["kge"]["sum"] > 0.8


# ----------------------------------------------------------------------------------------------
# extra_loss = lam * weight_l2(ps; normalize)  (extract_weights.jl:69-91, compute_loss.jl:31-34)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("shape", ["rbq10", "multinn", "wide"])
def test_weight_l2_extra_loss(normalize, shape):
    if shape == "wide":
        spec, theta, X, f, y = _rs6_case(20, (96, 128), 900)
    elif shape == "multinn":
        spec = ho.HybridSpec(4, [1], "rbq10", dict(ho.RBQ10_PARAMS), ["rb", "Q10"], [], ["reco"], "tanh", True, nets=[([0, 1], [8, 8]), ([2, 3], [16, 8])])
        rng = np.random.default_rng(5)
        X = rng.standard_normal((4, 900)).astype(np.float32)
        f = {"ta": rng.uniform(0, 30, 900).astype(np.float32)}
        y = {"reco": rng.uniform(1, 9, 900).astype(np.float32)}
        theta = ho.init_theta(spec, 6, np.float32)
    else:
        spec, theta, X, f, y = util.rbq10_case(900, "tanh", True, 0.1)
    lam = 0.37 * (int(ho.weight_mask(spec).sum()) / 10 if normalize else 1.0)       # comparable strength either way
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_weight_l2(lam, normalize)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, l2=(lam, normalize))
    lp, gp, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
    assert util.relerr(g0, gp) > 2e-4                         # the term is not negligible in this test
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    eng.opt_init("Adam", 0.003)
    batches = [(i * 300, 300) for i in range(3)] * 2
    losses = [eng.train_step(*b) for b in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, lr=0.003, dtype=np.float32, l2=(lam, normalize))
    assert np.allclose(losses, l_ref, rtol=1e-4)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    eng.set_weight_l2(0.0)
    l1, g1, _ = eng.loss_and_grad()
    l10, g10, _ = ho.loss_and_grad(spec, eng.get_params().astype(np.float64), X, f, y)
    assert l1 == pytest.approx(l10, rel=TOL) and util.relerr(g1, g10) <= TOL
    if shape == "rbq10":
        eng.set_weight_l2(lam)
        with pytest.raises(NotImplementedError, match="weight_l2"):
            eng.set_option("fused_update", 1)
    eng.close()


@pytest.mark.parametrize("shape", ["multinn", "multinn_lform", "single"])
def test_weight_l2_terms_per_network(shape):
    """several extra-loss terms -- the reference's own example `l2_Rb = lambda * weight_l2(ps.Rb; normalize = true)`
    (src/utils/extract_weights.jl:64), one per network with its own lambda, and a bias term -- as one coefficient per entry
    (eh_set_weight_l2_coef) against the oracle's term-by-term walk: loss, gradient, an Adam trajectory"""
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
    from easyhybrid_jl_amd.train import _extra_terms
    coef = model.l2_coefficients(_extra_terms(terms))
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_weight_l2_coef(coef)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, l2=oterms)
    lp, gp, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
    assert util.relerr(g0, gp) > 2e-4                         # the terms are not negligible in this test
    assert nv == sum(nv0) and loss == pytest.approx(l0, rel=TOL) and util.relerr(grad, g0) <= TOL
    eng.opt_init("Adam", 0.003)
    batches = [(i * 300, 300) for i in range(3)] * 2
    losses = [eng.train_step(*b) for b in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, lr=0.003, dtype=np.float32, l2=oterms)
    assert np.allclose(losses, l_ref, rtol=1e-4)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    # the one-lambda form replaces the coefficients, and the other way round; NULL switches the extra loss off
    eng.set_params(theta)
    eng.set_weight_l2(0.37, False)
    l1, g1, _ = eng.loss_and_grad()
    l10, g10, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, l2=(0.37, False))
    assert l1 == pytest.approx(l10, rel=TOL) and util.relerr(g1, g10) <= TOL
    eng.set_weight_l2_coef(coef)
    l2_, g2, _ = eng.loss_and_grad()
    assert l2_ == pytest.approx(l0, rel=TOL) and util.relerr(g2, g0) <= TOL
    eng.set_weight_l2_coef(None)
    l3, g3, _ = eng.loss_and_grad()
    assert l3 == pytest.approx(lp, rel=TOL) and util.relerr(g3, gp) <= TOL
    with pytest.raises(ValueError):
        eng.set_weight_l2_coef(coef[:-1])
    with pytest.raises(ValueError):
        eng.set_weight_l2_coef(-coef)
    eng.close()


def test_train_front_door_with_per_network_weight_l2_terms():
    """TrainConfig.extra_loss as the NamedTuple of terms the reference's closure returns: each entry and their sum in the history
    (compute_loss.jl:39-44), the penalised network's weights shrink, the other one's do not"""
    rng = np.random.default_rng(2)
    n = 3000
    cols = {"a": rng.standard_normal(n).astype(np.float32), "b": rng.standard_normal(n).astype(np.float32), "c": rng.standard_normal(n).astype(np.float32),
            "ta": rng.uniform(0, 30, n).astype(np.float32)}
    cols["reco"] = ((1.5 + np.tanh(cols["a"])) * (1.8 ** (0.1 * (cols["ta"] - 15.0)))).astype(np.float32)
    pars = dict(eh.synthetic.RBQ10_PARAMS)
    model = eh.constructHybridModel({"rb": ["a", "b"], "Q10": ["c"]}, ["ta"], ["reco"], eh.RbQ10, pars, [], hidden_layers={"rb": [16, 16], "Q10": [8]},
                                    activation="tanh", scale_nn_outputs=True)
    kw = dict(nepochs=8, batchsize=256, opt=eh.Adam(0.01), random_seed=3)
    plain = eh.train(model, cols, **kw)
    reg = eh.train(model, cols, extra_loss={"l2_rb": eh.WeightL2(0.1, net="rb"), "l2_Q10": eh.WeightL2(1e-6, net="Q10", normalize=True)}, **kw)
    mrb, mq = model.l2_mask("rb"), model.l2_mask("Q10")
    assert np.sum(reg.ps[mrb] ** 2) < 0.7 * np.sum(plain.ps[mrb] ** 2)
    last = reg.val_history[-1]["extra_loss"]
    assert set(last) == {"l2_rb", "l2_Q10", "sum"} and last["sum"] == pytest.approx(last["l2_rb"] + last["l2_Q10"])
    assert last["l2_rb"] == pytest.approx(0.1 * float(np.sum(reg.ps[mrb].astype(np.float64) ** 2)), rel=0.3)
    assert last["l2_Q10"] == pytest.approx(1e-6 * float(np.mean(reg.ps[mq].astype(np.float64) ** 2)), rel=0.3)


def test_weight_l2_through_the_data_parallel_seam():
    """the extra loss is a function of the replicated parameters: the shards exchange raw data sums only, eh_dp_apply adds
    2 lambda w to the weight gradients and lambda sum w^2 to the loss once (four virtual shards against the plain step)"""
    import torch
    spec, theta, X, f, y = util.rbq10_case(2048, "tanh", True, 0.1)
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01); ref.set_weight_l2(0.2, False)
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01); eng.set_weight_l2(0.2, False)
    ptr, n = eng.device_buffer(eh._lib.EH_BUF_GRAD)
    buf = torch.as_tensor(eh.dp._DevArray(ptr, n), device="cuda")
    for step in range(3):
        l_ref = ref.train_step(0, 2048)
        acc = torch.zeros_like(buf)
        for k in range(4):
            eng.dp_grad(k * 512, 512); eng.synchronize(); acc += buf
        buf.copy_(acc); torch.cuda.synchronize()
        l = eng.dp_apply(want_loss=True)
        assert l == pytest.approx(l_ref, rel=1e-5)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 3e-6
    ref.close(); eng.close()


def test_train_front_door_with_weight_l2_shrinks_the_weights():
    cols = eh.synthetic.make_synth_rbq10(3000, seed=5, nan_frac=0.05)
    cols = dict(cols); cols["sw_pot"] = cols["sw_pot"] / 50; cols["dsw_pot"] = cols["dsw_pot"] / 50
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    kw = dict(nepochs=10, batchsize=256, opt=eh.Adam(0.01), random_seed=3)
    plain = eh.train(model, cols, **kw)
    reg = eh.train(model, cols, extra_loss=eh.WeightL2(0.05, normalize=False), **kw)
    m = model.weight_mask()
    assert np.sum(reg.ps[m] ** 2) < 0.7 * np.sum(plain.ps[m] ** 2)
    last = reg.val_history[-1]["extra_loss"]
    assert last["weight_l2"] == pytest.approx(0.05 * float(np.sum(reg.ps[m].astype(np.float64) ** 2)), rel=0.3) and last["sum"] == last["weight_l2"]
    with pytest.raises(NotImplementedError, match="extra_loss"):
        eh.train(model, cols, extra_loss=lambda yhat, ps: 0.0, **kw)


# ----------------------------------------------------------------------------------------------
# "specialize": step kernels compiled at run time with the model descriptor as a compile-time constant (eh_jit.hip)
# ----------------------------------------------------------------------------------------------
def _spec_cases(case):
    if case == "rbq10":
        return util.rbq10_case(3000, "tanh", True, 0.1)
    if case == "rbq10-relu-unscaled":
        return util.rbq10_case(3000, "relu", False, 0.1, hidden=(32, 8, 16))
    if case == "config3":
        spec = ho.expo2pool_spec((64, 64), "tanh", True)
        X, f, y = ho.make_synth_expo2pool(3000, 7, 0.05)
        return spec, ho.init_theta(spec, 3, np.float32), X, f, y
    if case == "wide":
        return _rs6_case(20, (96, 128), 3000)
    raise ValueError(case)


@pytest.mark.parametrize("case", ["rbq10", "rbq10-relu-unscaled", "config3", "wide"])
def test_specialized_kernels_equal_the_kernels_built_ahead_of_time(case):
    spec, theta, X, f, y = _spec_cases(case)
    a = util.load_engine(spec, theta, X, f, y)
    a.set_option("specialize", 0)                            # (explicit: a test run may force it on through EH_SPECIALIZE)
    b = util.load_engine(spec, theta, X, f, y)
    b.set_option("specialize", 1)
    la, ga, na = a.loss_and_grad(first=7, count=2900)
    lb, gb, nb = b.loss_and_grad(first=7, count=2900)
    assert b.jit_status()[0] == 1 and a.jit_status()[0] == 0, b.jit_status()[1]
    sl = slice(7, 2907)
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X[:, sl], {k: v[sl] for k, v in f.items()}, {k: v[sl] for k, v in y.items()})
    assert na == nb == sum(nv0)
    assert abs(lb - l0) <= TOL * abs(l0) and util.relerr(gb, g0) <= TOL
    assert abs(la - lb) <= 1e-6 * abs(la) and util.relerr(gb, ga) <= 2e-6          # same arithmetic, possibly different contraction
    for eng in (a, b):
        eng.opt_init("Adam", 0.01)
    batches = [(i * 500, 500) for i in range(6)]
    loss_a = [a.train_step(*bt) for bt in batches]
    loss_b = [b.train_step(*bt) for bt in batches]
    assert np.allclose(loss_a, loss_b, rtol=1e-5)
    assert np.max(np.abs(a.get_params() - b.get_params())) <= 2e-5
    ma, mb = a.eval(0)[0], b.eval(0)[0]
    for k in ("mse", "r2", "kge"):
        assert ma[0][k] == pytest.approx(mb[0][k], rel=1e-5, abs=1e-6)
    # a change of the training loss is a different descriptor: a second kernel pair is compiled, the first stays cached
    b.set_training_loss("mae"); a.set_training_loss("mae")
    la, ga, _ = a.loss_and_grad(); lb, gb, _ = b.loss_and_grad()
    assert b.jit_status()[0] == 2 and abs(la - lb) <= 1e-6 * abs(la) and util.relerr(gb, ga) <= 2e-6
    a.close(); b.close()


def test_specialized_fused_update_trajectory_matches_oracle():
    spec, theta, X, f, y = util.rbq10_case(2048, "tanh", True, 0.1)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.opt_init("Adam", 0.01)
    eng.set_option("fused_update", 1)
    eng.set_option("specialize", 1)
    batches = [(i * 256, 256) for i in range(8)]
    losses = [eng.train_step(a, b) for a, b in batches]
    assert eng.jit_status()[0] == 1
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32)
    assert np.allclose(losses, l_ref, rtol=1e-4)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 3e-5 * max(1.0, float(np.max(np.abs(th_ref))))
    eng.close()


def test_train_front_door_with_specialize():
    cols = eh.synthetic.make_synth_rbq10(4000, seed=3, nan_frac=0.05)
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    ref = eh.train(model, cols, nepochs=4, batchsize=256, opt=eh.Adam(0.01), random_seed=1)
    out = eh.train(model, cols, nepochs=4, batchsize=256, opt=eh.Adam(0.01), random_seed=1, specialize=True)
    assert out.val_history[-1]["mse"]["sum"] == pytest.approx(ref.val_history[-1]["mse"]["sum"], rel=1e-4)


def test_compiled_kernels_are_cached_on_disk_and_a_damaged_entry_is_rebuilt(tmp_path, monkeypatch):
    monkeypatch.setenv("EH_JIT_CACHE", str(tmp_path))
    spec, theta, X, f, y = util.rbq10_case(900, "swish", True, 0.1, hidden=(24, 24))
    res = []
    for attempt in range(3):
        eng = util.load_engine(spec, theta, X, f, y)
        eng.set_option("specialize", 1)
        res.append(eng.loss_and_grad())
        assert eng.jit_status()[0] == 1, eng.jit_status()[1]
        eng.close()
        files = list(tmp_path.glob("*.eco"))
        assert len(files) == 1 and files[0].stat().st_size > 10000
        if attempt == 1:                                    # damage the entry: the next engine must notice, drop it and compile again
            files[0].write_bytes(files[0].read_bytes()[:5000])
    for l, g, n in res[1:]:
        assert l == res[0][0] and np.array_equal(g, res[0][1]) and n == res[0][2]


def test_hipgraph_capture_as_the_first_use_of_a_specialised_engine():
    # eh_graph_begin compiles and loads the run-time kernels before the capture starts (a module load inside a capture is not allowed)
    spec, theta, X, f, y = util.rbq10_case(6 * 512, "tanh", True, 0.1)
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01)
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01)
    eng.set_option("specialize", 1)
    eng.graph_begin()
    for i in range(6):
        eng.train_step(i * 512, 512, want_loss=False)
    g = eng.graph_end()
    assert eng.jit_status()[0] == 1, eng.jit_status()[1]
    for rep in range(2):
        eng.graph_launch(g)
        for i in range(6):
            ref.train_step(i * 512, 512, want_loss=False)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 1e-6
    eng.close(); ref.close()


# ----------------------------------------------------------------------------------------------
# models without a network: every parameter global or fixed (the reference builds `NN = Chain()` and its forward goes
# global / fixed parameters -> M, src/models/GenericHybridModel.jl:112-125,376-406)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("glob,n_pred", [(["rb", "Q10"], 0), (["Q10"], 0), (["rb"], 2)])
def test_model_without_a_network(glob, n_pred):
    B = 3000
    spec = ho.HybridSpec(n_pred, [], "rbq10", dict(ho.RBQ10_PARAMS), [], list(glob), ["reco"], "tanh", False)
    assert spec.n_theta == len(glob)
    X0, f, y = ho.make_synth_rbq10(B, 9, 0.15)
    X = X0[:n_pred]                                            # predictors may be listed: without a neural parameter nothing reads them
    theta = (ho.init_theta(spec, 1, np.float32) + np.float32(0.3)).astype(np.float32)
    eng = util.load_engine(spec, theta, X, f, y)
    assert eng.n_theta == len(glob)
    _check_grad(spec, theta, X, f, y, eng=eng)
    _check_grad(spec, theta, X, f, y, eng=eng, first=17, count=1234)
    idx = np.random.default_rng(4).permutation(B)[:700].astype(np.int32)
    _check_grad(spec, theta, X, f, y, eng=eng, idx=idx)
    out = eng.forward(eh.EH_SPLIT_TRAIN)
    ref = ho.forward(spec, theta.astype(np.float64), X, f)
    assert util.relerr(out["reco"], ref["reco"]) <= TOL
    for p in ("rb", "Q10"):
        assert util.relerr(out["parameters"][p], np.broadcast_to(ref["parameters"][p], (B,))) <= TOL
    m, _ = eng.eval(eh.EH_SPLIT_TRAIN)
    ev, _ = ho.evaluate(spec, theta.astype(np.float64), X, f, y, ("mse", "r2"))
    assert m[0]["mse"] == pytest.approx(ev["mse"]["reco"], rel=2e-5) and m[0]["r2"] == pytest.approx(ev["r2"]["reco"], abs=2e-5)
    eng.opt_init("Adam", 0.01)
    batches = [(i * 500, 500) for i in range(6)]
    losses = [eng.train_step(a, n) for a, n in batches]
    th_ref, l_ref = ho.train_steps(spec, theta, X, f, y, batches, dtype=np.float32)
    assert np.allclose(losses, l_ref, rtol=2e-5)
    assert np.max(np.abs(eng.get_params() - th_ref)) <= 2e-5
    mean_loss, nsteps = eng.train_epoch(512, seed=3, shuffle=True)
    assert nsteps == 6 and np.isfinite(mean_loss)
    eng.close()


def test_model_without_a_network_through_the_front_door():
    cols = __import__("easyhybrid_jl_amd.synthetic", fromlist=["x"]).make_synth_rbq10(4000, seed=1)
    # (start_from_default = false: the table's defaults ARE the values the series was made with)
    model = eh.constructHybridModel([], ["ta"], ["reco"], eh.RbQ10, {"rb": (3.0, 0.0, 13.0), "Q10": (2.0, 1.0, 4.0)}, [], ["rb", "Q10"], start_from_default=False)
    assert model.NN == [] and model.n_theta == 2 and model.fixed_param_names == []
    res = eh.train(model, cols, nepochs=60, batchsize=500, opt=eh.Adam(0.05), random_seed=1)
    # the synthetic series was made with rb ~ 3 (+- a covariate term of mean zero) and Q10 = 2: two global constants recover them
    assert res.val_history[-1]["mse"]["reco"] < res.val_history[0]["mse"]["reco"]
    from easyhybrid_jl_amd.models import scale_single_param
    _, raw = model.unpack(res.ps)
    q10, rb = (float(scale_single_param(n, raw[n], model.parameters)[0]) for n in ("Q10", "rb"))
    assert abs(q10 - 2.0) < 0.15 and abs(rb - 3.0) < 0.3, (q10, rb)
    with pytest.raises(ValueError):
        eh.constructHybridModel([], ["ta"], ["reco"], eh.RbQ10, {"rb": (3.0, 0.0, 13.0), "Q10": (2.0, 1.0, 4.0)}, ["rb"], ["Q10"])      # a neural parameter without predictors
    with pytest.raises(ValueError):
        eh.constructHybridModel([], ["ta"], ["reco"], eh.RbQ10, {"rb": (3.0, 0.0, 13.0), "Q10": (2.0, 1.0, 4.0)}, [], [])               # nothing to train


def test_model_without_a_network_two_targets_and_per_target_losses():
    """FluxPart (NEE, GPP) with RUE, Rb and Q10 all global: three raw parameters, two targets with their own gaps, each target its own
    loss -- rmse on a multi-target model takes the two forward passes (here: two runs of the mechanistic stage)"""
    rng = np.random.default_rng(12)
    B = 2500
    pars = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}
    spec = ho.HybridSpec(0, [], "fluxpart", pars, [], ["RUE", "Rb", "Q10"], ["NEE", "GPP"], "tanh", False)
    X = np.zeros((0, B), np.float32)
    f = {"SW_IN": (rng.random(B) * 400).astype(np.float32), "TA": (rng.random(B) * 30).astype(np.float32)}
    y = {"NEE": rng.standard_normal(B).astype(np.float32), "GPP": (rng.random(B) * 30).astype(np.float32)}
    y["NEE"][rng.random(B) < 0.3] = np.nan; y["GPP"][rng.random(B) < 0.1] = np.nan
    theta = (ho.init_theta(spec, 1, np.float32) + np.float32(0.2)).astype(np.float32)
    for kinds in (("mse", "mse"), ("mae", "rmse"), ("rmse", "nseLoss")):
        eng = util.load_engine(spec, theta, X, f, y)
        eng.set_training_loss(eh.PerTarget(kinds) if kinds[0] != kinds[1] else kinds[0])
        loss, grad, nv = eng.loss_and_grad()
        l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=kinds)
        assert nv == sum(nv0) and abs(loss - l0) <= TOL * abs(l0) and util.relerr(grad, g0) <= TOL, (kinds, loss, l0, util.relerr(grad, g0))
        eng.close()


def test_device_side_minibatch_indices_can_be_range_checked():
    """eh_train_step(idx on the device) trusts the caller by default (a whole epoch's permutation uploaded once: no copy, no check);
    the "check_idx" option (debug) range-checks them before every step: an index past the split is EH_EINVAL, not a wild read"""
    import torch
    spec, theta, X, f, y = util.rbq10_case(2000, "tanh", True, 0.0)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.opt_init("Adam", 0.01)
    good = torch.randperm(2000, dtype=torch.int32, device="cuda")
    bad = good.clone(); bad[700] = 2000; bad[900] = -3
    eng.set_option("check_idx", 1)
    l0 = eng.train_step(0, 1024, idx=good.cpu().numpy())
    eng.set_params(theta); eng.opt_init("Adam", 0.01)
    assert eng.train_step(0, 1024, idx=good.data_ptr()) == pytest.approx(l0, rel=1e-6)
    with pytest.raises(ValueError, match=r"2 of the 1024 device-side indices.*idx\[700\]"):
        eng.train_step(0, 1024, idx=bad.data_ptr())
    eng.train_step(1000, 1000, idx=bad.data_ptr())          # the window past the offenders is fine
    eng.close()
