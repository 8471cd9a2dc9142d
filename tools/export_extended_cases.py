"""Writes the EXTENDED cases for tools/emit_fixtures.jl -- what rounds 2-5 added to the path and the B = 12 RbQ10 x activation x scaling
cases do not touch -- as plain CSV under tests/golden/csv_for_emit_fixtures/ext_<case>/, every expected value from the oracle
(oracle/hybrid_oracle.py):  python tools/export_extended_cases.py

  ext_bn_rmsprop        input BatchNorm (train-mode step: batch statistics, running statistics after it; test-mode forward after it) + RMSProp(0.01)
  ext_adamw             AdamW(0.01, lambda = 0.1) one step          ext_descent        Descent(0.1) one step
  ext_loss_nse / _pearson / _kge   training_loss = :nseLoss / :pearsonLoss / :kgeLoss (loss_fn.jl:75-86,105-174): loss and gradient
  ext_two_targets_mean  two targets of a two-output closure, agg = mean (TrainingConfig.jl:76-77, compute_loss.jl:50-53)
  ext_weight_l2         extra_loss = (yhat, ps) -> (; l2 = 0.01 * weight_l2(ps)) (extract_weights.jl:69-91, compute_loss.jl:31-34)
  ext_multinn           MultiNNHybridModel: rb and Q10 from a network each, on predictor sets of their own (GenericHybridModel.jl:142-206)
  ext_chain_acts        hidden_layers = Chain(Dense(8, 6, relu)) under activation = tanh (NNModels.jl:145-219)
  ext_extra_two_outputs extra_loss = (yhat, ps) -> (; c = 0.05 * mean(yhat.reco .* yhat.half) * ps.Q10[1] + 0.5 * mean((yhat.reco .- yhat.half) .^ 2)): an entry
                        over two outputs of the model and a RAW global parameter (compute_loss.jl:31-34), one target, agg = sum
Never part of the build or the tests; regenerate when the oracle changes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import hybrid_oracle as ho
from tests import util

OUT = os.path.join(ROOT, "tests", "golden", "csv_for_emit_fixtures")
B = 12


def row(path, v, fmt="%.17g"):
    np.savetxt(path, np.atleast_2d(np.asarray(v, np.float64)), delimiter=",", fmt=fmt)


def data(seed=3, nan_at=(4,)):
    X, f, y = ho.make_synth_rbq10(B, seed, 0.0)
    X = (X / np.float32(50)).astype(np.float32)
    yv = y["reco"].copy(); yv[list(nan_at)] = np.nan
    return X, f, {"reco": yv}


def write(name, spec_lines, theta, X, f, y, expect):
    d = os.path.join(OUT, "ext_" + name)
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "spec.txt"), "w").write("\n".join(spec_lines) + "\n")
    row(os.path.join(d, "theta.csv"), theta, "%.9g")
    np.savetxt(os.path.join(d, "X.csv"), np.asarray(X, np.float64), delimiter=",", fmt="%.9g")
    for k, v in f.items(): row(os.path.join(d, k + ".csv"), v, "%.9g")
    for k, v in y.items(): row(os.path.join(d, k + ".csv"), v, "%.9g")
    for k, v in expect.items(): row(os.path.join(d, "expect_" + k + ".csv"), v)
    print("ext_" + name, "->", d)


def one_step(rule, theta, g, **kw):
    th, g = theta.astype(np.float32), g.astype(np.float32)
    if rule == "descent":
        return th - np.float32(kw["lr"]) * g
    if rule == "rmsprop":          # Optimisers.RMSProp(eta, rho = 0.9, eps = 1e-8): v = rho v + (1 - rho) g^2 ; theta -= eta g / (sqrt(v) + eps)
        v = np.float32(0.1) * g * g
        return th - g * (np.float32(kw["lr"]) / (np.sqrt(v) + np.float32(1e-8)))
    return ho.adam_step(th, g, ho.adam_init(th.size), kw.get("lr", 0.01), weight_decay=kw.get("wd", 0.0))


base = ["model=rbq10", "hidden=8,8", "activation=sigmoid", "scale_nn_outputs=true"]
spec = ho.rbq10_spec((8, 8), "sigmoid", True)
theta = ho.init_theta(spec, 11, np.float32)
X, f, y = data()
l, g, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
write("adamw", base + ["optimiser=AdamW", "lr=0.01", "lambda=0.1"], theta, X, f, y, dict(loss=[l], grad=g, theta_after_1=one_step("adamw", theta, g, lr=0.01, wd=0.1)))
write("descent", base + ["optimiser=Descent", "lr=0.1"], theta, X, f, y, dict(loss=[l], grad=g, theta_after_1=one_step("descent", theta, g, lr=0.1)))
for kind in ("nseLoss", "pearsonLoss", "kgeLoss"):
    lk, gk, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=kind)
    write("loss_" + kind[:-4].lower(), base + ["training_loss=" + kind], theta, X, f, y, dict(loss=[lk], grad=gk))
ll, gl, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, l2=(0.01, False))
write("weight_l2", base + ["l2_lambda=0.01"], theta, X, f, y, dict(loss=[ll], grad=gl))

# input BatchNorm + RMSProp: raw predictor scale (what the layer is there for)
sb = ho.rbq10_spec((8, 8), "sigmoid", True); sb.input_batchnorm = True
Xr = (X * np.float32(50)).astype(np.float32)
bn = ho.bn_init(sb)
lb, gb, _ = ho.loss_and_grad(sb, theta.astype(np.float64), Xr, f, y, bn_state=bn)
_, new = ho.batchnorm_input(np.asarray(Xr, np.float64), bn, True, np.dtype(np.float64)); bn.update(new)
th1 = one_step("rmsprop", theta, gb, lr=0.01)
yh_test = ho.forward(sb, th1.astype(np.float64), Xr, f, bn_state=bn, train_mode=False)["reco"]
write("bn_rmsprop", base + ["input_batchnorm=true", "optimiser=RMSProp", "lr=0.01"], theta, Xr, f, y,
      dict(loss=[lb], grad=gb, theta_after_1=th1, running_mean_after_1=bn["mean"], running_var_after_1=bn["var"], yhat_testmode_after_1=yh_test))

# two targets of a two-output closure, agg = mean
def reco2(*, ta, rb, Q10):
    reco = rb * Q10 ** (0.1 * (ta - 15.0))
    return dict(reco=reco, half=0.5 * rb + 0.015625 * ta)
util.register_closure("reco2", reco2, ["rb", "Q10"], ["ta"], ["reco", "half"])
s2 = ho.HybridSpec(2, [8, 8], "reco2", dict(ho.RBQ10_PARAMS), ["rb"], ["Q10"], ["reco", "half"], "tanh", True)
t2 = ho.init_theta(s2, 12, np.float32)
truth = ho.forward(s2, ho.init_theta(s2, 13, np.float32).astype(np.float64), X, f)
y2 = {"reco": y["reco"], "half": truth["half"].astype(np.float32).copy()}
y2["half"][[1, 7]] = np.nan
l2_, g2_, _ = ho.loss_and_grad(s2, t2.astype(np.float64), X, f, y2, agg="mean")
write("two_targets_mean", ["model=reco2", "hidden=8,8", "activation=tanh", "scale_nn_outputs=true", "agg=mean", "targets=reco,half"], t2, X, f, y2, dict(loss=[l2_], grad=g2_))

# MultiNN: rb from a network on (sw_pot, dsw_pot), Q10 from a network on a third predictor
rng = np.random.default_rng(5)
X3 = np.concatenate([X, rng.standard_normal((1, B)).astype(np.float32)], axis=0)
sm = ho.HybridSpec(3, [], "rbq10", dict(ho.RBQ10_PARAMS), ["rb", "Q10"], [], ["reco"], "tanh", True, nets=[([0, 1], [8]), ([2], [4])])
tm = ho.init_theta(sm, 14, np.float32)
lm, gm, _ = ho.loss_and_grad(sm, tm.astype(np.float64), X3, f, y)
write("multinn", ["model=rbq10_multinn", "hidden_rb=8", "hidden_Q10=4", "activation=tanh", "scale_nn_outputs=true"], tm, X3, f, y, dict(loss=[lm], grad=gm, yhat=ho.forward(sm, tm.astype(np.float64), X3, f)["reco"]))

# hidden_layers::Chain with an activation of its own on the second layer
sc = ho.rbq10_spec((8, 6), "tanh", True); sc.layer_activations = ["tanh", "relu"]
tc = ho.init_theta(sc, 15, np.float32)
lc, gc, _ = ho.loss_and_grad(sc, tc.astype(np.float64), X, f, y)
write("chain_acts", ["model=rbq10", "hidden=8,6", "activation=tanh", "chain=Dense(8,6,relu)", "scale_nn_outputs=true"], tc, X, f, y, dict(loss=[lc], grad=gc, yhat=ho.forward(sc, tc.astype(np.float64), X, f)["reco"]))

# extra_loss over two outputs of the two-output closure and the raw global parameter: the entry rides as one more output of the model's program
from easyhybrid_jl_amd.program import trace_extra_loss_mixed, _identity_entry_program
xfn = lambda yhat, ps: {"c": 0.05 * np.mean(yhat["reco"] * yhat["half"]) * ps.Q10[0] + 0.5 * np.mean((yhat["reco"] - yhat["half"]) ** 2)}
bounds = {"Q10": (ho.RBQ10_PARAMS["Q10"][1], ho.RBQ10_PARAMS["Q10"][2])}
progx, entx = trace_extra_loss_mixed(reco2, xfn, ["rb", "Q10"], ["ta"], ["reco"], ["Q10"], bounds)
ho.program_mech("reco2_x", progx.as_dict(), None)
sx = ho.HybridSpec(2, [8, 8], "reco2_x", dict(ho.RBQ10_PARAMS), ["rb"], ["Q10"], ["reco"], "tanh", True)
ho.loss_program("xl_ident", _identity_entry_program().as_dict(), None)
lx, gx, _ = ho.loss_and_grad(sx, t2.astype(np.float64), X, f, {"reco": y["reco"]}, extra=[(entx[0][1], "xl_ident", entx[0][2])])
write("extra_two_outputs", ["model=reco2", "hidden=8,8", "activation=tanh", "scale_nn_outputs=true", "targets=reco", "extra=two_outputs"], t2, X, f, {"reco": y["reco"]}, dict(loss=[lx], grad=gx))
