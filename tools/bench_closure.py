"""Cost of the interpreted mechanistic stage (EH_MECH_PROGRAM) next to the hand-derived registry kernels.
  python tools/bench_closure.py [--batch 65536] [--steps 500]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10


def rbq10_closure(*, ta, rb, Q10):
    return dict(reco=rb * Q10 ** (0.1 * (ta - 15.0)))


def flux_closure(*, sw, ta, vpd, alpha, gmax, rref, e0, k):       # 19 operations, two outputs
    lim = np.where(vpd > 10.0, np.exp(-k * (vpd - 10.0)), 1.0)
    gpp = lim * (alpha * sw * gmax) / (alpha * sw + gmax)
    reco = rref * np.exp(e0 * (1.0 / (10.0 + 46.02) - 1.0 / (np.maximum(ta, -40.0) + 46.02)))
    return dict(nee=reco - gpp, gpp=gpp, reco=reco)


ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--nbatches", type=int, default=8)
a = ap.parse_args()
B, NB = a.batch, a.nbatches
cols = make_synth_rbq10(NB * B, 1)
rng = np.random.default_rng(0)
cols["vpd"] = rng.uniform(0, 30, NB * B).astype(np.float32)
cols["sw"] = (cols["sw_pot"] * 8).astype(np.float32)
cols["nee"] = (cols["reco"] - 0.02 * cols["sw"]).astype(np.float32)


def run(label, model, X, F, Y, fused, specialize=0):
    eng = model.engine(0)
    eng.set_option("specialize", specialize)
    eng.set_data(0, X, F, Y)
    eng.set_params(model.initialparameters(1)); eng.opt_init("Adam", 0.01)
    eng.set_option("fused_update", fused)
    for s in range(30):
        eng.train_step((s % NB) * B, B, want_loss=False)
    eng.synchronize()
    t0 = time.perf_counter()
    for s in range(a.steps):
        eng.train_step((s % NB) * B, B, want_loss=False)
    eng.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / a.steps
    print(json.dumps({"model": label, "batch": B, "fused": fused, "compiled_at_run_time": eng.jit_status()[0], "us_per_step": round(us, 2), "samples_per_s": B / us * 1e6, "loss": eng.train_step(0, B)}))
    eng.close()


X2 = np.stack([cols["sw_pot"], cols["dsw_pot"]])
kw = dict(hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
reg = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"], **kw)
clo = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], rbq10_closure, dict(RBQ10_PARAMS), ["rb"], ["Q10"], **kw)
for fused in (0, 1):
    run("RbQ10 registry (K1|PS kernel), built ahead of time", reg, X2, [cols["ta"]], [cols["reco"]], fused)
    run("RbQ10 registry (K1|PS kernel), specialize option", reg, X2, [cols["ta"]], [cols["reco"]], fused, 1)
    run("RbQ10 closure (4-op program)", clo, X2, [cols["ta"]], [cols["reco"]], fused)
table = {"alpha": (0.05, 0.001, 0.2), "gmax": (20.0, 1.0, 60.0), "rref": (3.0, 0.1, 10.0), "e0": (150.0, 50.0, 400.0), "k": (0.05, 0.0, 0.5)}
flux = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["sw", "ta", "vpd"], ["nee"], flux_closure, table, ["alpha", "rref"], ["gmax", "e0", "k"], **kw)
run("flux closure (19-op program, 3 forcings)", flux, X2, [cols["sw"], cols["ta"], cols["vpd"]], [cols["nee"]], 0)
wide = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], rbq10_closure, dict(RBQ10_PARAMS), ["rb"], ["Q10"], hidden_layers=[128, 128], activation="tanh", scale_nn_outputs=True)
widr = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"], hidden_layers=[128, 128], activation="tanh", scale_nn_outputs=True)
run("RbQ10 registry, [2,128,128,1] row-split kernel, built ahead of time", widr, X2, [cols["ta"]], [cols["reco"]], 0)
run("RbQ10 registry, [2,128,128,1] row-split kernel, specialize option", widr, X2, [cols["ta"]], [cols["reco"]], 0, 1)
run("RbQ10 closure, [2,128,128,1] row-split kernel", wide, X2, [cols["ta"]], [cols["reco"]], 0)
