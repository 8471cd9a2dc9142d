"""A/B of the "specialize" option (step kernels compiled at run time with the model descriptor baked in) on the headline shape.
  python tools/bench_specialize.py [--steps 2000]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--hidden", type=int, nargs="+", default=[16, 16])
a = ap.parse_args()
B, NB = a.batch, 16
cols = make_synth_rbq10(NB * B, 1)
X = np.stack([cols["sw_pot"], cols["dsw_pot"]])
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                hidden_layers=a.hidden, activation="tanh", scale_nn_outputs=True)
for rep in range(2):
    for fused in (1, 0):
        for spec in (0, 1):
            eng = model.engine(0)
            eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
            eng.set_params(model.initialparameters(1)); eng.opt_init("Adam", 0.01)
            try:
                eng.set_option("fused_update", fused)
            except NotImplementedError:
                eng.close(); continue
            eng.set_option("specialize", spec)
            for s in range(100):
                eng.train_step((s % NB) * B, B, want_loss=False)
            eng.synchronize()
            t0 = time.perf_counter()
            for s in range(a.steps):
                eng.train_step((s % NB) * B, B, want_loss=False)
            eng.synchronize()
            us = 1e6 * (time.perf_counter() - t0) / a.steps
            n, log = eng.jit_status()
            print(json.dumps({"fused": fused, "specialize": spec, "us_per_step": round(us, 3), "jit_kernels": n, "log": log[:300], "loss": eng.train_step(0, B)}))
            eng.close()
