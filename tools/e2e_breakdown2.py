"""as tools/e2e_breakdown.py, with the predictions read after every call (lazy and eager): the breakdown of each call"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
B, NB = 65536, 64
cols = make_synth_rbq10(NB * B, seed=42)
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"], hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
kw = dict(nepochs=10, batchsize=B, opt=eh.Adam(0.01), loss_types=["mse", "r2"], keep_history=False)
eh.train(model, cols, **kw)
import gc
for mode in ("lazy-noread", "gcoff-lazy-noread", "gcfreeze-lazy-noread"):
    if mode.startswith("gcoff"): gc.disable()
    if mode.startswith("gcfreeze"): gc.enable(); gc.collect(); gc.freeze()
    for rep in range(12):
        t0 = time.perf_counter(); r = eh.train(model, cols, timing=True, predictions="lazy", **kw); t1 = time.perf_counter()
        if False:
            n = len(r.val_obs_pred["reco_pred"])
        t2 = time.perf_counter()
        if mode == "lazy-sleep": time.sleep(0.05)
        if mode == "lazy-gc":
            import gc; r = None; gc.collect()
        tm = r.timing
        print("%-11s call %5.1f ms read %4.1f | " % (mode, 1e3 * (t1 - t0), 1e3 * (t2 - t1)) + " ".join("%s %.2f" % (k[:-2], 1e3 * tm[k]) for k in ("prepare_s", "engine_s", "upload_s", "setup_s", "initial_eval_s", "loop_s", "final_predictions_s"))
              + " | close %.2f" % (1e3 * (t1 - t0 - tm["call_s_before_close"])), flush=True)
