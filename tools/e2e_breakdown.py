"""eh.train on the headline data set (64 x 65 536 rows, 10 epochs): the wall clock of six calls and, from TrainConfig.timing, where the time
outside the epoch loop goes -- prepare | engine | upload | setup | initial evaluation | loop | final predictions | close"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
B, NB = 65536, 64
cols = make_synth_rbq10(NB * B, seed=42)
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"], hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
kw = dict(nepochs=10, batchsize=B, opt=eh.Adam(0.01), loss_types=["mse", "r2"], keep_history=False)
eh.train(model, cols, **kw)
for rep in range(6):
    t0 = time.perf_counter(); r = eh.train(model, cols, timing=True, **kw); dt = time.perf_counter() - t0
    tm = r.timing
    print("call %5.1f ms | " % (1e3 * dt) + " ".join("%s %.2f" % (k[:-2], 1e3 * tm[k]) for k in ("prepare_s", "engine_s", "upload_s", "setup_s", "initial_eval_s", "loop_s", "final_predictions_s"))
          + " | close %.2f" % (1e3 * (dt - tm["call_s_before_close"])), flush=True)
for mode in ("lazy", "eager"):
    plain, read = [], []
    for rep in range(7):
        t0 = time.perf_counter(); r = eh.train(model, cols, predictions=mode, **kw); t1 = time.perf_counter()
        n = len(r.val_obs_pred["reco_pred"]); t2 = time.perf_counter()
        plain.append(1e3 * (t1 - t0)); read.append(1e3 * (t2 - t1))
    print("predictions = %s: calls (ms):" % mode, " ".join("%.1f" % v for v in plain), "median %.1f max/min %.2f" % (float(np.median(plain)), max(plain) / min(plain)),
          "| first read of the predictions afterwards (ms):", " ".join("%.1f" % v for v in read))
