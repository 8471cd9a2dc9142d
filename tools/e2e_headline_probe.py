"""eh.train on the headline data set (10 epochs): where the time OUTSIDE the epoch loop goes, call by call"""
import sys, os, time, cProfile, pstats, io
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
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable(); eh.train(model, cols, **kw); pr.disable(); dt = time.perf_counter() - t0
    st = pstats.Stats(pr)
    keep = {}
    for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
        if name in ("set_data", "prepare_data", "split_data", "forward", "eval", "close", "__init__", "train_epoch", "take") and "easyhybrid" in fn:
            keep[name] = keep.get(name, 0.0) + ct
    print("call %5.1f ms  " % (1e3 * dt) + "  ".join("%s %.1f" % (k, 1e3 * v) for k, v in sorted(keep.items())), flush=True)
