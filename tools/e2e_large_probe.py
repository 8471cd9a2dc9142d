"""eh.train on the tutorial's large network: where the epoch loop's time goes, by TrainConfig switches"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
cols = make_synth_rbq10(5000, seed=42)
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                hidden_layers=[1024, 512, 256, 128, 64], activation="sigmoid", scale_nn_outputs=True, input_batchnorm=True)
kw = dict(nepochs=20, batchsize=64, opt=eh.RMSProp(0.01), loss_types=["mse", "nse"], keep_history=False)
for name, extra in (("default", {}), ("specialize=False", {"specialize": False}), ("specialize=False, fused_update=False", {"specialize": False, "fused_update": False})):
    eh.train(model, cols, **kw, **extra)
    t0 = time.perf_counter(); eh.train(model, cols, **kw, **extra); t1 = time.perf_counter()
    tm = eh.train(model, cols, timing=True, **kw, **extra).timing
    print(name, "call %.1f ms" % (1e3 * (t1 - t0)), {k: round(1e3 * v, 1) for k, v in tm.items() if k.endswith("_s")})
