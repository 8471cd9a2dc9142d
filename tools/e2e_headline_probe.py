"""eh.train on the headline data set (10 epochs): where the time OUTSIDE the epoch loop goes (cProfile, cumulative)"""
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
t0 = time.perf_counter(); eh.train(model, cols, **kw); print("call %.1f ms" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); eh.train(model, cols, **kw); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:5000])
