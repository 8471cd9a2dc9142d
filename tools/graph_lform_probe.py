"""Does a recorded hipGraph help the layer-wise form at small batches?  (tutorial net, B = 64 / 256 / 1024)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
hidden = [1024, 512, 256, 128, 64]
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                hidden_layers=hidden, activation="sigmoid", scale_nn_outputs=True, input_batchnorm=True)
for B in (64, 256, 1024):
    nb = 8
    cols = make_synth_rbq10(nb * B, seed=1)
    X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
    res = {}
    for mode in ("eager", "graph"):
        eng = model.engine(0)
        eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
        eng.set_params(model.initialparameters(1)); eng.opt_init("RMSProp", 0.001)
        for s in range(4): eng.train_step((s % nb) * B, B, want_loss=False)
        eng.synchronize()
        steps = 320
        if mode == "graph":
            eng.graph_begin()
            for s in range(nb): eng.train_step((s % nb) * B, B, want_loss=False)
            g = eng.graph_end()
            eng.graph_launch(g); eng.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps // nb): eng.graph_launch(g)
        else:
            for s in range(nb): eng.train_step((s % nb) * B, B, want_loss=False)
            eng.synchronize()
            t0 = time.perf_counter()
            for s in range(steps): eng.train_step((s % nb) * B, B, want_loss=False)
        eng.synchronize()
        res[mode] = (1e6 * (time.perf_counter() - t0) / steps, eng.get_params())
        eng.close()
    print(json.dumps({"batch": B, "eager_us": round(res["eager"][0], 1), "graph_us": round(res["graph"][0], 1),
                      "max_abs_param_diff": float(np.max(np.abs(res["eager"][1] - res["graph"][1])))}), flush=True)
