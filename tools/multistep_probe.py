"""eh_train_epoch on small minibatches: one launch per step against several steps per launch with the state in LDS ("multi_step" option)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
cols = make_synth_rbq10(NR, seed=42)
X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
for bn, act in ((True, "sigmoid"), (False, "tanh")):
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"], hidden_layers=[16, 16], activation=act, scale_nn_outputs=True, input_batchnorm=bn)
    for B in (64, 256):
        for multi in (0, 1):
            eng = model.engine(0)
            eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
            eng.set_params(model.initialparameters(1)); eng.opt_init("RMSProp", 0.01); eng.set_option("fused_update", 1); eng.set_option("multi_step", multi)
            for k in range(3): eng.train_epoch(B, seed=k, shuffle=True, want_loss=False)
            eng.synchronize()
            t0 = time.perf_counter()
            for k in range(20): eng.train_epoch(B, seed=10 + k, shuffle=True, want_loss=False)
            eng.synchronize()
            dt = time.perf_counter() - t0
            ns = 20 * -(-NR // B)
            print("input_batchnorm=%s %s batch %d multi_step=%d: %.2f us/step (%d steps)" % (bn, act, B, multi, 1e6 * dt / ns, ns), flush=True)
            eng.close()
