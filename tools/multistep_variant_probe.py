"""small minibatches on the GENERIC kernels by tile variant: does a finer tile (16 samples: four busy waves at batch 64) shorten the in-kernel step?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
cols = make_synth_rbq10(4000, seed=42)
X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"], hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
for variant in (0, 1, 2, 3):
    for multi in (0, 1):
        eng = model.engine(0)
        eng.set_option("aot_spec", 0)
        try:
            eng.set_option("variant", variant)
        except Exception as e:
            print("variant", variant, "refused:", e); eng.close(); break
        eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
        eng.set_params(model.initialparameters(1)); eng.opt_init("Adam", 0.01); eng.set_option("fused_update", 1); eng.set_option("multi_step", multi)
        for k in range(3): eng.train_epoch(64, seed=k, shuffle=True, want_loss=False)
        eng.synchronize()
        t0 = time.perf_counter()
        for k in range(20): eng.train_epoch(64, seed=10 + k, shuffle=True, want_loss=False)
        eng.synchronize()
        print("variant %d multi_step=%d: %.2f us/step" % (variant, multi, 1e6 * (time.perf_counter() - t0) / (20 * 63)), flush=True)
        eng.close()
