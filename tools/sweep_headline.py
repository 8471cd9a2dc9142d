"""Wall-clock per step of the headline configuration (RbQ10 [2,16,16,1], batch 65 536, fused update, run-time specialised kernel)
over tile variants x workgroup counts:  python tools/sweep_headline.py [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B, NB = 65536, 64
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
cols = make_synth_rbq10(NB * B, seed=1)
X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
eng = model.engine(0)
eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
eng.set_params(model.initialparameters(1)); eng.opt_init("Adam", 0.01)
for fused in (1, 0):
    eng.set_option("fused_update", fused)
    for spec in (1, 0):
        eng.set_option("specialize", spec)
        for var in (0, 1, 2, 3):
            for mb in (256, 192, 128, 64):
                eng.set_option("variant", var); eng.set_option("max_blocks", mb)
                for s in range(200): eng.train_step((s % NB) * B, B, want_loss=False)
                eng.synchronize()
                t0 = time.perf_counter()
                for s in range(STEPS): eng.train_step((s % NB) * B, B, want_loss=False)
                eng.synchronize()
                us = 1e6 * (time.perf_counter() - t0) / STEPS
                print(json.dumps({"fused": fused, "specialize": spec, "variant": var, "max_blocks": mb, "us_per_step": round(us, 3)}), flush=True)
eng.close()
