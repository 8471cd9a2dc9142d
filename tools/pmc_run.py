"""Small fixed workload for rocprofv3 --pmc passes: 20 RbQ10 steps at B=65536 (and optionally other configs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
var = int(sys.argv[2]) if len(sys.argv) > 2 else 0
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
cols = make_synth_rbq10(max(B, 1 << 16), seed=1)
X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
eng = model.engine(0)
eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
eng.set_params(model.initialparameters(1)); eng.opt_init("Adam", 0.01)
eng.set_option("variant", var)
for _ in range(20):
    eng.train_step(0, B, want_loss=False)
eng.synchronize(); eng.close()
