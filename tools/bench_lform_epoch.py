"""The layer-wise tutorial net through the epoch driver (what train() runs): eh_train_epoch over 62 x 64 + 32 rows, shuffled; us per step.
   python tools/bench_lform_epoch.py [epochs]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
E = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols = make_synth_rbq10(4000, seed=42)
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                hidden_layers=[1024, 512, 256, 128, 64], activation="sigmoid", scale_nn_outputs=True, input_batchnorm=True)
X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
eng = model.engine(0)
eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
eng.set_params(model.initialparameters(1)); eng.opt_init("RMSProp", 0.01)
for ep in range(3): eng.train_epoch(64, seed=ep, shuffle=True, want_loss=False)
eng.synchronize()
t0 = time.perf_counter()
for ep in range(E): eng.train_epoch(64, seed=10 + ep, shuffle=True, want_loss=False)
eng.synchronize()
print(json.dumps({"what": "eh_train_epoch, tutorial net, 63 steps per epoch (62 x 64 + 32 rows), shuffled", "epochs": E, "us_per_step": round(1e6 * (time.perf_counter() - t0) / E / 63, 2)}))
eng.close()
