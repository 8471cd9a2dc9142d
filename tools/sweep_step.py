"""Kernel-time sweep helper (run under rocprofv3 --kernel-trace on the GPU box):
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sweep -- python3 tools/sweep_step.py
then tools/sweep_report.py gpurun_out/sweep prints mean kernel durations per configuration."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10

STEPS = 300
FUSED = int(os.environ.get("EH_FUSED", "0"))
configs = [(b, 256, v) for v in (0, 1, 2, 3) for b in (64, 65536, 131072, 1048576)] + \
          [(65536, mb, v) for v in (0, 1) for mb in (128,)]
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
cols = make_synth_rbq10(1 << 21, seed=1)
X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
eng = model.engine(0)
eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
eng.set_params(model.initialparameters(1))
eng.opt_init("Adam", 0.01)
eng.set_option("fused_update", FUSED)
plan = []
for b, mb, var in configs:
    eng.set_option("max_blocks", mb)
    eng.set_option("variant", var)
    for s in range(STEPS):
        eng.train_step(0, b, want_loss=False)
    eng.synchronize()
    plan.append({"batch": b, "max_blocks": mb, "variant": var, "steps": STEPS})
os.makedirs("gpurun_out", exist_ok=True)
json.dump(plan, open("gpurun_out/sweep_plan.json", "w"))
eng.close()
