"""The headline step (RbQ10 [2,16,16,1], batch 65 536) in its two forms on ONE lease: one kernel per step with float atomics (what bench.py
times; what an unseeded train() or train(fused_update = True) runs) and the deterministic step + reduce/optimiser pair (what a seeded
train() takes at this batch size):  python tools/bench_step_modes.py [steps]
(Round 6 also built a third form -- one kernel per step with the workgroups' partial sums as fixed-point integers {2^-20, 2^-76} added with
64-bit integer atomics, order-free and so bitwise reproducible: 13.86 us against 14.6 for the pair and 10.2 for the float atomics on
one lease, and its branches cost the float form 0.6 us (9.60 -> 10.18 us, bench.py, same lease).  Removed.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
B, NB = 65536, 16
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cols = make_synth_rbq10(NB * B, seed=42)
X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"], hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
out = {}
for rep in range(2):
    for name, fused, ordered in (("float_atomics", 1, 0), ("step_plus_reduce", 0, 0)):
        eng = model.engine(0)
        eng.set_data(eh.EH_SPLIT_TRAIN, X, [cols["ta"]], [cols["reco"]])
        eng.set_params(model.initialparameters(161803)); eng.opt_init("Adam", 0.01)
        eng.set_option("fused_update", fused)
        if ordered: eng.set_option("ordered_sums", 1)
        for s in range(200): eng.train_step((s % NB) * B, B, want_loss=False)
        eng.synchronize()
        t0 = time.perf_counter()
        for s in range(steps): eng.train_step((s % NB) * B, B, want_loss=False)
        eng.synchronize()
        out.setdefault(name, []).append(round(1e6 * (time.perf_counter() - t0) / steps, 3))
        eng.close()
print(json.dumps({"what": "us per step, headline model, batch 65 536, kernels specialised ahead of time; two rounds on one lease", **out, "lib": os.environ.get("EASYHYBRID_HIP_LIB", "default")}))
