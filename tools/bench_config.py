"""Throughput of other BASELINE configs (parity-test shapes, not the headline bench line).
  python tools/bench_config.py c3 [--batch 262144] [--steps 200] [--fused 1]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import EXPO2POOL_PARAMS, RBQ10_PARAMS, RS6_PARAMS, make_synth_expo2pool, make_synth_fluxnet32, make_synth_rbq10

ap = argparse.ArgumentParser()
ap.add_argument("config", choices=["c1", "c2", "c3", "c5"])
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--fused", type=int, default=1)
ap.add_argument("--nbatches", type=int, default=8)
ap.add_argument("--variant", type=int, default=-1)
ap.add_argument("--specialize", type=int, default=0)
ap.add_argument("--row-split", type=int, default=0)
a = ap.parse_args()
if a.config == "c3":
    B = a.batch or 262144
    model = eh.constructHybridModel([f"x{i}" for i in range(8)], ["T"], ["Resp_obs"], eh.Expo2Pool, dict(EXPO2POOL_PARAMS),
                                    ["R0a", "ka", "R0b", "kb"], [], hidden_layers=[64, 64], activation="tanh", scale_nn_outputs=True)
    cols = make_synth_expo2pool(a.nbatches * B, 1)
    X = np.stack([cols[f"x{i}"] for i in range(8)]); F = [cols["T"]]; Y = [cols["Resp_obs"]]
    flop, byts = 29184, 40
elif a.config == "c5":          # configs[4]: MLP [32,128,128,6] + Rs_components (RbQ10 family, three pools), fp32 here
    B = a.batch or 65536
    model = eh.constructHybridModel([f"x{i}" for i in range(32)], ["ta"], ["R_soil"], eh.Rs_components, dict(RS6_PARAMS),
                                    list(RS6_PARAMS), [], hidden_layers=[128, 128], activation="tanh", scale_nn_outputs=True)
    cols = make_synth_fluxnet32(a.nbatches * B, 1)
    X = np.stack([cols[f"x{i}"] for i in range(32)]); F = [cols["ta"]]; Y = [cols["R_soil"]]
    flop, byts = 6 * (32 * 128 + 128 * 128 + 128 * 6), 4 * 34
    a.fused = 0                   # the row-split kernel has no fused-update mode
else:
    B = a.batch or (1024 if a.config == "c1" else 65536)
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    cols = make_synth_rbq10(a.nbatches * B, 1)
    X = np.stack([cols["sw_pot"], cols["dsw_pot"]]); F = [cols["ta"]]; Y = [cols["reco"]]
    flop, byts = 1824, 16
eng = model.engine(0)
eng.set_data(0, X, F, Y)
eng.set_params(model.initialparameters(1)); eng.opt_init("Adam", 0.01)
eng.set_option("fused_update", a.fused)
eng.set_option("specialize", a.specialize)
if a.row_split:
    eng.set_option("row_split", 1)
if a.variant >= 0:
    eng.set_option("variant", a.variant)
def run(n, base=0):
    for s in range(n):
        eng.train_step(((base + s) % a.nbatches) * B, B, want_loss=False)
run(20); eng.synchronize()
t0 = time.perf_counter(); run(a.steps, 20); eng.synchronize(); dt = time.perf_counter() - t0
us = 1e6 * dt / a.steps
print(json.dumps({"config": a.config, "batch": B, "fused": a.fused, "specialize": a.specialize, "us_per_step": us, "samples_per_s": B / us * 1e6,
                  "algorithmic_TFLOPs": flop * B / us / 1e6, "frac_f32_peak": flop * B / us / 1e6 / 157.3,
                  "algorithmic_GBps": byts * B / us / 1e3, "final_loss": eng.train_step(0, B)}))
eng.close()
