"""Throughput of other BASELINE configs (parity-test shapes, not the headline bench line).
  python tools/bench_config.py c3 [--batch 262144] [--steps 200] [--fused 1]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import EXPO2POOL_PARAMS, RBQ10_PARAMS, RS6_PARAMS, make_synth_expo2pool, make_synth_fluxnet32, make_synth_fluxnet32_3f, make_synth_rbq10

ap = argparse.ArgumentParser()
ap.add_argument("config", choices=["c1", "c2", "c3", "c5"])
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--fused", type=int, default=1)
ap.add_argument("--nbatches", type=int, default=8)
ap.add_argument("--variant", type=int, default=-1)
ap.add_argument("--specialize", type=int, default=0)
ap.add_argument("--row-split", type=int, default=0)
ap.add_argument("--precision", default="", help="c5: bf16_fwd (default: bf16 forward, fp32-exact backward), bf16 (bf16 operands in both passes) or f32")
ap.add_argument("--epoch", type=int, default=0, help="also time N epochs of eh_train_epoch, contiguous and shuffled (what train() runs: every step gathers its records through the epoch's permutation)")
ap.add_argument("--n", type=int, default=0, help="resident samples (c5 default: 152 batches ~ 1e7 as BASELINE states; others nbatches * batch)")
a = ap.parse_args()
if a.config == "c3":
    B = a.batch or 262144
    model = eh.constructHybridModel([f"x{i}" for i in range(8)], ["T"], ["Resp_obs"], eh.Expo2Pool, dict(EXPO2POOL_PARAMS),
                                    ["R0a", "ka", "R0b", "kb"], [], hidden_layers=[64, 64], activation="tanh", scale_nn_outputs=True, **({"precision": a.precision} if a.precision else {}))
    cols = make_synth_expo2pool(a.nbatches * B, 1)
    X = np.stack([cols[f"x{i}"] for i in range(8)]); F = [cols["T"]]; Y = [cols["Resp_obs"]]
    flop, byts = 29184, 40
elif a.config == "c5":          # configs[4]: 1e7 samples, 32 covariates, 3 forcings, MLP [32,128,128,6] + the three-forcing RbQ10-family model, bf16 fwd / fp32 accumulate
    B = a.batch or 65536
    prec = a.precision or "bf16_fwd"
    model = eh.constructHybridModel([f"x{i}" for i in range(32)], ["ta", "sw_in", "vpd"], ["R_soil"], eh.Rs_components3F, dict(RS6_PARAMS),
                                    list(RS6_PARAMS), [], hidden_layers=[128, 128], activation="tanh", scale_nn_outputs=True, precision=prec)
    a.nbatches = max(1, (a.n or 10_000_000) // B)
    cols = make_synth_fluxnet32_3f(a.nbatches * B, 1)
    X = np.stack([cols[f"x{i}"] for i in range(32)]); F = [cols["ta"], cols["sw_in"], cols["vpd"]]; Y = [cols["R_soil"]]
    del cols
    flop, byts = 6 * (32 * 128 + 128 * 128 + 128 * 6), 4 * 36
    a.fused = 0                   # the row-split kernels have no fused-update mode
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
extra = {}
if a.config == "c5":
    # the bf16 modes run every product on the bf16 MFMA: priced against its dense peak (2 516 TFLOP/s, MI355X_MICROARCH.md); f32 against the fp32 one
    extra = {"precision": prec, "resident_samples": a.nbatches * B}
peak, pname = (2516.6, "frac_bf16_peak") if (a.config == "c5" and prec != "f32") else (157.3, "frac_f32_peak")
if a.epoch:
    for name, shuffle in (("contiguous", False), ("shuffled", True)):
        for k in range(2):
            eng.train_epoch(B, seed=11 + k, shuffle=shuffle, want_loss=False)
        eng.synchronize()
        t0 = time.perf_counter()
        for k in range(a.epoch):
            eng.train_epoch(B, seed=21 + k, shuffle=shuffle, want_loss=False)
        eng.synchronize()
        extra["epoch_us_per_step_" + name] = 1e6 * (time.perf_counter() - t0) / (a.epoch * a.nbatches)
    extra["shuffled_over_contiguous"] = extra["epoch_us_per_step_shuffled"] / extra["epoch_us_per_step_contiguous"]
print(json.dumps({**extra, "config": a.config, "batch": B, "fused": a.fused, "specialize": a.specialize, "us_per_step": us, "samples_per_s": B / us * 1e6,
                  "algorithmic_TFLOPs": flop * B / us / 1e6, pname: flop * B / us / 1e6 / peak,
                  "algorithmic_GBps": byts * B / us / 1e3, "final_loss": eng.train_step(0, B)}))
eng.close()
