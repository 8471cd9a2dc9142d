"""The path train() runs: eh_train_epoch(shuffle) over the headline data set (64 resident batches of 65 536 RbQ10 samples) -- every
step gathers its 16-byte records through the epoch's device-side permutation (reference: MLUtils.DataLoader(shuffle = true),
src/data/loaders.jl:1-12, consumed by run_epoch!, src/training/epoch.jl:13-33) -- beside the contiguous epoch.
    python tools/bench_epoch.py [--epochs 8] [--mode both|shuffled|contiguous]        one JSON line"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B, NB = 65536, 64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=8)
    ap.add_argument("--mode", default="both", choices=["both", "shuffled", "contiguous"])
    a = ap.parse_args()
    import easyhybrid_jl_amd as eh
    from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    cols = make_synth_rbq10(NB * B, seed=42)
    X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
    eng = model.engine(0)
    eng.set_data(eh.EH_SPLIT_TRAIN, X, [cols["ta"]], [cols["reco"]])
    eng.set_params(model.initialparameters(161803))
    eng.opt_init("Adam", 0.01)
    eng.set_option("fused_update", 1)
    eng.set_option("specialize", 1)
    eng.forward(eh.EH_SPLIT_TRAIN, 0, 1, params=False)
    out = {}
    for name, shuffle in (("contiguous", False), ("shuffled", True)):
        if a.mode not in ("both", name):
            continue
        for k in range(2):
            eng.train_epoch(B, seed=11 + k, shuffle=shuffle, want_loss=False)
        eng.synchronize()
        t0 = time.perf_counter()
        for k in range(a.epochs):
            eng.train_epoch(B, seed=100 + k, shuffle=shuffle, want_loss=False)
        eng.synchronize()
        per = (time.perf_counter() - t0) / (a.epochs * NB)
        out[name] = {"us_per_step": 1e6 * per, "samples_per_s": B / per, "algorithmic_GBps": 16 * B / per / 1e9}
    if len(out) == 2:
        out["shuffled_over_contiguous"] = out["shuffled"]["us_per_step"] / out["contiguous"]["us_per_step"]
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
