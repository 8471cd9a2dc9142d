"""Throughput of the layer-wise form on the reference's GPU tutorial network (docs/literate/tutorials/synthetic_respiration_gpu.jl:79-92:
RbQ10, hidden_layers = [1024, 512, 256, 128, 64], sigmoid, scale_nn_outputs, input_batchnorm, RMSProp):  python tools/bench_lform.py [batch ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

HIDDEN = [1024, 512, 256, 128, 64]
DIMS = [2] + HIDDEN + [1]
FLOP = 6 * sum(a * b for a, b in zip(DIMS[:-1], DIMS[1:]))      # per sample: forward, delta and weight-gradient products


def measure(B, device=0):
    """-> dict (one JSON line of this tool): steady-state time of eh_train_step at minibatch B.  Needs a GPU."""
    import easyhybrid_jl_amd as eh
    from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=HIDDEN, activation="sigmoid", scale_nn_outputs=True, input_batchnorm=True)
    nb = max(2, min(8, (1 << 19) // B))
    cols = make_synth_rbq10(nb * B, seed=1)
    X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
    eng = model.engine(device)
    try:
        eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
        eng.set_params(model.initialparameters(1)); eng.opt_init("RMSProp", 0.001)
        steps = max(10, min(300, (1 << 22) // B))
        for s in range(5): eng.train_step((s % nb) * B, B, want_loss=False)
        eng.synchronize()
        t0 = time.perf_counter()
        for s in range(steps): eng.train_step((s % nb) * B, B, want_loss=False)
        eng.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / steps
        return {"net": DIMS, "batch": B, "us_per_step": round(us, 1), "samples_per_s": round(B / us * 1e6), "algorithmic_TFLOPs": round(FLOP * B / us / 1e6, 2),
                "frac_f32_peak": round(FLOP * B / us / 1e6 / 157.3, 3), "steps": steps, "final_loss": eng.train_step(0, B)}
    finally:
        eng.close()


if __name__ == "__main__":
    for B in [int(v) for v in sys.argv[1:]] or [64, 256, 1024, 4096, 16384, 65536]:
        print(json.dumps(measure(B)), flush=True)
