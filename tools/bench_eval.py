"""eh_eval / eh_forward over a whole split: time per call by number of workgroups ("eval_blocks" option).
  python tools/bench_eval.py [--config c2|c3] [--n SAMPLES]
One line per setting: metrics only (what evaluate_epoch costs) and with the predictions written back and copied to the host."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--n", type=int, default=3355443)
    ap.add_argument("--blocks", default="256,512,768,1024,1536,2048,0")
    a = ap.parse_args()
    import easyhybrid_jl_amd as eh
    from oracle import hybrid_oracle as ho
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    import util
    if a.config == "c2":
        spec = ho.rbq10_spec((16, 16), "tanh", True)
        X, f, y = ho.make_synth_rbq10(a.n, 42, 0.0)
        flop = 608
    else:
        spec = ho.expo2pool_spec((64, 64), "tanh", True) if hasattr(ho, "expo2pool_spec") else None
        rng = np.random.default_rng(0)
        mm = ho.MECH[spec.mech][0]
        X = rng.standard_normal((spec.n_pred, a.n)).astype(np.float32)
        f = {k: rng.standard_normal(a.n).astype(np.float32) for k in mm.forcings}
        y = {k: rng.standard_normal(a.n).astype(np.float32) for k in spec.targets}
        flop = 2 * (8 * 64 + 64 * 64 + 64 * 4)
    theta = ho.init_theta(spec, 1, np.float32)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_option("aot_spec", 1)
    eng.set_option("specialize", 1)
    nbytes = 4 * (X.shape[0] + len(f) + len(y))
    for b in [int(v) for v in a.blocks.split(",")]:
        eng.set_option("eval_blocks", b)
        for _ in range(3):
            eng.eval(0)
        t0 = time.perf_counter()
        for _ in range(20):
            eng.eval(0)
        t_m = (time.perf_counter() - t0) / 20
        eng.forward(0)
        t0 = time.perf_counter()
        for _ in range(5):
            eng.forward(0, params=False)
        t_f = (time.perf_counter() - t0) / 5
        print(json.dumps({"config": a.config, "samples": a.n, "eval_blocks": b, "eval_ms": 1e3 * t_m, "eval_GBps": nbytes * a.n / t_m / 1e9,
                          "eval_TFLOPs": flop * a.n / t_m / 1e12, "forward_to_host_ms": 1e3 * t_f}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
