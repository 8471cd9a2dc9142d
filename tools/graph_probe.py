"""Does replaying the step sequence as a hipGraph shorten the gap between the dependent step kernels?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
B, NB = 65536, 64
spec, theta, X, f, y = util.rbq10_case(NB * B, "tanh", True, 0.0)
res = {}
for mode in ("stream", "graph"):
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01); eng.set_option("fused_update", 1)
    for i in range(192): eng.train_step((i % NB) * B, B, want_loss=False)
    if mode == "graph":                                   # (no synchronize: a fused-mode graph starts with an update pending)
        eng.graph_begin()
        for i in range(192): eng.train_step((i % NB) * B, B, want_loss=False)
        g = eng.graph_end()
        eng.graph_launch(g)
    else:
        for i in range(192): eng.train_step((i % NB) * B, B, want_loss=False)
    t0 = time.perf_counter()
    for rep in range(20):
        if mode == "graph":
            eng.graph_launch(g)
        else:
            for i in range(192): eng.train_step((i % NB) * B, B, want_loss=False)
    eng.synchronize()
    dt = time.perf_counter() - t0
    res[mode] = eng.get_params()
    print(f"{mode}: {1e6 * dt / (20 * 192):.2f} us/step", flush=True)
    eng.close()
print("max |theta_graph - theta_stream| =", float(np.max(np.abs(res["graph"] - res["stream"]))))
