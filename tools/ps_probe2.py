"""Diagnostic build -DEH_DBG_LACC: per-wave value of the loss accumulator before / after its wave sum, next to the raw batch
sums, on the failing kernel of tools/ps_relu_repro.py."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import hybrid_oracle as ho
from tests import util
from easyhybrid_jl_amd import _lib as L
from easyhybrid_jl_amd.dp import _DevArray
for act in ("relu", "tanh"):
    for B in (16, 64, 1000):
        spec, theta, X, f, y = util.rbq10_case(B, act, True, 0.1, hidden=(48, 33, 16))
        l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
        for fp in (3, 1):
            eng = util.load_engine(spec, theta, X, f, y)
            eng.set_option("fast_paths", fp); eng.set_option("max_blocks", 1)
            buf = (C.c_uint64 * 32)()
            eng._lib.eh_debug_stamps(eng._h, buf, 32)
            eng.dp_grad(0, B); eng.synchronize()
            eng._lib.eh_debug_stamps(eng._h, buf, 32)
            fl = np.frombuffer(bytes(buf), np.float32)
            p, n = eng.device_buffer(L.EH_BUF_GRAD)
            raw = torch.as_tensor(_DevArray(p, n), device="cuda").cpu().numpy()
            nt = spec.n_theta
            print(act, B, f"fast={fp} S = {raw[nt]:.5f} (oracle {l0 * raw[nt + 1]:.5f}); lacc lanes 0/1 per wave before the wave sum {fl[:8]}, wave sums {fl[16:20]}", flush=True)
            eng.close()
