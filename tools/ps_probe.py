"""Raw batch sums [S, n, Sy, Syy] of the P <= 4 path (fast_paths 3) against the K1-only kernels (fast_paths 1) and the oracle's
loss, for the failing shapes of tools/ps_relu_repro.py (same libraries: EH_DEBUG_PS_ALL=1 EASYHYBRID_HIP_LIB=dbg/lib_<v>.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import hybrid_oracle as ho
from tests import util
from easyhybrid_jl_amd import _lib as L
from easyhybrid_jl_amd.dp import _DevArray
for hidden in ((48, 33, 16), (61, 20, 22), (64, 64)):
    for act in ("relu", "tanh"):
        for B, mb in ((16, 256), (64, 256), (1000, 1), (1000, 256)):
            spec, theta, X, f, y = util.rbq10_case(B, act, True, 0.1, hidden=hidden)
            l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
            for fp in (3, 1):
                eng = util.load_engine(spec, theta, X, f, y)
                eng.set_option("fast_paths", fp); eng.set_option("max_blocks", mb)
                eng.dp_grad(0, B); eng.synchronize()
                p, n = eng.device_buffer(L.EH_BUF_GRAD)
                raw = torch.as_tensor(_DevArray(p, n), device="cuda").cpu().numpy()
                nt = spec.n_theta
                print(hidden, act, B, f"blocks<={mb} fast={fp} tail [S n Sy Syy] = {raw[nt:nt + 4]}  S/n = {raw[nt] / raw[nt + 1]:.6f} oracle loss {l0:.6f}", flush=True)
                eng.close()
