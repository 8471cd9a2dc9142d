"""The P <= 4 weight-gradient path (FAST bit 1) on shapes wider than one block, against the oracle per parameter block, for
fast_paths = 3 (K1 | PS) and 1 (K1 only).  Round 1's fuzz found ReLU kernels of the 64-wide three-layer shape whose loss sum came
out as exactly 0 (gradients right); round 2 traced it to the SLP vectoriser (-O3 packs the loss sum and the valid count into one
<2 x float> accumulator): the normal build now carries -fno-slp-vectorize and passes.  To see the failure again build the two
translation units with the vectoriser on and point the loader at that library:
    tools/ps_variants.sh slp "-fslp-vectorize"
    EASYHYBRID_HIP_LIB=dbg/lib_slp.so python tools/ps_relu_repro.py        (never part of the test suite)
tools/ps_probe.py prints the raw batch sums [S, n, Sy, Syy] of the same kernels, tools/ps_probe2.py (variant built with
-DEH_DBG_LACC) the per-wave loss accumulators -- with that extra store the miscompile disappears as well."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import hybrid_oracle as ho
from tests import util

def blocks(spec):
    out, off, inw = [], 0, spec.n_pred
    for l, w in enumerate(list(spec.hidden) + [len(spec.neural)]):
        out.append((f"W{l}", off, off + w * inw)); off += w * inw
        out.append((f"b{l}", off, off + w)); off += w
        inw = w
    out.append(("glob", off, off + len(spec.glob)))
    return out

bad = 0
for hidden in ((48, 33, 16), (61, 20, 22), (64, 64), (40, 40)):
    for act in ("tanh", "relu", "sigmoid", "swish"):
        for B in (1000, 16, 4096):
            spec, theta, X, f, y = util.rbq10_case(B, act, True, 0.1, hidden=hidden)
            l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
            for fp in (3, 1):
                eng = util.load_engine(spec, theta, X, f, y)
                eng.set_option("fast_paths", fp)
                loss, grad, nv = eng.loss_and_grad()
                le, ge = abs(loss - l0) / abs(l0), util.relerr(grad, g0)
                flag = "" if (le < 1e-5 and ge < 1e-5) else "  <-- WRONG"
                bad += bool(flag)
                det = " ".join(f"{n}:{util.relerr(grad[a:b], g0[a:b]):.0e}" for n, a, b in blocks(spec) if b > a) if flag else ""
                print("CASE", hidden, act, B, f"fast={fp} loss err {le:.1e} grad err {ge:.1e}{flag} {det}", flush=True)
                eng.close()
print("WRONG CASES:", bad)
