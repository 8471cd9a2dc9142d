"""Parity of one (precision, variant) of the config-5 kernels against the oracle, outside the test suite (variant A/Bs):
  python tools/check_c5_variant.py bf16 7 [specialize]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import hybrid_oracle as ho
from tests import util
prec, variant = sys.argv[1], int(sys.argv[2])
spec_on = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for B in (33, 1000, 4097):
    spec = ho.c5_spec((128, 128), "tanh", prec, 32)
    X, f, y = ho.make_synth_c5(B, 11, 0.1, 32)
    theta = ho.init_theta(spec, 3, np.float32)
    eng = util.load_engine(spec, theta, X, f, y)
    eng.set_option("variant", variant)
    eng.set_option("specialize", spec_on)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
    m, _ = eng.eval(0)
    print(f"{prec} variant {variant} specialize {spec_on} B {B}: loss rel {abs(loss - l0) / abs(l0):.2e} grad relerr {util.relerr(grad, g0):.2e} n {nv} {sum(nv0)} eval mse rel {abs(m[0]['mse'] - l0) / l0:.2e}")
    eng.close()
