"""-m gpu parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Tolerance: BASELINE.json north_star = 1e-5 relative (fp32); the oracle is
evaluated in fp64 so its own rounding does not eat the budget."""
import numpy as np
import pytest

from oracle import hybrid_oracle as ho
from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("act", ["tanh", "sigmoid", "relu", "swish"])
@pytest.mark.parametrize("scale", [False, True])
@pytest.mark.parametrize("B,nan_frac", [(12, 0.0), (64, 0.2), (1024, 0.2), (1000, 0.05)])
def test_rbq10_loss_and_grad(act, scale, B, nan_frac):
    spec, theta, X, f, y = util.rbq10_case(B, act, scale, nan_frac)
    eng = util.load_engine(spec, theta, X, f, y)
    loss, grad, nv = eng.loss_and_grad()
    l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
    assert nv == sum(nv0)
    assert abs(loss - l0) <= TOL * abs(l0)
    assert util.relerr(grad, g0) <= TOL
    eng.close()


def test_forward_matches_oracle():
    spec, theta, X, f, y = util.rbq10_case(300, "tanh", True, 0.1)
    eng = util.load_engine(spec, theta, X, f, y)
    out = eng.forward(0)
    ref = ho.forward(spec, theta.astype(np.float64), X, f)
    assert util.relerr(out["reco"], ref["reco"]) <= TOL
    assert util.relerr(out["parameters"]["rb"], ref["parameters"]["rb"]) <= TOL
    assert util.relerr(out["parameters"]["Q10"], np.broadcast_to(ref["parameters"]["Q10"], (300,))) <= TOL
    eng.close()
