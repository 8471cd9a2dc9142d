"""Diagnostics for the confined fast-path bug (P <= 4 weight-gradient path + ReLU on shapes wider than one block):
build with `make -C easyhybrid.jl_amd/csrc EXP=-DEH_PS_WIDE` (the wide P <= 4 kernels are not in the normal build), then
EH_DEBUG_PS_ALL=1 python tools/ps_relu_repro.py   (never part of the test suite)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import hybrid_oracle as ho
from tests import util
for hidden in ((48, 33, 16), (61, 20, 22), (64, 64)):
    for act in ("tanh", "relu"):
        spec, theta, X, f, y = util.rbq10_case(1000, act, True, 0.1, hidden=hidden)
        eng = util.load_engine(spec, theta, X, f, y)
        loss, grad, nv = eng.loss_and_grad()
        l0, g0, nv0 = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
        print("CASE", hidden, act, f"loss err {abs(loss - l0) / abs(l0):.1e} grad err {util.relerr(grad, g0):.1e}", flush=True)
        eng.close()
