"""ctypes wrapper of oracle/eh_oracle.c (plain-C CPU restatement)  --  TEST INFRASTRUCTURE ONLY.
Used by tests and by bench.py's `cpu_baseline` leg; never by the product path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import hybrid_oracle as ho

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libeh_oracle.so")
SRC = os.path.join(HERE, "eh_oracle.c")
SRC_FAST = os.path.join(HERE, "eh_oracle_fast.c")

ACT = {"tanh": 0, "sigmoid": 1, "relu": 2, "swish": 3, "identity": 4}
MECH = {"rbq10": 0, "expo": 1, "linear": 2, "expo2pool": 3, "rs_components": 4}     # single-output models only


class Spec(C.Structure):
    _fields_ = [("P", C.c_int), ("NL", C.c_int), ("hidden", C.c_int * 4), ("K", C.c_int), ("G", C.c_int),
                ("act", C.c_int), ("scale_nn", C.c_int), ("mech", C.c_int), ("n_par", C.c_int),
                ("par_kind", C.c_int * 8), ("par_idx", C.c_int * 8),
                ("lo", C.c_float * 8), ("hi", C.c_float * 8), ("def_", C.c_float * 8),
                ("F", C.c_int), ("forc_col", C.c_int * 4), ("T", C.c_int), ("targ_out", C.c_int * 4)]


def _gpu_is_up() -> bool:
    import sys
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


def build(force: bool = False) -> str:
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(SRC_FAST)):
        if _gpu_is_up():      # compiling means fork + exec, which a process that has initialised the GPU must not do on the GPU boxes
            raise RuntimeError("oracle/libeh_oracle.so is missing or stale and this process has already initialised the GPU: build it first "
                               "(python -c 'from oracle import c_oracle; c_oracle.build()', __graft_entry__.build(), or the start of a pytest session)")
        # the checker (eh_oracle.c) with strict fp32 semantics; the timed blocked form (eh_oracle_fast.c) with the vector math library
        obj = os.path.join(HERE, "eh_oracle_fast.o")
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-ffast-math", "-fPIC", "-c", SRC_FAST, "-o", obj])
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", SRC, obj, "-o", SO, "-lm"])
        os.unlink(obj)
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.eho_n_theta.restype = C.c_long
        _lib.eho_loss_and_grad.restype = C.c_float
        _lib.eho_train_steps.restype = C.c_float
        _lib.eho_loss_and_grad_fast.restype = C.c_float
        _lib.eho_train_steps_fast.restype = C.c_float
    return _lib


def to_c(spec: ho.HybridSpec) -> Spec:
    mm = ho.MECH[spec.mech][0]
    s = Spec()
    s.P, s.NL, s.K, s.G = spec.n_pred, len(spec.hidden), len(spec.neural), len(spec.glob)
    for i, w in enumerate(spec.hidden):
        s.hidden[i] = w
    s.act, s.scale_nn, s.mech, s.n_par = ACT[spec.activation], int(spec.scale_nn_outputs), MECH[spec.mech], len(mm.params)
    for j, p in enumerate(mm.params):
        if p in spec.neural:
            s.par_kind[j], s.par_idx[j] = 0, spec.neural.index(p)
        elif p in spec.glob:
            s.par_kind[j], s.par_idx[j] = 1, spec.glob.index(p)
        else:
            s.par_kind[j], s.par_idx[j] = 2, 0
        s.def_[j], s.lo[j], s.hi[j] = spec.parameters[p]
    s.F = len(mm.forcings)
    for f in range(4):
        s.forc_col[f] = f if f < s.F else -1
    s.T = len(spec.targets)
    for t, name in enumerate(spec.targets):
        s.targ_out[t] = mm.outputs.index(name)
    return s


def _ptrs(arrs):
    return (C.c_void_p * max(1, len(arrs)))(*[a.ctypes.data for a in arrs])


def _pack(spec, X, forcings, targets):
    mm = ho.MECH[spec.mech][0]
    Xf = np.asfortranarray(np.asarray(X, np.float32))
    fs = [np.ascontiguousarray(forcings[f], np.float32) for f in mm.forcings]
    ts = [np.ascontiguousarray(targets[t], np.float32) for t in spec.targets]
    return Xf, fs, ts


def loss_and_grad(spec, theta, X, forcings, targets, nthreads=1, fast=False):
    """fast: the blocked form bench.py times (eh_oracle_fast.c: 16 samples per SIMD block); else the checker"""
    s = to_c(spec)
    Xf, fs, ts = _pack(spec, X, forcings, targets)
    theta = np.ascontiguousarray(theta, np.float32)
    grad = np.zeros(theta.size, np.float32)
    nv = (C.c_long * 4)()
    fn = lib().eho_loss_and_grad_fast if fast else lib().eho_loss_and_grad
    loss = fn(C.byref(s), C.c_void_p(theta.ctypes.data), C.c_void_p(Xf.ctypes.data), _ptrs(fs), _ptrs(ts),
                                   C.c_long(Xf.shape[1]), C.c_void_p(grad.ctypes.data), nv, C.c_int(nthreads))
    return float(loss), grad, [int(nv[t]) for t in range(len(spec.targets))]


def train_steps(spec, theta, X, forcings, targets, batch, nsteps, lr=0.01, nthreads=1, fast=False):
    """Adam steps over contiguous batches; returns (theta, last loss).  Timed by bench.py (fast = True: the blocked form)."""
    s = to_c(spec)
    Xf, fs, ts = _pack(spec, X, forcings, targets)
    theta = np.ascontiguousarray(theta, np.float32).copy()
    m = np.zeros_like(theta); v = np.zeros_like(theta); bt = np.array([0.9, 0.999], np.float32)
    fn = lib().eho_train_steps_fast if fast else lib().eho_train_steps
    loss = fn(C.byref(s), C.c_void_p(theta.ctypes.data), C.c_void_p(m.ctypes.data), C.c_void_p(v.ctypes.data),
                                 C.c_void_p(bt.ctypes.data), C.c_void_p(Xf.ctypes.data), _ptrs(fs), _ptrs(ts), C.c_long(Xf.shape[1]),
                                 C.c_long(batch), C.c_long(nsteps), C.c_float(lr), C.c_int(nthreads))
    return theta, float(loss)
