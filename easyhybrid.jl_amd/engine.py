"""Thin object wrapper over the C ABI (one `HybridEngine` = one `eh_handle` = one GPU).

All numerics run in the HIP library; this class only marshals NumPy arrays and converts status
codes into the exception types the reference raises (ArgumentError -> ValueError, unknown model
pieces -> NotImplementedError, call-order / device problems -> RuntimeError).
"""
from __future__ import annotations

import ctypes as C
import numbers
import os
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib as L

_F = C.POINTER(C.c_float)


class EngineError(RuntimeError):
    pass


def _raise(status: int, msg: str):
    if status == L.EH_EINVAL:
        raise ValueError(msg)
    if status == L.EH_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if status == L.EH_ENOMEM:
        raise MemoryError(msg)
    raise EngineError(msg)


def _fptr(a: Optional[np.ndarray]):
    return a.ctypes.data_as(_F) if a is not None else None


class PerTarget:
    """loss_spec = PerTarget((l_1, ..., l_T)) of the reference (src/losses/compute_loss.jl:128-145): target t is trained on its own
    loss; the device builds mse / mae / nseLoss per target."""

    def __init__(self, losses):
        self.losses = tuple(losses)


class _LazySums(dict):
    """split -> sums over that split's host arrays, computed at the first read (the arrays are referenced until then or until the split is replaced)"""

    def __init__(self, fn):
        super().__init__()
        self._fn, self._src = fn, {}

    def put(self, key, src):
        self._src[key] = src
        dict.pop(self, key, None)

    def _materialise(self, key):
        if key in self._src:
            dict.__setitem__(self, key, self._fn(*self._src.pop(key)))

    def get(self, key, default=None):
        self._materialise(key)
        return dict.get(self, key, default)

    def __getitem__(self, key):
        self._materialise(key)
        return dict.__getitem__(self, key)

    def __contains__(self, key):
        return key in self._src or dict.__contains__(self, key)


class HybridEngine:
    """Device-resident hybrid model: parameters, optimiser state and datasets live in HBM."""

    def __init__(self, desc: L.ModelDesc, n_par: int, target_names: Sequence[str], param_names: Sequence[str], n_pseudo: int = 0):
        self._lib = L.lib()
        self._h = C.c_void_p()
        self.desc = desc
        st = self._lib.eh_create(C.byref(desc), C.byref(self._h))
        if st != L.EH_OK:
            self._h = C.c_void_p()
            _raise(st, self._lib.eh_last_error(None).decode())
        n = C.c_int64()
        self._chk(self._lib.eh_n_theta(self._h, C.byref(n)))
        self.n_theta = int(n.value)
        self.n_par = n_par
        self.target_names = list(target_names)
        # entries of an extra loss of the predictions ride on targets of their own behind the data targets (set_extra_entries): the
        # device sees len(target_names) + n_pseudo targets, the caller only ever the data targets
        self.n_pseudo = int(n_pseudo)
        self.n_targets_total = len(self.target_names) + self.n_pseudo
        self.extra_entries = []
        self.param_names = list(param_names)
        self.n_samples = {L.EH_SPLIT_TRAIN: 0, L.EH_SPLIT_VAL: 0}
        self.x_sum = _LazySums(lambda X, N: ((np.array([r.sum(dtype=np.float64) for r in X]) if isinstance(X, list) else X.sum(axis=1, dtype=np.float64)), N))
        self.y_sum = _LazySums(lambda ts: np.array([[np.nansum(t, dtype=np.float64), np.count_nonzero(~np.isnan(t))] for t in ts], np.float64))   # (sum, n valid) per target
        if os.environ.get("EH_MAX_BLOCKS"):               # several ranks sharing one GPU (tests): every kernel must fit beside the others
            self.set_option("max_blocks", int(os.environ["EH_MAX_BLOCKS"]))

    # -- plumbing --------------------------------------------------------------------------------
    def _chk(self, st: int):
        if st != L.EH_OK:
            _raise(st, self._lib.eh_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.eh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: int):
        self._chk(self._lib.eh_set_stream(self._h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._chk(self._lib.eh_synchronize(self._h))

    def set_option(self, name: str, value: int):
        self._chk(self._lib.eh_set_option(self._h, name.encode(), int(value)))

    # -- data ------------------------------------------------------------------------------------
    def set_data(self, split: int, X: np.ndarray, forcings: Sequence[np.ndarray], targets: Sequence[np.ndarray]):
        """X: (P, N) like the reference (features x samples) -- or a list of P arrays of (N,), the caller's own predictor columns, which go
        to the device without being stacked into a matrix first (EH_DATA_X_ROWS); forcings / targets: lists of (N,)."""
        rows = None
        if isinstance(X, (list, tuple)):
            rows = [np.ascontiguousarray(r, np.float32) for r in X]
            P, N = len(rows), (len(rows[0]) if rows else len(targets[0]))
            for r in rows:
                if r.shape != (N,):
                    raise ValueError("predictor rows must have one value per sample")
        else:
            X = np.asarray(X, np.float32)
            P, N = X.shape
        if P != self.desc.n_predictors:
            raise ValueError(f"X has {P} predictor rows, model expects {self.desc.n_predictors}")
        if len(forcings) != self.desc.n_forcings or len(targets) != len(self.target_names):
            raise ValueError("number of forcing / target arrays does not match the model")
        targets = list(targets) + [np.zeros(N, np.float32)] * self.n_pseudo      # an extra-loss entry "observes" every sample (no NaN)
        # sums only the data-parallel driver asks for (common BatchNorm shift, common target shift): taken when they are asked for --
        # two more passes over 4 M rows are a tenth of a short train() call
        self.x_sum.put(split, (X if rows is None else rows, N))
        self.y_sum.put(split, (list(targets),))
        xf = np.ascontiguousarray(X) if rows is None else None      # row-major (P, N): handed over as P planes (EH_DATA_X_PLANES) -- no transposed copy of the whole matrix
        fs = [np.ascontiguousarray(f, np.float32) for f in forcings]
        ts = [np.ascontiguousarray(t, np.float32) for t in targets]
        for a in fs + ts:
            if a.shape != (N,):
                raise ValueError("forcing / target arrays must have one value per sample")
        fp = (C.c_void_p * max(1, len(fs)))(*[a.ctypes.data for a in fs])
        tp = (C.c_void_p * max(1, len(ts)))(*[a.ctypes.data for a in ts])
        if rows is None:
            self._chk(self._lib.eh_set_data(self._h, split, N, C.c_void_p(xf.ctypes.data), fp, tp, 2))
        else:
            xp = (C.c_void_p * max(1, P))(*[r.ctypes.data for r in rows])
            self._chk(self._lib.eh_set_data(self._h, split, N, C.cast(xp, C.c_void_p), fp, tp, 4))
        self.n_samples[split] = N

    def set_data_device(self, split: int, n: int, x_ptr: int, forcing_ptrs: Sequence[int], target_ptrs: Sequence[int], planes: bool = False):
        """Same as set_data with pointers that already live on the handle's device; x as the reference holds it -- (P x N) column-major, N
        records of P -- or, planes = True, as P arrays of N (a row-major (P, N) tensor)."""
        fp = (C.c_void_p * max(1, len(forcing_ptrs)))(*forcing_ptrs)
        tp = (C.c_void_p * max(1, len(target_ptrs)))(*target_ptrs)
        self._chk(self._lib.eh_set_data(self._h, split, n, C.c_void_p(x_ptr), fp, tp, 1 | (2 if planes else 0)))
        self.n_samples[split] = n

    # -- parameters ------------------------------------------------------------------------------
    def set_params(self, theta: np.ndarray):
        theta = np.ascontiguousarray(theta, np.float32)
        self._chk(self._lib.eh_set_params(self._h, _fptr(theta), theta.size))

    def get_params(self) -> np.ndarray:
        out = np.empty(self.n_theta, np.float32)
        self._chk(self._lib.eh_get_params(self._h, _fptr(out), out.size))
        return out

    # -- forward / eval --------------------------------------------------------------------------
    def _outs(self, count, want_yhat, want_params):
        ys = [np.empty(count, np.float32) for _ in range(self.n_targets_total)] if want_yhat else None      # (zip with target_names below: the data targets)
        ps = [np.empty(count, np.float32) for _ in range(self.n_par)] if want_params else None
        yp = (_F * len(ys))(*[_fptr(a) for a in ys]) if ys else None
        pp = (_F * len(ps))(*[_fptr(a) for a in ps]) if ps else None
        return ys, ps, yp, pp

    def forward(self, split: int, first: int = 0, count: Optional[int] = None, params: bool = True):
        count = self.n_samples[split] - first if count is None else count
        ys, ps, yp, pp = self._outs(count, True, params)
        self._chk(self._lib.eh_forward(self._h, split, first, count, yp, pp))
        out = dict(zip(self.target_names, ys))
        if params:
            out["parameters"] = dict(zip(self.param_names, ps))
        return out

    def eval(self, split: int, first: int = 0, count: Optional[int] = None, predictions: bool = False):
        count = self.n_samples[split] - first if count is None else count
        m = (L.TargetMetrics * self.n_targets_total)()
        ys, _, yp, _ = self._outs(count, predictions, False)
        self._chk(self._lib.eh_eval(self._h, split, first, count, m, yp, None))
        metrics = [{f: getattr(m[t], f) for f, _ in L.TargetMetrics._fields_} for t in range(len(self.target_names))]
        return metrics, (dict(zip(self.target_names, ys)) if predictions else None)

    def loss_and_grad(self, split: int = L.EH_SPLIT_TRAIN, first: int = 0, count: Optional[int] = None, idx=None):
        loss = C.c_float()
        nv = C.c_int64()
        grad = np.empty(self.n_theta, np.float32)
        if idx is not None:
            idx = np.ascontiguousarray(idx, np.int32)
            ip, first, count = idx.ctypes.data_as(C.POINTER(C.c_int32)), 0, idx.size
        else:
            ip = None
            count = self.n_samples[split] - first if count is None else count
        self._chk(self._lib.eh_loss_and_grad(self._h, split, ip, first, count, C.byref(loss), _fptr(grad), C.byref(nv)))
        return float(loss.value), grad, int(nv.value)

    # -- optimiser / training --------------------------------------------------------------------
    def opt_init(self, rule: str = "Adam", lr: float = 0.01, beta1: float = 0.9, beta2: float = 0.999,
                 eps: float = 1e-8, weight_decay: float = 0.0):
        if rule not in L.OPT_RULES:
            raise NotImplementedError(f"optimiser rule {rule} is not implemented on the device")
        self._chk(self._lib.eh_opt_init(self._h, L.OPT_RULES[rule], lr, beta1, beta2, eps, weight_decay))

    def get_bn_state(self):
        """running (mean, var) of the input BatchNorm layer -- the `st.st_nn` part of the model state"""
        P = self.desc.n_predictors
        m = np.empty(P, np.float32); v = np.empty(P, np.float32)
        self._chk(self._lib.eh_get_bn_state(self._h, _fptr(m), _fptr(v), P))
        return m, v

    def set_bn_state(self, mean, var):
        m = np.ascontiguousarray(mean, np.float32); v = np.ascontiguousarray(var, np.float32)
        self._chk(self._lib.eh_set_bn_state(self._h, _fptr(m), _fptr(v), m.size))

    def set_training_loss(self, name):
        """TrainConfig.training_loss (src/config/TrainingConfig.jl:64): mse | rmse | mae | nseLoss (one pass) |
        pearsonLoss | kgeLoss | pbkgeLoss (forward passes for the batch moments first; not in fused_update mode, not data parallel;
        rmse joins them on multi-target models), or a function f(yhat, y) = mean of per-sample terms, which is recorded
        (program.trace_loss) and compiled into the step kernel at run time.  Applied to every target like the reference does
        (src/losses/compute_loss.jl:115-126); PerTarget((l_1, ..., l_T)) / a list gives each target its own (:128-145)."""
        self._loss_spec = name
        per_target = isinstance(name, PerTarget)
        if per_target:
            name = list(name.losses)
        if self.n_pseudo and not (per_target or (isinstance(name, (list, tuple)) and not (name and callable(name[0])))):
            name, per_target = [name] * len(self.target_names), True      # (extra-loss entries make every model a multi-target one: the per-target path)
        if not per_target and isinstance(name, (list, tuple)) and name and callable(name[0]):
            # (f, args) / (f, kwargs) / (f, args, kwargs)  (src/losses/loss_fn.jl:92-107): f(yhat, y, args...; kwargs...)
            f, rest = name[0], list(name[1:])
            args = next((tuple(r) for r in rest if isinstance(r, (tuple, list))), ())
            kwargs = next((dict(r) for r in rest if isinstance(r, dict)), {})
            name = (lambda yh, y, _f=f, _a=args, _k=kwargs: _f(yh, y, *_a, **_k))
        if isinstance(name, (list, tuple)):                      # PerTarget: one loss per target (compute_loss.jl:128-145)
            if len(name) != len(self.target_names):
                raise AssertionError("Length of targets and PerTarget losses tuple must match")
            def bind(n):                                         # (f, args) / (f, kwargs) / (f, args, kwargs) per target, loss_fn.jl:92-107
                if isinstance(n, (list, tuple)) and n and callable(n[0]):
                    f_, rest = n[0], list(n[1:])
                    a_ = next((tuple(r) for r in rest if isinstance(r, (tuple, list))), ())
                    k_ = next((dict(r) for r in rest if isinstance(r, dict)), {})
                    return lambda yh, y, _f=f_, _a=a_, _k=k_: _f(yh, y, *_a, **_k)
                return n
            name = [bind(n) for n in name]
            for n in name:
                if not callable(n) and n not in L.TRAINING_LOSSES:
                    raise NotImplementedError(f"training loss {n!r} is not implemented on the device (have {sorted(L.TRAINING_LOSSES)})")
            for t, n in enumerate(name):
                if callable(n):
                    self._set_loss_program(n, target=t)          # every function its own program
            self._put_extra_programs()                           # (a kind EH_LOSS_PROGRAM needs its program in place)
            codes = [L.EH_LOSS_PROGRAM if callable(n) else L.TRAINING_LOSSES[n] for n in name] + [L.EH_LOSS_PROGRAM] * self.n_pseudo
            kinds = (C.c_int32 * len(codes))(*codes)
            self._chk(self._lib.eh_set_target_losses(self._h, kinds, len(codes)))
            self._loss_kinds = ["program" if callable(n) else n for n in name]
            return
        if callable(name):
            self._set_loss_program(name)
            self.set_option("training_loss", L.EH_LOSS_PROGRAM)
            self._loss_kinds = ["program"] * len(self.target_names)
            return
        if name not in L.TRAINING_LOSSES:
            raise NotImplementedError(f"training loss {name!r} is not implemented in the fused kernel (have {sorted(L.TRAINING_LOSSES)})")
        self.set_option("training_loss", L.TRAINING_LOSSES[name])
        self._loss_kinds = [name] * len(self.target_names)

    def _set_loss_program(self, fn, target=None):
        """record f(yhat, y) = mean of per-sample terms (program.trace_loss) and hand it to the library: for every target
        (eh_set_loss_program) or for one (eh_set_target_loss_program)"""
        from .program import trace_loss
        pg = trace_loss(fn)
        words = (C.c_uint32 * len(pg.code))(*pg.words())
        consts = (C.c_float * max(1, len(pg.consts)))(*pg.consts)
        if target is None:
            self._chk(self._lib.eh_set_loss_program(self._h, words, len(pg.code), consts, len(pg.consts), pg.out[0]))
        else:
            self._chk(self._lib.eh_set_target_loss_program(self._h, int(target), words, len(pg.code), consts, len(pg.consts), pg.out[0]))

    def set_extra_entries(self, entries):
        """entries of an extra loss that is a function of the predictions (program.trace_extra_loss: [(name, output, "sum" | "mean",
        Program)]): entry i rides on device target len(target_names) + i (include/easyhybrid_hip.h: eh_set_target_roles)"""
        if len(entries) != self.n_pseudo:
            raise ValueError("the engine was created for a different number of extra-loss entries")
        self.extra_entries = list(entries)
        roles = [0] * len(self.target_names) + [2 if e[2] == "sum" else 1 for e in entries]
        self._chk(self._lib.eh_set_target_roles(self._h, (C.c_int32 * len(roles))(*roles), len(roles)))
        self.set_training_loss(getattr(self, "_loss_spec", "mse"))

    def _put_extra_programs(self):
        for i, (_, _, _, pg) in enumerate(self.extra_entries):
            words = (C.c_uint32 * len(pg.code))(*pg.words())
            consts = (C.c_float * max(1, len(pg.consts)))(*pg.consts)
            self._chk(self._lib.eh_set_target_loss_program(self._h, len(self.target_names) + i, words, len(pg.code), consts, len(pg.consts), pg.out[0]))

    def set_weight_l2(self, lam: float, normalize: bool = False):
        """extra_loss = lam * weight_l2(ps; normalize) (src/utils/extract_weights.jl:69-91); lam = 0 switches it off"""
        self._chk(self._lib.eh_set_weight_l2(self._h, float(lam), int(bool(normalize))))

    def set_agg(self, agg: str = "sum", n_extra_terms: int = 0):
        """`agg` of the training configuration (TrainingConfig.jl:76-77): the training loss is agg([agg(per-target losses), extra loss
        entries...]) (compute_loss.jl:31-34,50-53); "sum" or "mean".  n_extra_terms: the entries the extra loss returns (mean only)."""
        if agg not in ("sum", "mean"):
            raise NotImplementedError(f"agg {agg!r}: sum or mean")
        self.set_option("extra_terms", int(n_extra_terms))
        self.set_option("agg", 1 if agg == "mean" else 0)

    def set_weight_l2_coef(self, coef):
        """extra loss = sum_i coef[i] * theta_i^2 -- several weight_l2 terms folded into one coefficient per flat-theta entry
        (HybridModel.l2_coefficients); None / all zero switches it off"""
        if coef is None:
            self._chk(self._lib.eh_set_weight_l2_coef(self._h, None, 0))
            return
        c = np.ascontiguousarray(coef, np.float32)
        self._chk(self._lib.eh_set_weight_l2_coef(self._h, c.ctypes.data_as(C.POINTER(C.c_float)), c.size))

    def get_opt_state(self):
        m = np.empty(self.n_theta, np.float32)
        v = np.empty(self.n_theta, np.float32)
        bt = np.empty(2, np.float32)
        self._chk(self._lib.eh_get_opt_state(self._h, _fptr(m), _fptr(v), m.size, _fptr(bt)))
        return m, v, bt

    def set_opt_state(self, m, v, bt):
        m = np.ascontiguousarray(m, np.float32); v = np.ascontiguousarray(v, np.float32); bt = np.ascontiguousarray(bt, np.float32)
        self._chk(self._lib.eh_set_opt_state(self._h, _fptr(m), _fptr(v), m.size, _fptr(bt)))

    def train_step(self, first: int, count: int, want_loss: bool = True, idx=None):
        """One training step on train samples [first, first+count), or -- idx given -- on the minibatch idx[first : first+count]
        the caller's own loader drew: a host int32 array, or a device pointer (int; e.g. `tensor.data_ptr()` of a whole epoch's
        permutation) that must stay alive until the step has run."""
        loss = C.c_float()
        ip, on_dev = None, 0
        if isinstance(idx, numbers.Integral):       # (also numpy integers: np.int64(tensor.data_ptr()) is a pointer, not a one-element index list)
            ip, on_dev = C.cast(C.c_void_p(int(idx)), C.POINTER(C.c_int32)), 1
        elif idx is not None:
            self._idx_keep = np.ascontiguousarray(idx, np.int32)        # (kept until the next call: the copy is asynchronous)
            ip = self._idx_keep.ctypes.data_as(C.POINTER(C.c_int32))
        self._chk(self._lib.eh_train_step(self._h, ip, on_dev, first, count, C.byref(loss) if want_loss else None))
        return float(loss.value) if want_loss else None

    # -- hipGraph capture of a step sequence (see include/easyhybrid_hip.h) --------------------------
    def graph_begin(self):
        self._chk(self._lib.eh_graph_begin(self._h))

    def graph_end(self) -> int:
        g = C.c_int32()
        self._chk(self._lib.eh_graph_end(self._h, C.byref(g)))
        return int(g.value)

    def graph_launch(self, graph_id: int):
        self._chk(self._lib.eh_graph_launch(self._h, graph_id))

    def train_epoch(self, batchsize: int, seed: int = 0, shuffle: bool = True, want_loss: bool = True):
        loss = C.c_float()
        ns = C.c_int64()
        self._chk(self._lib.eh_train_epoch(self._h, batchsize, seed & (2**64 - 1), int(shuffle),
                                           C.byref(loss) if want_loss else None, C.byref(ns)))
        return (float(loss.value) if want_loss else None), int(ns.value)

    # -- the mechanistic stage on its own (NN outside the library) ----------------------------------
    def mech_loss_vjp(self, count: int, o_ptr: int, forcing_ptrs: Sequence[int], target_ptrs: Sequence[int], d_o_ptr: int,
                      yhat_ptr: int = 0, ld: Optional[int] = None, n_valid_in: Optional[Sequence[int]] = None, results: bool = True):
        """Device pointers in (o [K][ld], the F forcing and T target arrays), d loss / d o out ([K][ld]); returns
        (loss, gradient of the raw global parameters, n_valid), or None with results=False (asynchronous on the stream).
        include/easyhybrid_hip.h: eh_mech_loss_vjp."""
        fp = (C.c_void_p * max(1, len(forcing_ptrs)))(*forcing_ptrs)
        tp = (C.c_void_p * max(1, len(target_ptrs)))(*target_ptrs)
        nin = (C.c_int64 * len(n_valid_in))(*n_valid_in) if n_valid_in is not None else None
        G = sum(1 for j in range(self.desc.n_params) if self.desc.param_kind[j] == L.PAR_GLOBAL)
        loss, nv = C.c_float(), C.c_int64()
        gg = np.zeros(max(G, 1), np.float32)
        self._chk(self._lib.eh_mech_loss_vjp(self._h, count, ld if ld is not None else count, o_ptr, fp, tp, nin, d_o_ptr, yhat_ptr or None,
                                             C.byref(loss) if results else None, gg.ctypes.data_as(L._F) if results else None,
                                             C.byref(nv) if results else None))
        return (float(loss.value), gg[:G], int(nv.value)) if results else None

    # -- data-parallel seam ----------------------------------------------------------------------
    def dp_grad(self, first: int, count: int):
        self._chk(self._lib.eh_dp_grad(self._h, first, count))

    def dp_counts(self, first: int, count: int):
        """multi-target models: this shard's per-target sums of the window into EH_BUF_TCOUNT (all-reduce it, then dp_grad)"""
        self._chk(self._lib.eh_dp_counts(self._h, first, count))

    def dp_moments(self, first: int, count: int, stage: int):
        """two-pass training losses under data parallelism: this shard's moment sums of the window into EH_BUF_MOMENT -- stage 0 about
        the common target shift, stage 1 (after the all-reduce) about the global mean of the predictions; all-reduce after each"""
        self._chk(self._lib.eh_dp_moments(self._h, first, count, stage))

    @property
    def two_pass_loss(self) -> bool:
        """the training loss of some target needs batch moments of the predictions ahead of the pass (pearson / kge / pbkge; rmse on a
        multi-target model)"""
        kinds = getattr(self, "_loss_kinds", None) or ["mse"] * len(self.target_names)
        return any(k in ("pearsonLoss", "kgeLoss", "pbkgeLoss") or (k == "rmse" and len(kinds) > 1) for k in kinds)

    def set_target_shift(self, shift, split: int = L.EH_SPLIT_TRAIN):
        """common shift of the shifted target sums (every rank must pass the same vector, e.g. the global mean of each target)"""
        c = np.ascontiguousarray(shift, np.float32)
        self._chk(self._lib.eh_set_target_shift(self._h, split, _fptr(c), c.size))

    def set_bn_shift(self, shift):
        """common per-predictor shift of the cross-GPU BatchNorm sums (every rank must pass the same vector)"""
        c = np.ascontiguousarray(shift, np.float32)
        self._chk(self._lib.eh_set_bn_shift(self._h, _fptr(c), c.size))

    def dp_bn_stats(self, first: int, count: int):
        self._chk(self._lib.eh_dp_bn_stats(self._h, first, count))

    def dp_shuffle(self, seed: int = 0, on: bool = True):
        """per-shard epoch shuffle of the windows the dp_* calls take (same permutation generator as train_epoch)"""
        self._chk(self._lib.eh_dp_shuffle(self._h, seed & (2**64 - 1), int(on)))

    def dp_apply(self, want_loss: bool = False):
        loss = C.c_float()
        self._chk(self._lib.eh_dp_apply(self._h, C.byref(loss) if want_loss else None))
        return float(loss.value) if want_loss else None

    def dp_fused_step(self, first: int, count: int) -> int:
        k = C.c_int32()
        self._chk(self._lib.eh_dp_fused_step(self._h, first, count, C.byref(k)))
        return int(k.value)

    # -- cross-GPU exchange without a collective call (include/easyhybrid_hip.h: eh_p2p_*) -----------
    def p2p_init(self, world: int, rank: int) -> bytes:
        buf = C.create_string_buffer(64)
        self._chk(self._lib.eh_p2p_init(self._h, world, rank, C.cast(buf, C.c_void_p), 64))
        return buf.raw

    def p2p_attach(self, handles: Sequence[bytes]):
        blob = C.create_string_buffer(b"".join(handles), 64 * len(handles))
        self._chk(self._lib.eh_p2p_attach(self._h, C.cast(blob, C.c_void_p), 64))

    def p2p_selftest(self, rounds: int = 8) -> bool:
        ok = C.c_int32()
        self._chk(self._lib.eh_p2p_selftest(self._h, rounds, C.byref(ok)))
        return bool(ok.value)

    def p2p_disable(self):
        self._chk(self._lib.eh_p2p_disable(self._h))

    @staticmethod
    def _group_call(fn, engines, *args):
        lib = L.lib()
        hs = (C.c_void_p * len(engines))(*[e._h.value for e in engines])
        st = fn(hs, len(engines), *args)
        if st != L.EH_OK:
            msg = lib.eh_last_error(None).decode()
            for e in engines:
                m = lib.eh_last_error(e._h).decode()
                if m:
                    msg = m
            _raise(st, msg)

    @staticmethod
    def p2p_init_local(engines: Sequence["HybridEngine"], selftest_rounds: int = 8) -> bool:
        """ONE process, several engines (one per device, all in fused_update mode): their step kernels exchange the sums
        themselves through plain pointers to each other's receive buffers -- no IPC, no collective call per step
        (include/easyhybrid_hip.h: eh_p2p_init_local).  False: the start-up self-test failed, the engines keep all-reducing."""
        ok = C.c_int32()
        HybridEngine._group_call(L.lib().eh_p2p_init_local, engines, selftest_rounds, C.byref(ok))
        return bool(ok.value)

    @staticmethod
    def p2p_check_local(engines: Sequence["HybridEngine"]) -> bool:
        """drain every member; False: an exchange ran into its deadline -- every member has left the peer-to-peer exchange and taken
        member 0's parameters and optimiser state (eh_p2p_check_local)"""
        ok = C.c_int32()
        HybridEngine._group_call(L.lib().eh_p2p_check_local, engines, C.byref(ok))
        return bool(ok.value)

    # -- the library's own RCCL communicator (include/easyhybrid_hip.h, eh_comm_*) ---------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """rank 0: the id of a new communicator, to be handed to every rank over any host channel"""
        lib = L.lib()
        buf = C.create_string_buffer(L.EH_COMM_ID_BYTES)
        st = lib.eh_comm_unique_id(buf, L.EH_COMM_ID_BYTES)
        if st != L.EH_OK:
            _raise(st, lib.eh_last_error(None).decode())
        return buf.raw

    def comm_init(self, unique_id: bytes, world: int, rank: int):
        self._chk(self._lib.eh_comm_init(self._h, C.c_char_p(unique_id), len(unique_id), world, rank))

    def comm_destroy(self):
        self._chk(self._lib.eh_comm_destroy(self._h))

    @staticmethod
    def comm_init_local(engines: Sequence["HybridEngine"]):
        """ONE process driving several engines (one per device): they become a local group whose all-reduces run inside the
        library without RCCL (include/easyhybrid_hip.h: eh_comm_init_local); rank = position in `engines`."""
        lib = L.lib()
        hs = (C.c_void_p * len(engines))(*[e._h.value for e in engines])
        st = lib.eh_comm_init_local(hs, len(engines))
        if st != L.EH_OK:
            msg = lib.eh_last_error(None).decode()
            for e in engines:
                m = lib.eh_last_error(e._h).decode()
                if m:
                    msg = m
            _raise(st, msg)

    @staticmethod
    def comm_group_begin():
        st = L.lib().eh_comm_group_begin()
        if st != L.EH_OK:
            _raise(st, L.lib().eh_last_error(None).decode())

    @staticmethod
    def comm_group_end():
        st = L.lib().eh_comm_group_end()
        if st != L.EH_OK:
            _raise(st, L.lib().eh_last_error(None).decode())

    @staticmethod
    def dp_train_step_group(engines: Sequence["HybridEngine"], firsts, count: int, want_loss: bool = False):
        """a whole data-parallel step of `engines` from this one thread (eh_dp_train_step_group); firsts[i]: engine i's window"""
        lib = L.lib()
        hs = (C.c_void_p * len(engines))(*[e._h.value for e in engines])
        fs = (C.c_int64 * len(engines))(*[int(f) for f in firsts])
        loss = C.c_float()
        st = lib.eh_dp_train_step_group(hs, len(engines), fs, count, C.byref(loss) if want_loss else None)
        if st != L.EH_OK:
            msg = lib.eh_last_error(None).decode()
            for e in engines:
                m = lib.eh_last_error(e._h).decode()
                if m:
                    msg = m
            _raise(st, msg)
        return float(loss.value) if want_loss else None

    def dp_allreduce(self, which: int, index: int = 0):
        self._chk(self._lib.eh_dp_allreduce(self._h, which, index))

    def dp_train_step(self, first: int, count: int, want_loss: bool = False):
        loss = C.c_float()
        self._chk(self._lib.eh_dp_train_step(self._h, first, count, C.byref(loss) if want_loss else None))
        return float(loss.value) if want_loss else None

    def device_buffer(self, which: int):
        p = C.c_void_p()
        n = C.c_int64()
        self._chk(self._lib.eh_device_buffer(self._h, which, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    # -- profiling -------------------------------------------------------------------------------
    def profile_enable(self, on):
        """False/0 off, True/1 = events around every kernel, S > 1 = one event pair per burst of S steps."""
        self._chk(self._lib.eh_profile_enable(self._h, int(on)))

    def jit_status(self):
        """(kernel pairs compiled at run time for a recorded closure and in use, compiler / failure log)"""
        n = C.c_int32()
        buf = C.create_string_buffer(8192)
        self._chk(self._lib.eh_jit_status(self._h, C.byref(n), buf, len(buf)))
        return int(n.value), buf.value.decode(errors="replace")

    def profile_samples(self, cap: int = 8192) -> np.ndarray:
        """ms of every recorded step (or burst); call before profile_read, which consumes them."""
        buf = np.empty(cap, np.float64)
        n = C.c_int64()
        self._chk(self._lib.eh_profile_samples(self._h, buf.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(n)))
        return buf[:int(n.value)].copy()

    def profile_read(self):
        n = C.c_int64(); a = C.c_double(); b = C.c_double()
        self._chk(self._lib.eh_profile_read(self._h, C.byref(n), C.byref(a), C.byref(b)))
        return int(n.value), float(a.value), float(b.value)
