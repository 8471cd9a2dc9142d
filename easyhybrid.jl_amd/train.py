"""Host-side mirror of the reference's training front door.

  train(model, data; nepochs, batchsize, opt, training_loss, loss_types, agg, random_seed, ...)
                                             -- src/training/train.jl:95-136,211-219
  TrainConfig (the fields the step consumes) -- src/config/TrainingConfig.jl:9-160, validate_config :162-185
  DataConfig  (split fields)                 -- src/config/DataConfig.jl:7-59
  prepare_data / split_data                  -- src/data/prepare_data.jl:31-63, src/data/split_data.jl:8-79
  EarlyStopping / best_or_final / TrainResults -- src/training/early_stopping.jl, src/config/TrainingConfig.jl:190-223

The per-minibatch work (`run_epoch!`, src/training/epoch.jl:13-33) and the per-epoch evaluation
(`evaluate_epoch`, :53-66) are calls into the HIP engine; this file is bookkeeping only.
"""
from __future__ import annotations

import copy
import os
import time
from dataclasses import dataclass, field, replace
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import _lib as L
from .models import SingleNNHybridModel

# ---------------------------------------------------------------------------------------------
# optimiser rule specs (Optimisers.jl constructors; reference re-exports them, EasyHybrid.jl:59)
# ---------------------------------------------------------------------------------------------


@dataclass(frozen=True)
class Adam:
    eta: float = 0.001
    beta: tuple = (0.9, 0.999)
    epsilon: float = 1e-8


@dataclass(frozen=True)
class AdamW:
    eta: float = 0.001
    beta: tuple = (0.9, 0.999)
    lambda_: float = 0.0
    epsilon: float = 1e-8


@dataclass(frozen=True)
class RMSProp:
    eta: float = 0.001
    rho: float = 0.9
    epsilon: float = 1e-8


@dataclass(frozen=True)
class Descent:
    eta: float = 0.1


def _opt_args(opt):
    if isinstance(opt, Adam):
        return dict(rule="Adam", lr=opt.eta, beta1=opt.beta[0], beta2=opt.beta[1], eps=opt.epsilon)
    if isinstance(opt, AdamW):
        return dict(rule="AdamW", lr=opt.eta, beta1=opt.beta[0], beta2=opt.beta[1], eps=opt.epsilon, weight_decay=opt.lambda_)
    if isinstance(opt, RMSProp):
        return dict(rule="RMSProp", lr=opt.eta, beta1=opt.rho, eps=opt.epsilon)
    if isinstance(opt, Descent):
        return dict(rule="Descent", lr=opt.eta)
    raise NotImplementedError(f"optimiser {opt!r}: only Optimisers.jl-style Adam/AdamW/RMSProp/Descent run on the device "
                              "(the Optimization.jl path, src/training/train_optimization.jl, is out of scope)")


# ---------------------------------------------------------------------------------------------
# configs
# ---------------------------------------------------------------------------------------------

_MAXIMIZE = {"pearson", "r2", "nse", "kge"}                      # loss_fn.jl:181-187
_DEVICE_METRICS = {"mse", "rmse", "mae", "r2", "nse", "pearson", "kge", "pbkge"}


def isbetter(new, best, loss_type) -> bool:                      # loss_fn.jl:189-194
    return new > best if loss_type in _MAXIMIZE else new < best


def check_training_loss(loss_type):                              # loss_fn.jl:196-205
    if loss_type in _MAXIMIZE:
        raise ValueError(f"Got a metric that is defined as `to be maximized` as a training loss: {loss_type}. "
                         "For training you must use a true loss (to be minimized), e.g. :mse.")


@dataclass
class WeightL2:
    """One term of extra_loss = (yhat, ps) -> (; name = lam * weight_l2(ps; key, normalize), ...) (src/utils/extract_weights.jl:
    64-91) -- the extra losses the device knows: functions of the parameters that are sums of squares of Dense leaves.
    net: the network of a MultiNNHybridModel the walk starts at (`weight_l2(ps.Rb; ...)`, the reference's own example), None =
    the whole tree; key: "weight" (default) or "bias".  TrainConfig.extra_loss takes one term, a list of terms or a dict
    name -> term (the NamedTuple the closure returns).  Closures of yhat cannot run inside the kernel."""
    lam: float
    normalize: bool = False
    net: Optional[str] = None
    key: str = "weight"
    name: Optional[str] = None

    def label(self) -> str:
        return self.name or ("weight_l2" if self.net is None and self.key == "weight" else f"l2_{self.net or self.key}")


def _extra_fn(extra_loss):
    """TrainConfig.extra_loss -> the function of the predictions in it (`extra_loss(yhat[, ps])`, compute_loss.jl:31-34), or None.
    It may stand alone or in a list next to WeightL2 terms; it is recorded (program.trace_extra_loss) when the engine is created."""
    if callable(extra_loss) and not isinstance(extra_loss, WeightL2):
        return extra_loss
    if isinstance(extra_loss, (list, tuple)):
        fns = [v for v in extra_loss if callable(v) and not isinstance(v, WeightL2)]
        if len(fns) > 1:
            raise ValueError("extra_loss: one function of the predictions (it may return several entries)")
        return fns[0] if fns else None
    return None


def _extra_terms(extra_loss) -> List[WeightL2]:
    """TrainConfig.extra_loss -> its WeightL2 terms ([] for None; a function of the predictions is _extra_fn's); anything else is refused"""
    if extra_loss is None:
        return []
    if callable(extra_loss) and not isinstance(extra_loss, WeightL2):
        return []
    if isinstance(extra_loss, (list, tuple)) and any(callable(v) and not isinstance(v, WeightL2) for v in extra_loss):
        extra_loss = [v for v in extra_loss if isinstance(v, WeightL2)]
        if not extra_loss:
            return []
    if isinstance(extra_loss, WeightL2):
        return [extra_loss]
    if isinstance(extra_loss, dict) and extra_loss and all(isinstance(v, WeightL2) for v in extra_loss.values()):
        return [WeightL2(v.lam, v.normalize, v.net, v.key, str(k)) for k, v in extra_loss.items()]
    if isinstance(extra_loss, (list, tuple)) and extra_loss and all(isinstance(v, WeightL2) for v in extra_loss):
        terms = list(extra_loss)
        if len({t.label() for t in terms}) != len(terms):
            raise ValueError("extra_loss: two terms with the same name (a NamedTuple has distinct fields): give them `name`s")
        return terms
    raise NotImplementedError("extra_loss: WeightL2 terms (one, a list or a dict of them) and / or ONE function of the predictions "
                              "`f(yhat[, ps])` returning np.sum(...) / np.mean(...) entries are built")


def _apply_extra_loss(eng, model, terms: List[WeightL2], agg: str = "sum", n_fn_entries: int = 0):
    """the extra loss terms and `agg` (TrainingConfig.jl:76-77): the training loss is agg([agg(per-target losses), extra entries...])
    (compute_loss.jl:31-34,50-53) -- the engine needs the number of extra entries for agg = mean"""
    if len(terms) == 1 and terms[0].net is None and terms[0].key == "weight":
        eng.set_weight_l2(terms[0].lam, terms[0].normalize)
    elif terms:
        eng.set_weight_l2_coef(model.l2_coefficients(terms))
    eng.set_agg(agg, len(terms) + n_fn_entries)          # (entries of a function of the predictions ride on targets of their own: engine(extra_fn=...))


def _agg_name(agg) -> str:
    """TrainConfig.agg -> "sum" / "mean" (the two the device implements); the functions themselves are accepted like the reference's
    `agg::Function` (sum, np.sum, np.mean, statistics.mean)"""
    name = agg if isinstance(agg, str) else getattr(agg, "__name__", None)
    if name in ("sum", "nansum", "fsum"):
        return "sum"
    if name in ("mean", "nanmean", "fmean", "average"):
        return "mean"
    raise NotImplementedError(f"agg {agg!r}: the device implements sum and mean (TrainingConfig.jl:76-77)")


def _extra_loss_values(model, theta, terms: List[WeightL2], agg: str = "sum", fn=None, preds=None) -> Dict[str, float]:
    """the extra losses of flat parameters `theta` -- and, fn / preds, of the predictions of one split -- (host side, for the history;
    compute_loss.jl:39-44: each entry and their agg)"""
    th = np.asarray(theta, np.float64)
    out = {}
    if fn is not None and preds is not None:
        import inspect
        try:
            two = len([p for p in inspect.signature(fn).parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) and p.default is p.empty]) >= 2
        except (TypeError, ValueError):
            two = False
        class _Ps:                                    # `ps` as the function sees it in the reference: the global parameters' raw one-element vectors
            def __init__(self, glob): self._g = glob
            def __getattr__(self, k):
                if k.startswith("_") or k not in self._g:
                    raise AttributeError(k)
                return np.asarray(self._g[k], np.float64)
            def __getitem__(self, k): return np.asarray(self._g[k], np.float64)
        res = fn(preds, _Ps(model.unpack(np.asarray(theta, np.float32))[1])) if two else fn(preds)
        if hasattr(res, "_asdict"):
            res = res._asdict()
        items = res.items() if isinstance(res, dict) else [(f"extra_{i + 1}", v) for i, v in enumerate(res if isinstance(res, (list, tuple)) else [res])]
        for k, v in items:
            out[str(k)] = float(v)
    for t in terms:
        m = model.l2_mask(t.net, t.key)
        sq = float(np.sum(th[m] * th[m]))
        out[t.label()] = float(t.lam) * (sq / max(1, int(m.sum())) if t.normalize else sq)
    vals = list(out.values())
    out[agg] = float(sum(vals) / len(vals)) if (agg == "mean" and vals) else float(sum(vals))      # (; extra_loss_values..., Symbol(agg) => agg(...)), compute_loss.jl:42-44
    return out


@dataclass
class TrainConfig:
    nepochs: int = 200
    batchsize: int = 64
    opt: Any = field(default_factory=lambda: Adam(0.01))
    patience: int = 2**62
    training_loss: Any = "mse"           # a name, or a function f(yhat, y) -> np.mean(per-sample terms) (loss_fn.jl: training_loss::Function)
    loss_types: List[str] = field(default_factory=lambda: ["mse", "r2"])
    agg: Any = "sum"                     # TrainingConfig.jl:76-77 `agg::Function`: "sum" / "mean" (or the functions themselves)
    extra_loss: Any = None               # TrainingConfig.jl:74; None, a WeightL2 term, or a list / dict (name -> term) of them
    train_from: Any = None
    random_seed: Optional[int] = 161803
    return_model: str = "best"
    keep_history: bool = True
    show_progress: bool = False
    device: int = 0
    # not in the reference (it has no multi-GPU path): None = data parallel iff torch.distributed is initialised with
    # more than one rank; True / False force it.  `batchsize` stays the GLOBAL minibatch, split evenly over the ranks.
    distributed: Optional[bool] = None
    # not in the reference -- how the device runs the step (DESIGN.md sections 3.8 and 6):
    #   specialize   "auto" (default): step kernels compiled at run time (hiprtc) with this model's descriptor as a compile-time
    #                constant -- before the first step when `random_seed` is set (a reproducible run: about a second the first time, instant
    #                from the disk cache), in a background thread otherwise (training starts on the kernels built ahead of time and
    #                switches when the compiled one is ready; the two binaries agree to ~1e-6, not bit for bit).  True: compile before the
    #                first step.  False: only the kernels built ahead of time.  Every compiled kernel is checked against the one built
    #                ahead of time on the first batch before it takes over (eh_jit_status says when it was refused).
    #   fused_update "auto" (default): one kernel per step where the model allows it (single target, per-wave kernel family, no
    #                weight_l2 / moment-based loss) -- the optimiser update of a step runs in the prologue of the next one and the
    #                partial sums meet through float atomics, reproducible to ~1e-7, not bitwise.  A run with `random_seed` set -- the
    #                default, as in the reference (TrainingConfig.jl:85-86), where a seeded CPU run IS reproducible -- takes that form
    #                only where it is bitwise reproducible: minibatches one workgroup covers (up to 256 samples: several steps per
    #                launch, sums in one fixed order); larger minibatches run the deterministic step + reduce/optimiser pair (14.6
    #                against 10.2 us per step at batch 65 536 on the headline model, tools/bench_step_modes.py: `random_seed=None` or
    #                `fused_update=True` buys the difference back).  False: always the deterministic pair.  True: insist on one kernel
    #                per step (raises where it is not built).  (An ORDERED one-kernel form -- fixed-point sums, integer atomics -- was
    #                built and measured in round 6: 13.9 us, and 0.6 us on the float form it shared the kernel with; removed.)
    # bench.py measures specialize on and fused_update = True at the engine.
    specialize: Any = "auto"
    fused_update: Any = "auto"
    # not in the reference: "lazy" (default) = the observed-vs-predicted tables and the physical parameters of the returned model
    # (TrainResults.train_obs_pred / val_obs_pred / train_diffs / val_diffs, train.jl:130-136) are computed when they are first read --
    # the engine stays resident until then (or until the results are dropped / TrainResults.release()); "eager" = before train() returns,
    # as the reference does.  Same values either way.  (An engine the caller passed in is never kept: eager.)
    predictions: str = "lazy"
    # not in the reference: True = TrainResults.timing splits the wall-clock of the epoch loop into training steps / evaluation passes /
    # host bookkeeping (one extra device synchronisation per epoch, after the steps, so that the split is clean: bench.py `train_e2e`)
    timing: bool = False


@dataclass
class DataConfig:
    shuffleobs: bool = False
    split_by_id: Any = None
    split_data_at: float = 0.8
    folds: Any = None
    val_fold: Optional[int] = None


def _apply_step_mode(eng, tc: "TrainConfig"):
    """TrainConfig.specialize / fused_update -> engine options (single-GPU training)"""
    if tc.fused_update not in (True, False, "auto") or tc.specialize not in (True, False, "auto"):
        raise ValueError("specialize / fused_update must be True, False or 'auto'")
    if tc.fused_update is not False:
        try:
            # "auto" in a seeded run: one kernel per step only where that is bitwise reproducible (engine option value 2: minibatches one
            # workgroup covers), the deterministic pair elsewhere -- two default train(random_seed = s) calls are the same bits
            # "auto" in a seeded run: one kernel per step only where that is bitwise reproducible (engine option value 2: minibatches one
            # workgroup covers), the deterministic pair elsewhere -- two default train(random_seed = s) calls are the same bits
            eng.set_option("fused_update", 2 if (tc.fused_update == "auto" and tc.random_seed is not None) else 1)
        except (NotImplementedError, RuntimeError):
            if tc.fused_update is True:
                raise
    # "auto": a run that asked for reproducibility (random_seed set -- the default, TrainingConfig.jl:86) compiles BEFORE the first step
    # (about a second the first time, instant from the disk cache): the background build would switch binaries at a timing-dependent step,
    # and the two agree to ~1e-6, not bit for bit (advisor, round 3).  Without a seed the build runs in the background.
    eng.set_option("specialize", (1 if tc.random_seed is not None else 2) if tc.specialize == "auto" else int(bool(tc.specialize)))


def validate_config(cfg: TrainConfig):                           # TrainingConfig.jl:162-185
    if cfg.nepochs < 0:
        raise ValueError("nepochs must be >= 0")
    if cfg.batchsize < 1:
        raise ValueError("batchsize must be >= 1")
    if cfg.return_model not in ("best", "final"):
        raise ValueError("return_model must be :best or :final")
    if not cfg.loss_types:
        raise ValueError("loss_types must not be empty")
    from .engine import PerTarget
    tl = cfg.training_loss
    per_target = tl.losses if isinstance(tl, PerTarget) else (tl if isinstance(tl, (list, tuple)) and tl and not callable(tl[0]) else None)
    if per_target is not None:                                   # PerTarget((l_1, ..., l_T)), compute_loss.jl:128-145
        for lt in per_target:
            if callable(lt) or (isinstance(lt, (list, tuple)) and lt and callable(lt[0])):      # a function, or (f, args) / (f, kwargs): recorded and compiled into the step kernel
                continue
            check_training_loss(lt)
            if lt not in L.TRAINING_LOSSES:
                raise NotImplementedError(f"training_loss {lt!r}: the device implements {sorted(L.TRAINING_LOSSES)}")
    elif not callable(tl) and not (isinstance(tl, (list, tuple)) and tl and callable(tl[0])):      # a function (or (f, args), (f, kwargs): loss_fn.jl:92-107) is recorded and compiled into the step kernel
        check_training_loss(cfg.training_loss)
        if cfg.training_loss not in L.TRAINING_LOSSES:
            raise NotImplementedError(f"training_loss {cfg.training_loss!r}: the fused kernel implements {sorted(L.TRAINING_LOSSES)} "
                                      "or a function f(yhat, y) = mean of per-sample terms")
    _agg_name(cfg.agg)                    # sum or mean
    _extra_terms(cfg.extra_loss)          # (refuses what the device cannot run)
    for lt in cfg.loss_types:
        if lt not in _DEVICE_METRICS:
            raise NotImplementedError(f"loss type {lt!r} is not computed by the eval kernel (have {sorted(_DEVICE_METRICS)})")


# ---------------------------------------------------------------------------------------------
# data preparation (host; runs once)
# ---------------------------------------------------------------------------------------------


def _columns(data) -> Dict[str, np.ndarray]:
    if hasattr(data, "columns") and hasattr(data, "__getitem__") and not isinstance(data, dict):   # pandas DataFrame
        return {str(c): np.asarray(data[c]) for c in data.columns}
    if isinstance(data, dict):
        return {k: np.asarray(v) for k, v in data.items()}
    raise TypeError("data must be a dict of columns, a pandas DataFrame or a prepared ((X, forcings), targets) tuple")


def prepare_data(model: SingleNNHybridModel, data, drop_missing_rows: bool = True):
    """-> ((X (P,N) float32, {forcing: (N,)}), {target: (N,)}); prepare_data.jl:31-63."""
    if isinstance(data, tuple):
        return data
    cols = _columns(data)
    need = list(dict.fromkeys(list(model.predictors) + list(model.forcing) + list(model.targets)))
    for c in need:
        if c not in cols:
            raise KeyError(f"column {c!r} missing from data")
    # (float32 / float64 columns stay what they are -- the device data are Float32 like the reference's, prepare_data.jl:58-60 -- anything
    #  else goes through float64; no copy is made where none is needed: at 4 M rows the copies ARE the cost of this function)
    arr = {c: (a if (a := np.asarray(cols[c])).dtype in (np.float32, np.float64) else a.astype(np.float64)) for c in need}
    n = len(next(iter(arr.values())))
    if drop_missing_rows:
        predforce = [c for c in need if c not in model.targets]
        miss = np.zeros(n, bool)
        for c in predforce:
            miss |= np.isnan(arr[c])
        some_target = np.zeros(n, bool)
        for c in model.targets:
            some_target |= ~np.isnan(arr[c])
        keep = ~miss & some_target
        if not keep.all():
            arr = {c: v[keep] for c, v in arr.items()}
    nkept = len(next(iter(arr.values())))
    if model.predictors:
        X = np.empty((len(model.predictors), nkept), np.float32)
        for i, p_ in enumerate(model.predictors):
            X[i] = arr[p_]                                        # (one pass per column, cast on the way)
    else:
        X = np.zeros((0, nkept), np.float32)                      # (no predictors: a model without a network)
    return (X, {f: arr[f].astype(np.float32, copy=False) for f in model.forcing}), {t: arr[t].astype(np.float32, copy=False) for t in model.targets}


def _no_nan(a) -> bool:
    """no NaN in `a`, found by ONE reduction pass without a temporary: a NaN anywhere makes the sum NaN (so does inf - inf: then the exact
    test decides).  np.isnan(a) | ... over 4 M rows allocates and walks a boolean array per column -- a third of a short train() call."""
    return not bool(np.isnan(np.add.reduce(a))) or not bool(np.isnan(a).any())


def _column_views(model, data, cfg: DataConfig):
    """the caller's table as float32 column views, if the common case applies (see _split_columns) -> (columns, names, rows) or None"""
    if isinstance(data, tuple) or cfg.shuffleobs or cfg.split_by_id is not None or cfg.folds is not None or cfg.val_fold is not None:
        return None
    cols = _columns(data)
    need = list(dict.fromkeys(list(model.predictors) + list(model.forcing) + list(model.targets)))
    arr = {}
    for c in need:
        if c not in cols:
            raise KeyError(f"column {c!r} missing from data")
        a = np.asarray(cols[c])
        if a.dtype != np.float32 or a.ndim != 1 or not a.flags.c_contiguous:
            return None
        arr[c] = a
    if not model.predictors:
        return None
    n = len(next(iter(arr.values())))
    if any(len(a) != n for a in arr.values()) or n == 0:
        return None
    return arr, need, n


def _no_row_to_drop(model, arr, need, n) -> bool:
    """prepare_data.jl:31-63 keeps a row whose predictors and forcings are all there and that has some target: true if that is every row"""
    if n >= (1 << 20) and len(need) > 1:                              # (the reductions release the interpreter lock: one thread per column)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(4, len(need))) as ex:
            clean = dict(zip(need, ex.map(_no_nan, [arr[c] for c in need])))
    else:
        clean = {c: _no_nan(arr[c]) for c in need}
    if not all(clean[c] for c in need if c not in model.targets):
        return False                                                  # rows to drop: the general path
    if len(model.targets) == 1:
        if not clean[model.targets[0]]:
            return False                                              # (a row without its only target is dropped)
    elif not any(clean[t] for t in model.targets):
        some = np.zeros(n, bool)
        for t in model.targets:
            some |= ~np.isnan(arr[t])
        if not some.all():
            return False
    return True


def _take_views(model, arr, cfg: DataConfig, n):
    k = int(np.clip(round(cfg.split_data_at * n), 0, n))             # MLUtils.splitobs(at = ...)

    def take(sl):
        return ([arr[p][sl] for p in model.predictors], {f: arr[f][sl] for f in model.forcing}), {t: arr[t][sl] for t in model.targets}
    return take(slice(0, k)), take(slice(k, n))


def _split_columns(model, data, cfg: DataConfig):
    """The common case of train(model, table; ...) without copying the table: float columns, the contiguous split of MLUtils.splitobs
    (no shuffleobs / split_by_id / folds), no row to drop (prepare_data.jl:31-63 keeps a row whose predictors and forcings are all there and
    that has some target) -> (train, val), each ((predictor rows: list of (N,) views), {forcing: view}), {target: view}); None = take the
    general path (prepare_data + split_data: same result, through copies)."""
    cv = _column_views(model, data, cfg)
    if cv is None or not _no_row_to_drop(model, *cv):
        return None
    return _take_views(model, cv[0], cfg, cv[2])


class _EarlyUpload:
    """train() on a large table: the engine is created and both splits are uploaded by a worker thread WHILE the columns are screened for
    rows to drop (1.6 ms of a 16 ms call on the 4 M-row headline data set -- the screen only decides whether the views may be used as they
    are; the upload's staged copies release the interpreter lock).  If the screen says no, the engine is closed and the general path runs."""
    def __init__(self, model, device, xfn, views):
        import threading
        self.engine, self.error, self.engine_s, self.upload_s = None, None, 0.0, 0.0
        (xtr, ftr, ytr), (xva, fva, yva) = [(a[0][0], a[0][1], a[1]) for a in views]

        def work():
            try:
                t0 = time.perf_counter()
                eng = model.engine(device, extra_fn=xfn)
                self.engine = eng
                t1 = time.perf_counter()
                eng.set_data(L.EH_SPLIT_TRAIN, xtr, [ftr[f] for f in model.forcing], [ytr[t] for t in model.targets])
                eng.set_data(L.EH_SPLIT_VAL, xva, [fva[f] for f in model.forcing], [yva[t] for t in model.targets])
                self.engine_s, self.upload_s = t1 - t0, time.perf_counter() - t1
            except BaseException as e:                  # (re-raised by the caller's thread)
                self.error = e
        self._t = threading.Thread(target=work, name="eh-early-upload")
        self._t.start()

    def join(self, keep: bool):
        self._t.join()
        if self.error is not None or not keep:
            if self.engine is not None:
                self.engine.close()
            if self.error is not None:
                raise self.error
            return None
        return self.engine


def split_data(data, model, cfg: DataConfig = DataConfig(), rng: Optional[np.random.Generator] = None):
    """split_data.jl:8-79 -> (train, val) each ((X, forcings), targets)."""
    raw_cols = _columns(data) if not isinstance(data, tuple) else None
    (X, forc), targ = prepare_data(model, data)
    n = X.shape[1]
    if cfg.split_by_id is not None and cfg.folds is not None:
        raise ValueError("split_by_id and folds are not supported together; do the split when constructing folds")
    if cfg.split_by_id is not None:
        ids = np.asarray(raw_cols[cfg.split_by_id] if isinstance(cfg.split_by_id, str) else cfg.split_by_id)
        if len(ids) != n:
            raise ValueError("split_by_id needs one id per retained sample (drop NaN rows first)")
        uniq = np.array(list(dict.fromkeys(ids.tolist())))
        if cfg.shuffleobs:
            uniq = (rng or np.random.default_rng()).permutation(uniq)
        k = int(np.clip(round(cfg.split_data_at * len(uniq)), 0, len(uniq)))
        tr = np.flatnonzero(np.isin(ids, uniq[:k])); va = np.flatnonzero(np.isin(ids, uniq[k:]))
    elif cfg.folds is not None or cfg.val_fold is not None:
        if cfg.folds is None or cfg.val_fold is None:
            raise AssertionError("Provide folds together with val_fold.")
        f = np.asarray(raw_cols[cfg.folds] if isinstance(cfg.folds, str) else cfg.folds)
        if len(f) != n:
            raise AssertionError(f"length(folds) ({len(f)}) must equal number of samples ({n}).")
        va = np.flatnonzero(f == cfg.val_fold)
        if va.size == 0:
            raise AssertionError(f"No samples assigned to validation fold {cfg.val_fold}.")
        tr = np.setdiff1d(np.arange(n), va)
    else:
        k = int(np.clip(round(cfg.split_data_at * n), 0, n))           # MLUtils.splitobs(at = ...)
        if cfg.shuffleobs:
            idx = (rng or np.random.default_rng()).permutation(n)
            tr, va = idx[:k], idx[k:]
        else:
            tr, va = slice(0, k), slice(k, n)                           # contiguous halves: views, no gather

    def take(ix):
        return (np.ascontiguousarray(X[:, ix]), {k: v[ix] for k, v in forc.items()}), {k: v[ix] for k, v in targ.items()}
    return take(tr), take(va)


# ---------------------------------------------------------------------------------------------
# results
# ---------------------------------------------------------------------------------------------


@dataclass
class EpochSnapshot:                                               # initialization.jl:53-58
    l_train: dict
    l_val: dict
    y_train: Optional[dict] = None
    y_val: Optional[dict] = None


def _prediction_tables(eng, model, ytr, yva):
    """(train_obs_pred, val_obs_pred, train_diffs, val_diffs) of the model the engine holds: one forward pass per split (train.jl:130-136)"""
    def obs_pred(split, y):
        if eng.n_samples[split] == 0:
            return {}, None
        o = eng.forward(split)
        d = {t: y[t] for t in model.targets}
        d.update({t + "_pred": o[t] for t in model.targets})
        return d, o["parameters"]
    tr_op, tr_diff = obs_pred(L.EH_SPLIT_TRAIN, ytr)
    va_op, va_diff = obs_pred(L.EH_SPLIT_VAL, yva)
    return tr_op, va_op, tr_diff, va_diff


class _PendingPredictions:
    """The observed-vs-predicted tables and physical parameters of a finished train() call that nobody has looked at yet: the engine --
    parameters and BatchNorm state of the returned model set, both splits resident -- stays open until the first of the four fields is
    read (one forward pass per split + the copies to the host, then the engine is closed) or the results are dropped.  On the
    headline data set the eager copies are 50 MB into freshly mapped host memory: 1-22 ms of a 24-45 ms call, most of its
    run-to-run spread (tools/e2e_breakdown.py)."""

    def __init__(self, eng, model, ytr, yva):
        self.eng, self.model, self.ytr, self.yva, self.out = eng, model, ytr, yva, None

    def resolve(self):
        if self.out is None:
            try:
                self.out = _prediction_tables(self.eng, self.model, self.ytr, self.yva)
            finally:
                self.release()
        return self.out

    def release(self):
        eng, self.eng = self.eng, None
        if eng is not None:
            eng.close()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class TrainResults:                                                # TrainingConfig.jl:190-223
    """train_history, val_history, epoch_history, train_obs_pred, val_obs_pred, train_diffs, val_diffs, ps, st, best_epoch, best_loss
    (+ timing).  The four prediction fields may be pending (TrainConfig.predictions = "lazy", the default for an engine train() made
    itself): they are computed, exactly as the eager form would have, when first read; `release()` drops them and the device memory."""

    def __init__(self, train_history, val_history, epoch_history, train_obs_pred, val_obs_pred, train_diffs, val_diffs, ps, st, best_epoch, best_loss,
                 timing=None, pending: Optional[_PendingPredictions] = None):
        self.train_history, self.val_history, self.epoch_history = train_history, val_history, epoch_history
        self._pred = (train_obs_pred, val_obs_pred, train_diffs, val_diffs)
        self._pending = pending
        self.ps, self.st, self.best_epoch, self.best_loss = ps, st, best_epoch, best_loss
        self.timing = timing           # TrainConfig.timing: seconds of the call by part

    def _get(self, i):
        if self._pending is not None:
            self._pred = self._pending.resolve()
            self._pending = None
        return self._pred[i]

    train_obs_pred = property(lambda self: self._get(0))
    val_obs_pred = property(lambda self: self._get(1))
    train_diffs = property(lambda self: self._get(2))
    val_diffs = property(lambda self: self._get(3))

    @property
    def predictions_pending(self) -> bool:
        return self._pending is not None

    def release(self):
        """drop pending predictions (and the engine that would have made them) without computing them"""
        if self._pending is not None:
            self._pending.release()
            self._pending = None
            self._pred = ({}, {}, None, None)


def _losses(engine, split, targets, loss_types, agg="sum"):
    """(mse = (reco = .., sum = ..), r2 = (...)) as nested dicts, the aggregate under the name of `agg`; compute_loss.jl:55-66."""
    if engine.n_samples[split] == 0:
        return {lt: {**{t: float("nan") for t in targets}, agg: float("nan")} for lt in loss_types}
    metrics, _ = engine.eval(split)
    out = {}
    for lt in loss_types:
        per = {t: metrics[i][lt] for i, t in enumerate(targets)}
        tot = float(sum(per[t] for t in targets))
        per[agg] = tot / len(targets) if agg == "mean" else tot
        out[lt] = per
    return out


def _want_distributed(tc: TrainConfig) -> bool:
    if tc.distributed is not None:
        return bool(tc.distributed)
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:
        return False


def _train_distributed(model, tc: TrainConfig, rng, train_split, val_split) -> TrainResults:
    """The epoch loop of `train` with the steps sharded over the ranks of the default process group (one process per
    GPU).  Every rank holds a contiguous shard of the training split as its train data and takes batchsize / world
    samples of it per step (per-shard shuffling: the sample ORDER differs from a global shuffle, the distribution does
    not); gradients meet through DataParallel (peer-to-peer stores or one RCCL all-reduce per step).  Evaluation is
    replicated: a second engine per rank holds the full train / validation splits, so history, early stopping and the
    returned TrainResults are identical on every rank without further communication."""
    import torch
    import torch.distributed as dist
    from .dp import DataParallel, shard_range
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("train(distributed=True) needs an initialised torch.distributed process group (one rank per GPU)")
    (xtr, ftr, ytr), (xva, fva, yva) = train_split, val_split
    world, rank = dist.get_world_size(), dist.get_rank()
    device = torch.cuda.current_device()
    N = xtr.shape[1]
    lo, hi = shard_range(N, rank, world)
    xfn = _extra_fn(tc.extra_loss)                    # extra_loss as a function of the predictions: its entries are targets of the TRAINING engine
    ev = model.engine(device)                         # evaluation replica: full splits
    eng = model.engine(device, extra_fn=xfn)          # training replica: this rank's shard
    try:
        ev.set_data(L.EH_SPLIT_TRAIN, xtr, [ftr[f] for f in model.forcing], [ytr[t] for t in model.targets])
        ev.set_data(L.EH_SPLIT_VAL, xva, [fva[f] for f in model.forcing], [yva[t] for t in model.targets])
        eng.set_data(L.EH_SPLIT_TRAIN, xtr[:, lo:hi], [ftr[f][lo:hi] for f in model.forcing], [ytr[t][lo:hi] for t in model.targets])
        if tc.train_from is None:
            theta = model.initialparameters(rng)      # same seed -> same start on every rank
        else:
            theta = np.asarray(tc.train_from.ps if isinstance(tc.train_from, TrainResults) else tc.train_from[0], np.float32)
        eng.set_params(theta); ev.set_params(theta)
        eng.opt_init(**_opt_args(tc.opt))
        eng.set_training_loss(tc.training_loss)
        xterms = _extra_terms(tc.extra_loss)          # functions of the replicated parameters: every rank adds the same terms in eh_dp_apply
        aggn = _agg_name(tc.agg)
        _apply_extra_loss(eng, model, xterms, aggn, eng.n_pseudo)
        drv = DataParallel(eng, fused=tc.fused_update is not False, specialize=bool(tc.specialize))      # ("auto" compiles before the first step here: every rank has to be ready together)
        has_bn = bool(model.config.get("input_batchnorm"))
        first_lt = tc.loss_types[0]

        def snapshot():
            eng.synchronize()
            ev.set_params(eng.get_params())
            if has_bn:
                ev.set_bn_state(*eng.get_bn_state())
            snap = EpochSnapshot(_losses(ev, L.EH_SPLIT_TRAIN, model.targets, tc.loss_types, aggn),
                                 _losses(ev, L.EH_SPLIT_VAL, model.targets, tc.loss_types, aggn))
            if xterms or xfn is not None:
                for d, split in ((snap.l_train, L.EH_SPLIT_TRAIN), (snap.l_val, L.EH_SPLIT_VAL)):
                    preds = ev.forward(split, params=False) if (xfn is not None and ev.n_samples[split]) else None
                    d["extra_loss"] = _extra_loss_values(model, eng.get_params(), xterms, aggn, xfn, preds)
            return snap
        init = snapshot()
        history = [init]
        best_loss, best_ps, best_epoch, counter = init.l_val[first_lt][aggn], theta.copy(), 0, 0
        best_bn = eng.get_bn_state() if has_bn else None            # early_stopping.jl update!: best_ps AND best_st
        seed0 = tc.random_seed if tc.random_seed is not None else 0
        b = -(-tc.batchsize // world)                 # samples per rank per step
        steps = -(-(-(-N // world)) // b)             # ceil(largest shard / b): the same number of steps on every rank
        n_loc = hi - lo
        for epoch in range(1, tc.nepochs + 1):
            eng.dp_shuffle(seed0 + epoch + 7919 * rank)
            for s_ in range(steps):
                first = min(s_ * b, n_loc)
                drv.step(first, min(b, n_loc - first))
            snap = snapshot()
            if tc.keep_history:
                history.append(snap)
            cur = snap.l_val[first_lt][aggn]
            if isbetter(cur, best_loss, first_lt):
                best_loss, best_ps, best_epoch, counter = cur, eng.get_params(), epoch, 0
                if has_bn:
                    best_bn = eng.get_bn_state()
                if not tc.keep_history:
                    history[0] = snap
            else:
                counter += 1
            if counter >= tc.patience:
                break
        eng.synchronize()
        ps = best_ps if tc.return_model == "best" else eng.get_params()
        bn_out = (best_bn if tc.return_model == "best" else eng.get_bn_state()) if has_bn else None      # best_or_final
        ev.set_params(ps)
        if has_bn:
            ev.set_bn_state(*bn_out)

        def obs_pred(split, y):
            if ev.n_samples[split] == 0:
                return {}, None
            out = ev.forward(split)
            d = {t: y[t] for t in model.targets}
            d.update({t + "_pred": out[t] for t in model.targets})
            return d, out["parameters"]
        tr_op, tr_diff = obs_pred(L.EH_SPLIT_TRAIN, ytr)
        va_op, va_diff = obs_pred(L.EH_SPLIT_VAL, yva)
        st = {"fixed": {f: np.float32(model.parameters.default(f)) for f in model.fixed_param_names}}
        if has_bn:
            st["st_nn"] = {"running_mean": bn_out[0], "running_var": bn_out[1]}
        return TrainResults([s.l_train for s in history], [s.l_val for s in history], history, tr_op, va_op, tr_diff, va_diff,
                            ps, st, best_epoch, best_loss)
    finally:
        eng.close(); ev.close()


_GC_FROZEN = 0


def _freeze_gc_once(at_end=False):
    """The first train() call of a process runs ONE full garbage collection and freezes what survives it (gc.freeze: the interpreter's,
    NumPy's and -- if it is loaded -- torch's module-level objects, about a million of them).  Without this the cyclic collector's first
    full pass lands in the middle of the second to fourth call and walks all of them: 40-100 ms of a 17 ms call (tools/e2e_breakdown2.py:
    the spike sat wherever the allocation count happened to cross the threshold -- engine creation, upload, the first evaluation -- and
    was the 2 x run-to-run spread of a short train() call that rounds 4 and 5 could not explain).  EH_NO_GC_FREEZE=1 leaves the
    collector alone."""
    global _GC_FROZEN
    import sys
    # (torch imported since the last freeze: its objects are new to the collector -- once more; and once more when the first call of the
    #  process ENDS -- what it imported and built on the way, the engine library's bindings and the run-time compiler among it, was a 45 ms
    #  full pass at the end of the second call, tools/e2e_spikes.py)
    state = (4 if "torch" in sys.modules else 2) + (1 if at_end else 0)
    if _GC_FROZEN >= state or os.environ.get("EH_NO_GC_FREEZE"):
        return
    _GC_FROZEN = state
    import gc
    if gc.isenabled():
        gc.collect()
        gc.freeze()


def train(model: SingleNNHybridModel, data, save_ps=(), *, train_cfg: Optional[TrainConfig] = None,
          data_cfg: Optional[DataConfig] = None, engine=None, **kwargs) -> Optional[TrainResults]:
    """train(model, data; kwargs...) -> TrainResults (train.jl:211-219 -> _train :95-136).
    Flat kwargs override config fields exactly like override_configs (train.jl:300-314)."""
    tc = copy.copy(train_cfg) if train_cfg else TrainConfig()
    dc = copy.copy(data_cfg) if data_cfg else DataConfig()
    for k, v in kwargs.items():
        if hasattr(tc, k):
            setattr(tc, k, v)
        elif hasattr(dc, k):
            setattr(dc, k, v)
        else:
            raise TypeError(f"train: unknown keyword {k!r}")
    validate_config(tc)
    _freeze_gc_once()
    t_call = time.perf_counter()
    rng = np.random.default_rng(tc.random_seed)
    dist_run = _want_distributed(tc)
    own = engine is None
    xfn = _extra_fn(tc.extra_loss)                    # extra_loss as a function of the predictions (compute_loss.jl:31-34): recorded, its entries ride on targets of their own
    if xfn is not None and engine is not None and not engine.n_pseudo:
        raise ValueError("train(engine = ...): an extra_loss of the predictions needs an engine created with it (model.engine(device, extra_fn = f))")
    # the caller's float32 columns as they are: views, no stacked copy (4 M rows: 7 -> 1 ms); on a large table the upload starts before the
    # screen for rows to drop has finished (_EarlyUpload)
    fast, early = None, None
    cv = None if dist_run else _column_views(model, data, dc)
    if cv is not None:
        views = _take_views(model, cv[0], dc, cv[2])
        if own and cv[2] >= (1 << 20) and len(views[0][0][0][0]) > 0 and not os.environ.get("EH_NO_EARLY_UPLOAD"):      # (a large table, a training split that is not empty)
            early = _EarlyUpload(model, tc.device, xfn, views)
        ok = False
        try:
            ok = _no_row_to_drop(model, *cv)
        finally:
            if early is not None:
                engine_early = early.join(keep=ok)
        fast = views if ok else None
        if not ok:
            early = None
    (xtr, ftr, ytr), (xva, fva, yva) = [(a[0][0], a[0][1], a[1]) for a in (fast if fast is not None else split_data(data, model, dc, rng))]
    if (len(xtr[0]) if isinstance(xtr, list) else xtr.shape[1]) == 0:
        if early is not None:
            engine_early.close()
        return None                                                # train.jl:186 ("returns nothing on empty splits")
    if dist_run:
        return _train_distributed(model, tc, rng, (xtr, ftr, ytr), (xva, fva, yva))
    t_prep = time.perf_counter()
    if early is not None:
        eng = engine_early
        # (prepare_s: what the call waited for beyond the worker's engine + upload -- the part of the screen that did not hide behind them)
        t_prep = t_call + max(0.0, (t_prep - t_call) - early.engine_s - early.upload_s)
        t_eng = t_prep + early.engine_s
    else:
        eng = engine if engine is not None else model.engine(tc.device, extra_fn=xfn)
        t_eng = time.perf_counter()
    keep_engine = False
    try:
        if early is None:
            eng.set_data(L.EH_SPLIT_TRAIN, xtr, [ftr[f] for f in model.forcing], [ytr[t] for t in model.targets])
            eng.set_data(L.EH_SPLIT_VAL, xva, [fva[f] for f in model.forcing], [yva[t] for t in model.targets])
        if tc.timing:
            eng.synchronize()
        t_up = time.perf_counter()
        if tc.train_from is None:
            theta = model.initialparameters(rng)
        else:
            theta = np.asarray(tc.train_from.ps if isinstance(tc.train_from, TrainResults) else tc.train_from[0], np.float32)
        eng.set_params(theta)
        eng.opt_init(**_opt_args(tc.opt))
        eng.set_training_loss(tc.training_loss)
        xterms = _extra_terms(tc.extra_loss)
        aggn = _agg_name(tc.agg)
        _apply_extra_loss(eng, model, xterms, aggn, eng.n_pseudo)
        _apply_step_mode(eng, tc)
        first_lt = tc.loss_types[0]

        def snapshot():
            snap = EpochSnapshot(_losses(eng, L.EH_SPLIT_TRAIN, model.targets, tc.loss_types, aggn),
                                 _losses(eng, L.EH_SPLIT_VAL, model.targets, tc.loss_types, aggn))
            if xterms or xfn is not None:                # compute_loss.jl:39-44: eval mode reports the extra losses next to the metrics
                for d, split in ((snap.l_train, L.EH_SPLIT_TRAIN), (snap.l_val, L.EH_SPLIT_VAL)):
                    preds = eng.forward(split, params=False) if (xfn is not None and eng.n_samples[split]) else None
                    d["extra_loss"] = _extra_loss_values(model, eng.get_params(), xterms, aggn, xfn, preds)
            return snap
        t_setup = time.perf_counter()
        init = snapshot()
        history = [init]
        best_loss, best_ps, best_epoch, counter = init.l_val[first_lt][aggn], theta.copy(), 0, 0
        has_bn = bool(model.config.get("input_batchnorm"))
        best_bn = eng.get_bn_state() if has_bn else None            # early_stopping.jl update!: best_ps AND best_st
        seed0 = tc.random_seed if tc.random_seed is not None else int(rng.integers(2**31))
        tm = {"steps_s": 0.0, "eval_s": 0.0, "host_s": 0.0, "epochs": 0} if tc.timing else None
        if tm is not None:
            # outside the epoch loop, part by part: split + cast of the caller's columns | engine creation | interleave +
            # upload of both splits | parameters, optimiser, options (incl. a run-time compilation that is not in the disk cache) | the
            # evaluation of epoch 0
            tm.update(prepare_s=t_prep - t_call, engine_s=t_eng - t_prep, upload_s=t_up - t_eng, setup_s=t_setup - t_up, initial_eval_s=time.perf_counter() - t_setup)
        t_loop = time.perf_counter()
        for epoch in range(1, tc.nepochs + 1):
            t0 = time.perf_counter()
            eng.train_epoch(tc.batchsize, seed=seed0 + epoch, shuffle=True, want_loss=False)      # run_epoch!
            if tm is not None:
                eng.synchronize()
                t1 = time.perf_counter()
            snap = snapshot()                                                                  # evaluate_epoch
            if tm is not None:
                t2 = time.perf_counter()
                tm["steps_s"] += t1 - t0; tm["eval_s"] += t2 - t1; tm["epochs"] += 1
            if tc.keep_history:
                history.append(snap)
            cur = snap.l_val[first_lt][aggn]
            if isbetter(cur, best_loss, first_lt):                                             # early_stopping.jl:16-42
                best_loss, best_ps, best_epoch, counter = cur, eng.get_params(), epoch, 0
                if has_bn:
                    best_bn = eng.get_bn_state()
                if not tc.keep_history:
                    history[0] = snap
            else:
                counter += 1
            if counter >= tc.patience:
                break
        if tm is not None:
            eng.synchronize()
            tm["loop_s"] = time.perf_counter() - t_loop
            tm["host_s"] = tm["loop_s"] - tm["steps_s"] - tm["eval_s"]
        t_final = time.perf_counter()
        ps = best_ps if tc.return_model == "best" else eng.get_params()                        # best_or_final
        bn_out = (best_bn if tc.return_model == "best" else eng.get_bn_state()) if has_bn else None
        eng.set_params(ps)
        if has_bn:
            eng.set_bn_state(*bn_out)                                # the predictions below use the state that belongs to `ps`

        fixed = {f: np.float32(model.parameters.default(f)) for f in model.fixed_param_names}
        st = {"fixed": fixed}
        if has_bn:
            st["st_nn"] = {"running_mean": bn_out[0], "running_var": bn_out[1]}      # Lux BatchNorm state of the returned model (best_or_final)
        if tc.predictions not in ("lazy", "eager"):
            raise ValueError("predictions must be 'lazy' or 'eager'")
        if own and tc.predictions == "lazy":
            pending = _PendingPredictions(eng, model, ytr, yva)
            keep_engine = True           # (the results own it from here: closed when the predictions are read, released or dropped)
            preds = (None, None, None, None)
        else:
            pending, preds = None, _prediction_tables(eng, model, ytr, yva)
        if tm is not None:
            tm["final_predictions_s"] = time.perf_counter() - t_final       # (eager: forward over both splits with the returned parameters + the copies to the host)
            tm["call_s_before_close"] = time.perf_counter() - t_call
        return TrainResults([s.l_train for s in history], [s.l_val for s in history], history, *preds,
                            ps, st, best_epoch, best_loss, tm, pending)
    finally:
        if own and not keep_engine:
            eng.close()
        _freeze_gc_once(at_end=True)
