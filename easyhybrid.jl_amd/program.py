"""Recording a user's mechanistic closure as a device program (EH_MECH_PROGRAM).

The reference calls `mechanistic_model(; forcing..., params...)` with arrays and lets Zygote differentiate it
(src/models/GenericHybridModel.jl:420-425).  A closure cannot run inside a kernel, so this module calls it ONCE with
tracer numbers (`Sym`), which record every arithmetic operation as one instruction of a straight-line program over
value slots (include/easyhybrid_hip.h, `eh_prog_op`).  The step kernel evaluates that program per sample and runs the
reverse sweep over the same tape.  Nothing is evaluated numerically here: without the device library a traced model
can be constructed but not run.

What a closure may use: + - * / ** unary minus, abs, comparisons (> < >= <=) as arguments of `where`, and the functions
below (`exp log sqrt tanh sigmoid maximum minimum sin cos where`) -- or their NumPy namesakes, which dispatch here through
`__array_ufunc__`.  Everything is elementwise over samples; cross-sample operations (sum, mean, cumsum) are not
expressible, and Python control flow on values (`if x > 0`) is refused (use `where`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np

from . import _lib as L

OPS = {"add": 0, "sub": 1, "mul": 2, "div": 3, "neg": 4, "exp": 5, "log": 6, "pow": 7, "sqrt": 8, "tanh": 9, "sigmoid": 10,
       "max": 11, "min": 12, "abs": 13, "sin": 14, "cos": 15, "select": 16, "gt": 17}
OP_NAMES = {v: k for k, v in OPS.items()}
SLOT_PAR, SLOT_FORC, SLOT_CONST, SLOT_INSTR = 0, 8, 12, 28
MAX_PROG, MAX_CONST, MAX_OUT = 64, 16, 3


@dataclass(frozen=True)
class Program:
    """The ABI encoding: slots 0..7 parameters, 8..11 forcings, 12..27 constants, 28+i the result of instruction i."""
    params: Tuple[str, ...]
    forcings: Tuple[str, ...]          # canonical forcings = the ones the outputs depend on
    outputs: Tuple[str, ...]           # outputs the program computes (the targets)
    consts: Tuple[float, ...]
    code: Tuple[Tuple[int, int, int, int], ...]     # (op, a, b, c) slots
    out: Tuple[int, ...]               # slot of each output

    def words(self) -> List[int]:
        return [op | a << 8 | b << 16 | c << 24 for op, a, b, c in self.code]

    def as_dict(self) -> dict:
        """Plain data for the oracle / fixtures (no reference to this package)."""
        return dict(params=list(self.params), forcings=list(self.forcings), outputs=list(self.outputs),
                    consts=[float(c) for c in self.consts], code=[list(i) for i in self.code], out=list(self.out))


class _Graph:
    def __init__(self):
        self.nodes: List[tuple] = []          # ("par", j) | ("frc", name) | ("const", value) | (op, a, b, c) node ids
        self.key: Dict[tuple, int] = {}

    def node(self, *k) -> int:
        if k not in self.key:
            self.key[k] = len(self.nodes)
            self.nodes.append(k)
        return self.key[k]

    def const(self, v) -> int:
        v = float(np.float32(v))
        if math.isnan(v) or math.isinf(v):
            raise ValueError(f"constant {v} in a mechanistic program")
        return self.node("const", v)


def _is_num(x):
    return isinstance(x, (bool, np.bool_, int, float, np.integer, np.floating)) or (isinstance(x, np.ndarray) and x.ndim == 0)


class Sym:
    """A traced per-sample value."""
    __array_priority__ = 1000

    def __init__(self, g: _Graph, nid: int):
        self.g, self.nid = g, nid

    # -- plumbing ------------------------------------------------------------------------------
    def _lift(self, x) -> "Sym":
        if isinstance(x, Sym):
            if x.g is not self.g:
                raise ValueError("values of two different traces meet")
            return x
        if _is_num(x):
            return Sym(self.g, self.g.const(x))
        raise NotImplementedError(f"a mechanistic program cannot use a value of type {type(x).__name__} (only traced values and scalar constants)")

    def _op(self, name, *args) -> "Sym":
        ids = [self._lift(a).nid for a in args]
        return Sym(self.g, self.g.node(name, *ids))

    def _cval(self):
        n = self.g.nodes[self.nid]
        return n[1] if n[0] == "const" else None

    # -- arithmetic ----------------------------------------------------------------------------
    def __add__(self, o): return self._op("add", self, o)
    def __radd__(self, o): return self._lift(o)._op("add", o, self)
    def __sub__(self, o): return self._op("sub", self, o)
    def __rsub__(self, o): return self._lift(o)._op("sub", o, self)
    def __mul__(self, o): return o.__rmul__(self) if isinstance(o, MeanOf) else self._op("mul", self, o)
    def __rmul__(self, o): return self._lift(o)._op("mul", o, self)
    def __getitem__(self, i):                                       # (`ps.Q10[1]`: a global parameter is a one-element vector in the reference)
        if _is_uniform(self) and i in (0, 1, -1, Ellipsis):
            return self
        raise NotImplementedError("indexing a traced per-sample value")
    def __truediv__(self, o): return self._op("div", self, o)
    def __rtruediv__(self, o): return self._lift(o)._op("div", o, self)
    def __neg__(self): return self._op("neg", self)
    def __pos__(self): return self
    def __abs__(self): return self._op("abs", self)

    def __pow__(self, e):
        if _is_num(e):
            e = float(e)
            if e == int(e) and abs(e) <= 8:               # integer powers as products: valid for negative bases too
                n = int(abs(e))
                if n == 0:
                    return self._lift(1.0)
                r, b = None, self
                while n:
                    if n & 1:
                        r = b if r is None else r * b
                    n >>= 1
                    if n:
                        b = b * b
                return r if e > 0 else 1.0 / r
            if e == 0.5:
                return self._op("sqrt", self)
        return self._op("pow", self, e)                   # base > 0

    def __rpow__(self, b):
        if _is_num(b) and float(b) <= 0.0:
            raise ValueError(f"power with the non-positive constant base {b}")
        return self._lift(b)._op("pow", b, self)

    # -- comparisons: 1.0 / 0.0 values for `where` ------------------------------------------------
    def __gt__(self, o): return self._op("gt", self, o)
    def __lt__(self, o): return self._lift(o)._op("gt", o, self)
    def __ge__(self, o): return 1.0 - self._lift(o)._op("gt", o, self)
    def __le__(self, o): return 1.0 - self._op("gt", self, o)

    def __bool__(self):
        raise NotImplementedError("Python control flow on a traced value: use where(cond, a, b)")

    def mean(self, **kw):
        return MeanOf(self)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs:
            raise NotImplementedError(f"numpy.{ufunc.__name__}.{method} in a mechanistic program")
        f = _UFUNCS.get(ufunc.__name__)
        if f is None:
            raise NotImplementedError(f"numpy.{ufunc.__name__} is not a device program operation")
        return f(*inputs)

    def __array_function__(self, func, types, args, kwargs):
        f = _ARRAY_FUNCS.get(func.__name__)
        if f is None:
            raise NotImplementedError(f"numpy.{func.__name__} in a mechanistic program: only elementwise operations can be recorded "
                                      "(a closure that couples samples, e.g. through sum / mean / cumsum, has no per-sample device form)")
        return f(*args, **kwargs)


def _sym_of(*xs) -> Sym:
    for x in xs:
        if isinstance(x, Sym):
            return x
    raise TypeError("no traced value among the arguments (these functions are for use inside a mechanistic closure)")


def exp(x): return _sym_of(x)._op("exp", x)
def log(x): return _sym_of(x)._op("log", x)
def sqrt(x): return _sym_of(x)._op("sqrt", x)
def tanh(x): return _sym_of(x)._op("tanh", x)
def sigmoid(x): return _sym_of(x)._op("sigmoid", x)
def sin(x): return _sym_of(x)._op("sin", x)
def cos(x): return _sym_of(x)._op("cos", x)
def maximum(a, b): return _sym_of(a, b)._op("max", a, b)
def minimum(a, b): return _sym_of(a, b)._op("min", a, b)


def where(cond, a, b):
    """ifelse.(cond, a, b): `cond` a comparison of traced values (or any value, > 0 meaning true)."""
    return _sym_of(cond, a, b)._op("select", cond, a, b)


_UFUNCS = {
    "add": lambda a, b: _sym_of(a, b)._lift(a) + b, "subtract": lambda a, b: _sym_of(a, b)._lift(a) - b,
    "multiply": lambda a, b: _sym_of(a, b)._lift(a) * b, "divide": lambda a, b: _sym_of(a, b)._lift(a) / b,
    "true_divide": lambda a, b: _sym_of(a, b)._lift(a) / b,
    "power": lambda a, b: (a ** b) if isinstance(a, Sym) else b.__rpow__(a), "float_power": lambda a, b: (a ** b) if isinstance(a, Sym) else b.__rpow__(a),
    "negative": lambda a: -a, "positive": lambda a: a, "absolute": abs, "fabs": abs,
    "exp": exp, "log": log, "sqrt": sqrt, "tanh": tanh, "sin": sin, "cos": cos, "maximum": maximum, "minimum": minimum,
    "square": lambda a: a * a, "reciprocal": lambda a: 1.0 / a,
    "exp2": lambda a: 2.0 ** a, "log2": lambda a: log(a) * (1.0 / math.log(2.0)), "log10": lambda a: log(a) * (1.0 / math.log(10.0)),
    "greater": lambda a, b: _sym_of(a, b)._lift(a) > b, "less": lambda a, b: _sym_of(a, b)._lift(a) < b,
    "greater_equal": lambda a, b: _sym_of(a, b)._lift(a) >= b, "less_equal": lambda a, b: _sym_of(a, b)._lift(a) <= b,
}

def _is_uniform(x) -> bool:
    """a traced value that is the same for every sample of a batch: built from constants and the parameters the trace marked as global
    (`_Graph.uniform_pars`, trace_extra_loss_mixed) -- what a sum / mean over the samples may be scaled by"""
    if not isinstance(x, Sym):
        return False
    up = getattr(x.g, "uniform_pars", None)
    if up is None:
        return False
    seen, stack = set(), [x.nid]
    while stack:
        nid = stack.pop()
        if nid in seen:
            continue
        seen.add(nid)
        n = x.g.nodes[nid]
        if n[0] == "frc" or (n[0] == "par" and n[1] not in up):
            return False
        if n[0] not in ("par", "const"):
            stack.extend(n[1:])
    return True


class MeanOf:
    """np.mean(<traced per-sample value>): the only reduction a recorded training loss may end in.  What is linear in the mean stays
    a mean -- `w * mean(l)`, `mean(l) / c`, `mean(l) + c`, `mean(a) + mean(b)` (the reference's own test scales one:
    test/test_compute_loss.jl:36-47) -- anything else applied to it (sqrt, powers, a ratio of means) is refused."""
    def __init__(self, sym: "Sym"):
        self.sym = sym

    def _no(self, *a, **k):
        raise NotImplementedError("a recorded training loss has the form mean(l(yhat, y)): nothing but scaling / shifting can be applied to the mean "
                                  "(sqrt(mean(...)) etc. are the built-in rmse / nseLoss / kgeLoss)")

    @staticmethod
    def _num(x):
        return isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, bool)

    # (a traced value that is the same for every sample -- a global parameter of the model, trace_extra_loss_mixed -- scales a sum / mean like a constant)
    def __mul__(self, c): return type(self)(self.sym * float(c)) if self._num(c) else (type(self)(self.sym * c) if _is_uniform(c) else self._no())
    __rmul__ = __mul__
    def __truediv__(self, c): return type(self)(self.sym / float(c)) if self._num(c) else (type(self)(self.sym / c) if _is_uniform(c) else self._no())
    def __neg__(self): return type(self)(-self.sym)
    def __add__(self, o): return type(self)(self.sym + (o.sym if isinstance(o, MeanOf) else float(o))) if (self._num(o) or type(o) is type(self)) else self._no()
    __radd__ = __add__
    def __sub__(self, o): return type(self)(self.sym - (o.sym if isinstance(o, MeanOf) else float(o))) if (self._num(o) or type(o) is type(self)) else self._no()
    def __rsub__(self, o): return type(self)(float(o) - self.sym) if self._num(o) else self._no()
    __rtruediv__ = __pow__ = __rpow__ = __abs__ = _no

    def __array_ufunc__(self, ufunc, method, *inputs, **k):
        if method == "__call__" and not k and ufunc.__name__ in ("multiply", "add", "subtract", "true_divide", "divide", "negative") and len(inputs) <= 2:
            a = inputs[0]; b = inputs[1] if len(inputs) > 1 else None
            if ufunc.__name__ == "negative": return -self
            if ufunc.__name__ == "multiply": return (a if isinstance(a, MeanOf) else b) * (b if isinstance(a, MeanOf) else a)
            if ufunc.__name__ == "add": return (a if isinstance(a, MeanOf) else b) + (b if isinstance(a, MeanOf) else a)
            if ufunc.__name__ == "subtract": return a - b if isinstance(a, MeanOf) else b.__rsub__(a)
            if isinstance(a, MeanOf): return a / b
        self._no()


class SumOf(MeanOf):
    """np.sum(<traced per-sample value>): the other reduction an entry of the extra loss may end in (`sum(abs, yhat.var1)`,
    test/test_compute_loss.jl:259-261).  Same rules as MeanOf: scaling by constants and sums of sums stay a sum; a constant ADDED to a
    sum is refused (it would be added once per sample)."""
    def __add__(self, o): return type(self)(self.sym + o.sym) if type(o) is type(self) else self._no()
    __radd__ = __add__
    def __sub__(self, o): return type(self)(self.sym - o.sym) if type(o) is type(self) else self._no()
    def __rsub__(self, o): return self._no()


_ARRAY_FUNCS = {
    "mean": lambda x, **kw: MeanOf(x),
    "sum": lambda x, **kw: MeanOf(x) if _is_uniform(x) else SumOf(x),      # (sum over a global parameter's one-element vector: the value itself)
    "where": lambda cond, a, b: where(cond, a, b),
    "clip": lambda x, lo=None, hi=None, **kw: (x if lo is None else maximum(x, lo)) if hi is None else minimum(x if lo is None else maximum(x, lo), hi),
}

_ARITY = {"neg": 1, "exp": 1, "log": 1, "sqrt": 1, "tanh": 1, "sigmoid": 1, "abs": 1, "sin": 1, "cos": 1, "select": 3}


def _fold(g: _Graph, nid: int, memo: Dict[int, int]) -> int:
    """Constant folding in fp32 (an operation on constants only is done here, like Julia would do it per element)."""
    if nid in memo:
        return memo[nid]
    n = g.nodes[nid]
    if n[0] in ("par", "frc", "const"):
        memo[nid] = nid
        return nid
    args = [_fold(g, a, memo) for a in n[1:]]
    vals = [g.nodes[a][1] if g.nodes[a][0] == "const" else None for a in args]
    out = None
    if all(v is not None for v in vals):
        f32 = np.float32
        x = [f32(v) for v in vals]
        with np.errstate(all="ignore"):
            r = {"add": lambda: x[0] + x[1], "sub": lambda: x[0] - x[1], "mul": lambda: x[0] * x[1], "div": lambda: x[0] / x[1],
                 "neg": lambda: -x[0], "exp": lambda: np.exp(x[0]), "log": lambda: np.log(x[0]), "pow": lambda: np.power(x[0], x[1]),
                 "sqrt": lambda: np.sqrt(x[0]), "tanh": lambda: np.tanh(x[0]), "sigmoid": lambda: f32(1) / (f32(1) + np.exp(-x[0])),
                 "max": lambda: max(x[0], x[1]), "min": lambda: min(x[0], x[1]), "abs": lambda: abs(x[0]), "sin": lambda: np.sin(x[0]),
                 "cos": lambda: np.cos(x[0]), "select": lambda: x[1] if x[0] > 0 else x[2], "gt": lambda: f32(x[0] > x[1])}[n[0]]()
        out = g.const(r)
    if out is None:
        out = g.node(n[0], *args)
    memo[nid] = out
    return out


def trace(fn: Callable, params: Sequence[str], forcings: Sequence[str], targets: Sequence[str]) -> Program:
    """Call `fn(**forcings, **params)` with tracer values and encode what the `targets` outputs depend on."""
    params, forcings, targets = list(params), list(forcings), list(targets)
    if not 1 <= len(params) <= L.EH_MAX_PARAMS:
        raise NotImplementedError(f"a mechanistic program takes 1..{L.EH_MAX_PARAMS} parameters, the table has {len(params)}")
    g = _Graph()
    kw = {f: Sym(g, g.node("frc", f)) for f in forcings}
    for j, p in enumerate(params):
        if p in kw:
            raise ValueError(f"{p!r} is both a forcing and a parameter")
        kw[p] = Sym(g, g.node("par", j))
    res = fn(**kw)
    if hasattr(res, "_asdict"):
        res = res._asdict()
    if not isinstance(res, dict):
        raise TypeError("a mechanistic model returns a dict (NamedTuple) of named outputs")
    outs = []
    for t in targets:
        if t not in res:
            raise ValueError(f"target {t!r} is not an output of {getattr(fn, '__name__', 'the mechanistic model')} {tuple(res)}")
        if t not in outs:
            outs.append(t)
    if len(outs) > MAX_OUT:
        raise NotImplementedError(f"{len(outs)} distinct target outputs (device limit {MAX_OUT})")
    some = next(iter(kw.values()))
    memo: Dict[int, int] = {}
    for t in outs:
        if isinstance(res[t], MeanOf):
            raise NotImplementedError(f"output {t!r} of the mechanistic model is a sum / mean over samples: only elementwise operations can be recorded "
                                      "(a closure that couples samples has no per-sample device form)")
    roots = [_fold(g, some._lift(res[t]).nid, memo) for t in outs]
    # emit what the roots reach, in dependency order
    used_f: List[str] = []
    consts: List[float] = []
    code: List[Tuple[int, int, int, int]] = []
    slot: Dict[int, int] = {}

    def emit(nid: int) -> int:
        if nid in slot:
            return slot[nid]
        n = g.nodes[nid]
        if n[0] == "par":
            s = SLOT_PAR + n[1]
        elif n[0] == "frc":
            if n[1] not in used_f:
                used_f.append(n[1])
            s = SLOT_FORC + used_f.index(n[1])
        elif n[0] == "const":
            if n[1] not in consts:
                consts.append(n[1])
            s = SLOT_CONST + consts.index(n[1])
        else:
            ops = [emit(a) for a in n[1:]] + [0, 0]
            code.append((OPS[n[0]], ops[0], ops[1], ops[2]))
            s = SLOT_INSTR + len(code) - 1
        slot[nid] = s
        return s

    out = []
    for r in roots:
        s = emit(r)
        if s < SLOT_INSTR:                    # an output that is a bare input / constant: give it an instruction (x + 0 keeps the kernel uniform)
            zero = emit(g.const(0.0))
            code.append((OPS["add"], s, zero, 0))
            s = SLOT_INSTR + len(code) - 1
        out.append(s)
    if len(used_f) > L.EH_MAX_FORC:
        raise NotImplementedError(f"the program reads {len(used_f)} forcings (device limit {L.EH_MAX_FORC})")
    if len(consts) > MAX_CONST:
        raise NotImplementedError(f"the program has {len(consts)} distinct constants (device limit {MAX_CONST})")
    if len(code) > MAX_PROG:
        raise NotImplementedError(f"the program has {len(code)} operations (device limit {MAX_PROG})")
    return Program(tuple(params), tuple(used_f), tuple(outs), tuple(consts), tuple(code), tuple(out))


def trace_loss(fn: Callable) -> Program:
    """Record a custom training loss `fn(yhat, y)` (loss_fn.jl: training_loss::Function, called on the valid samples of a
    target).  It must be a mean of per-sample terms: `np.mean(l(yhat, y))` -- or the per-sample term itself, which is then
    averaged.  Value slot 0 = yhat, slot 1 = y."""
    g = _Graph()
    yhat, y = Sym(g, g.node("par", 0)), Sym(g, g.node("par", 1))
    res = fn(yhat, y)
    if isinstance(res, SumOf):
        raise NotImplementedError("numpy.sum in a training loss: only elementwise operations can be recorded, closed by ONE np.mean over the valid samples "
                                  "(a sum over samples is what an entry of the extra loss may be: program.trace_extra_loss)")
    if isinstance(res, MeanOf):
        res = res.sym
    if not isinstance(res, Sym):
        raise TypeError("a training loss function returns np.mean(<elementwise expression of yhat and y>)")
    root = _fold(g, res.nid, {})
    consts: List[float] = []
    code: List[Tuple[int, int, int, int]] = []
    slot: Dict[int, int] = {}

    def emit(nid: int) -> int:
        if nid in slot:
            return slot[nid]
        n = g.nodes[nid]
        if n[0] == "par":
            s = SLOT_PAR + n[1]
        elif n[0] == "const":
            if n[1] not in consts:
                consts.append(n[1])
            s = SLOT_CONST + consts.index(n[1])
        else:
            ops = [emit(a) for a in n[1:]] + [0, 0]
            code.append((OPS[n[0]], ops[0], ops[1], ops[2]))
            s = SLOT_INSTR + len(code) - 1
        slot[nid] = s
        return s

    out = emit(root)
    if out < SLOT_INSTR:
        zero = emit(g.const(0.0))
        code.append((OPS["add"], out, zero, 0))
        out = SLOT_INSTR + len(code) - 1
    if len(consts) > MAX_CONST or len(code) > MAX_PROG:
        raise NotImplementedError(f"the loss has {len(code)} operations / {len(consts)} constants (device limits {MAX_PROG} / {MAX_CONST})")
    return Program(("yhat", "y"), (), ("loss",), tuple(consts), tuple(code), (out,))


def trace_extra_loss(fn: Callable, outputs: Sequence[str]):
    """Record `extra_loss(yhat[, ps])` where it is a function of the PREDICTIONS (src/losses/compute_loss.jl:31-34): `fn` is called
    once with a dict output name -> traced per-sample value (and, if it takes a second argument, a stand-in for `ps` that refuses to
    be used: parameter penalties are WeightL2 terms) and must return a list / tuple / dict of entries, each `np.sum(...)` or
    `np.mean(...)` of an elementwise expression of ONE output -- the reference's own test: `[sum(abs, yhat.var1), sum(abs, yhat.var2)]`
    (test/test_compute_loss.jl:257-285).  Returns [(name, output name, "sum" | "mean", Program)]; the program has the form of a
    recorded training loss (value slot 0 = yhat, slot 1 = an unused y) so that the entry can ride on one more target of the model."""
    import inspect
    g = _Graph()
    yh = {o: Sym(g, g.node("par", k)) for k, o in enumerate(outputs)}

    class _NoPs:
        def __getattr__(self, k): raise NotImplementedError("extra_loss: the recorded form takes the predictions only; penalties on the parameters are WeightL2 terms")
        __getitem__ = __getattr__
    try:
        npar = len([p for p in inspect.signature(fn).parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) and p.default is p.empty])
    except (TypeError, ValueError):
        npar = 1
    res = fn(yh, _NoPs()) if npar >= 2 else fn(yh)
    if hasattr(res, "_asdict"):
        res = res._asdict()
    items = list(res.items()) if isinstance(res, dict) else [(f"extra_{i + 1}", v) for i, v in enumerate(res if isinstance(res, (list, tuple)) else [res])]
    out = []
    for name, v in items:
        if not isinstance(v, MeanOf):
            raise NotImplementedError(f"extra_loss entry {name!r}: np.sum(...) or np.mean(...) of an elementwise expression of one prediction (got {type(v).__name__})")
        root = _fold(g, v.sym.nid, {})
        deps, seen, stack = set(), set(), [root]
        while stack:
            nid = stack.pop()
            if nid in seen:
                continue
            seen.add(nid)
            n = g.nodes[nid]
            if n[0] == "par":
                deps.add(n[1])
            elif n[0] not in ("const", "frc"):
                stack.extend(n[1:])
        if len(deps) != 1:
            raise NotImplementedError(f"extra_loss entry {name!r} reads {len(deps)} predictions: an entry rides on ONE output (it becomes a target of its own)")
        k = deps.pop()
        consts: List[float] = []
        code: List[Tuple[int, int, int, int]] = []
        slot: Dict[int, int] = {}

        def emit(nid: int) -> int:
            if nid in slot:
                return slot[nid]
            n = g.nodes[nid]
            if n[0] == "par":
                s_ = SLOT_PAR                               # the one prediction the entry reads = value slot 0 (yhat of a recorded loss)
            elif n[0] == "const":
                if n[1] not in consts:
                    consts.append(n[1])
                s_ = SLOT_CONST + consts.index(n[1])
            else:
                ops = [emit(a) for a in n[1:]] + [0, 0]
                code.append((OPS[n[0]], ops[0], ops[1], ops[2]))
                s_ = SLOT_INSTR + len(code) - 1
            slot[nid] = s_
            return s_
        o_ = emit(root)
        if o_ < SLOT_INSTR:
            zero = emit(g.const(0.0))
            code.append((OPS["add"], o_, zero, 0))
            o_ = SLOT_INSTR + len(code) - 1
        if len(consts) > MAX_CONST or len(code) > MAX_PROG:
            raise NotImplementedError(f"extra_loss entry {name!r} has {len(code)} operations / {len(consts)} constants (device limits {MAX_PROG} / {MAX_CONST})")
        out.append((str(name), outputs[k], "sum" if isinstance(v, SumOf) else "mean", Program(("yhat", "y"), (), ("loss",), tuple(consts), tuple(code), (o_,))))
    return out


def _identity_entry_program() -> Program:
    """the per-sample function of an entry that IS an output of the model: l(yhat, y) = yhat + 0"""
    return Program(("yhat", "y"), (), ("loss",), (0.0,), ((OPS["add"], SLOT_PAR, SLOT_CONST, 0),), (SLOT_INSTR,))


def trace_extra_loss_mixed(model_fn: Callable, extra_fn: Callable, params: Sequence[str], forcings: Sequence[str], targets: Sequence[str],
                           global_params: Sequence[str], bounds: Dict[str, Tuple[float, float]]):
    """`extra_loss(yhat, ps)` whose entries read SEVERAL predictions, or predictions and global parameters (src/losses/compute_loss.jl:
    31-34: any function of the model's outputs and its parameter NamedTuple), for a mechanistic model that is itself a recorded closure:
    the model is traced again, `extra_fn` is called on its traced outputs, and the per-sample expression of every entry becomes one more
    OUTPUT of the mechanistic program -- the entry then is the sum / mean of that output over the batch and rides on a target of its own
    like the one-prediction entries of trace_extra_loss; its derivative reaches every prediction and parameter it reads through the
    program's reverse sweep.  `ps.<g>` / `ps["g"]` of a global parameter g is its RAW value, as in the reference (the model sees
    lower + (upper - lower) * sigmoid(raw), GenericHybridModel.jl:348-352): recovered inside the program as logit((value - lower) /
    (upper - lower)).  The network's weights are not reachable here (penalties on them are WeightL2 terms).  What stays linear in the
    reduction is allowed: `np.sum(a * b)`, `np.mean((a - b) ** 2) * 0.1`, `np.mean(a) * ps.Q10`, `np.sum(ps.Q10 ** 2)`; a product of
    two reductions is refused.  Returns (Program with the extra outputs, [(entry name, output name, "sum" | "mean", Program)])."""
    import inspect
    params, forcings, targets, global_params = list(params), list(forcings), list(targets), list(global_params)
    g = _Graph()
    g.uniform_pars = {params.index(p) for p in global_params if p in params}
    kw = {f: Sym(g, g.node("frc", f)) for f in forcings}
    for j, p in enumerate(params):
        kw[p] = Sym(g, g.node("par", j))
    res = model_fn(**kw)
    if hasattr(res, "_asdict"):
        res = res._asdict()
    some = next(iter(kw.values()))
    yh = {o: some._lift(v) for o, v in res.items() if not isinstance(v, MeanOf)}

    class _Ps:
        def __getattr__(self, k):
            if k in global_params and k in params:
                lo, hi = bounds[k]
                u = (kw[k] - float(lo)) / (float(hi) - float(lo))
                return log(u / (1.0 - u))
            raise NotImplementedError(f"extra_loss: ps.{k} -- only the global parameters {tuple(global_params)} are reachable in the recorded form "
                                      "(penalties on the network's weights are WeightL2 terms)")
        __getitem__ = __getattr__
    try:
        npar = len([p for p in inspect.signature(extra_fn).parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) and p.default is p.empty])
    except (TypeError, ValueError):
        npar = 1
    out = extra_fn(yh, _Ps()) if npar >= 2 else extra_fn(yh)
    if hasattr(out, "_asdict"):
        out = out._asdict()
    items = list(out.items()) if isinstance(out, dict) else [(f"extra_{i + 1}", v) for i, v in enumerate(out if isinstance(out, (list, tuple)) else [out])]
    outs: List[str] = []
    for t in targets:
        if t not in res:
            raise ValueError(f"target {t!r} is not an output of the mechanistic model {tuple(res)}")
        if t not in outs:
            outs.append(t)
    memo: Dict[int, int] = {}
    roots = [_fold(g, yh[t].nid, memo) for t in outs]
    entries = []
    for i, (name, v) in enumerate(items):
        if _is_uniform(v):
            v = MeanOf(v)                                  # a function of the global parameters alone: the same for every sample
        if not isinstance(v, MeanOf):
            raise NotImplementedError(f"extra_loss entry {name!r}: np.sum(...) / np.mean(...) of a per-sample expression, possibly scaled by constants "
                                      f"or global parameters (got {type(v).__name__})")
        oname = f"_xl{i + 1}"
        outs.append(oname)
        roots.append(_fold(g, v.sym.nid, memo))
        entries.append((str(name), oname, "sum" if isinstance(v, SumOf) else "mean", _identity_entry_program()))
    if len(outs) > MAX_OUT:
        raise NotImplementedError(f"{len(outs)} outputs (the model's targets + one per extra-loss entry): device limit {MAX_OUT}")
    used_f: List[str] = []
    consts: List[float] = []
    code: List[Tuple[int, int, int, int]] = []
    slot: Dict[int, int] = {}

    def emit(nid: int) -> int:
        if nid in slot:
            return slot[nid]
        n = g.nodes[nid]
        if n[0] == "par":
            s_ = SLOT_PAR + n[1]
        elif n[0] == "frc":
            if n[1] not in used_f:
                used_f.append(n[1])
            s_ = SLOT_FORC + used_f.index(n[1])
        elif n[0] == "const":
            if n[1] not in consts:
                consts.append(n[1])
            s_ = SLOT_CONST + consts.index(n[1])
        else:
            ops = [emit(a) for a in n[1:]] + [0, 0]
            code.append((OPS[n[0]], ops[0], ops[1], ops[2]))
            s_ = SLOT_INSTR + len(code) - 1
        slot[nid] = s_
        return s_
    out_slots = []
    for r in roots:
        s_ = emit(r)
        if s_ < SLOT_INSTR or s_ in out_slots:             # a bare input / constant, or the same value twice: an instruction of its own
            zero = emit(g.const(0.0))
            code.append((OPS["add"], s_, zero, 0))
            s_ = SLOT_INSTR + len(code) - 1
        out_slots.append(s_)
    if len(used_f) > L.EH_MAX_FORC or len(consts) > MAX_CONST or len(code) > MAX_PROG:
        raise NotImplementedError(f"model + extra-loss entries: {len(code)} operations / {len(consts)} constants / {len(used_f)} forcings "
                                  f"(device limits {MAX_PROG} / {MAX_CONST} / {L.EH_MAX_FORC})")
    return Program(tuple(params), tuple(used_f), tuple(outs), tuple(consts), tuple(code), tuple(out_slots)), entries
