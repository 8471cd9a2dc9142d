"""Mechanistic closures as device programs (EH_MECH_PROGRAM): recording (easyhybrid.jl_amd/program.py), the descriptor, and the
oracle's side of it -- the program's forward values against the closure itself on NumPy arrays, its reverse sweep against
central differences of the closure, and the hand-written RbQ10 closure against the registry model.  No GPU."""
import ctypes

import numpy as np
import pytest

import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd import _lib as L
from easyhybrid_jl_amd import program as P
from oracle import hybrid_oracle as ho
from tests import closures as cl
from tests import util


def _inputs(names, B, seed, lo=-1.0, hi=1.0):
    rng = np.random.default_rng(seed)
    return {n: rng.uniform(lo, hi, B) for n in names}


def test_rbq10_closure_records_four_operations():
    pg = P.trace(cl.rbq10_closure, ["rb", "Q10"], ["ta"], ["reco"])
    assert [P.OP_NAMES[c[0]] for c in pg.code] == ["sub", "mul", "pow", "mul"]
    assert pg.forcings == ("ta",) and pg.outputs == ("reco",) and pg.out == (P.SLOT_INSTR + 3,)
    assert pg.consts == (float(np.float32(0.1)), 15.0)
    w = pg.words()[0]
    assert (w & 255, (w >> 8) & 255, (w >> 16) & 255, w >> 24) == (1, P.SLOT_FORC, P.SLOT_CONST + 1, 0)


def test_only_what_the_targets_need_is_recorded():
    pg3 = P.trace(cl.flux_closure, list(cl.FLUX_TABLE), ["sw", "ta", "vpd"], ["nee", "gpp", "reco"])
    pg1 = P.trace(cl.flux_closure, list(cl.FLUX_TABLE), ["sw", "ta", "vpd"], ["reco"])
    assert pg1.forcings == ("ta",) and len(pg1.code) < len(pg3.code) and len(pg3.out) == 3
    # common subexpressions are shared: alpha * sw appears once
    muls = [c for c in pg3.code if P.OP_NAMES[c[0]] == "mul" and set(c[1:3]) == {P.SLOT_PAR + 0, P.SLOT_FORC + pg3.forcings.index("sw")}]
    assert len(muls) == 1


@pytest.mark.parametrize("fn,table,forc,targets", [
    (cl.rbq10_closure, cl.RBQ10_TABLE, ["ta"], ["reco"]),
    (cl.flux_closure, cl.FLUX_TABLE, ["sw", "ta", "vpd"], ["nee", "gpp", "reco"]),
    (cl.allops_closure, cl.ALLOPS_TABLE, ["u", "v"], ["y", "z"]),
])
def test_program_forward_equals_closure_and_vjp_equals_differences(fn, table, forc, targets):
    B = 257
    prog = P.trace(fn, list(table), forc, targets).as_dict()
    dt = np.dtype(np.float64)
    rng = np.random.default_rng(5)
    par = {n: rng.uniform(lo + 0.1 * (hi - lo), hi - 0.1 * (hi - lo), B) for n, (_, lo, hi) in table.items()}
    frc = _inputs(forc, B, 6, 0.0, 30.0) if fn is not cl.allops_closure else _inputs(forc, B, 6)
    val = ho.program_values(prog, par, frc, dt)
    direct = fn(**frc, **par)
    for o, s in zip(prog["outputs"], prog["out"]):
        # (the program holds its constants in fp32, as a Float32 closure of the reference does -- 0.1f0 in RbQ10; allops_closure only uses
        # constants that fp32 represents exactly and must agree to rounding)
        assert util.relerr(val[s], direct[o]) <= (1e-13 if fn is cl.allops_closure else 1e-6), o
    name = "t_" + fn.__name__
    ho.program_mech(name, prog, None)
    _, fwd, vjp = ho.MECH[name]
    out, aux = fwd(par, frc, dt)
    dout = {o: rng.normal(size=B) for o in prog["outputs"]}
    dpar = vjp(par, frc, out, aux, dout, dt)
    for n in table:
        h = 1e-6 * max(1.0, float(np.max(np.abs(par[n]))))
        up, dn = dict(par), dict(par)
        up[n] = par[n] + h; dn[n] = par[n] - h
        ou, _ = fwd(up, frc, dt); od, _ = fwd(dn, frc, dt)
        fd = sum(dout[o] * (ou[o] - od[o]) / (2 * h) for o in prog["outputs"])
        # (kinks of max / min / abs / where: a sample within h of one is skipped)
        ok = np.abs(dpar[n] - fd) <= 1e-5 * (1.0 + np.abs(fd))
        assert ok.mean() > 0.98, (n, float(np.max(np.abs(dpar[n] - fd))))


def test_closure_rbq10_equals_registry_rbq10_in_the_oracle():
    spec, theta, X, f, y = util.rbq10_case(300, "tanh", True, 0.1)
    util.register_closure("rbq10_closure", cl.rbq10_closure, list(cl.RBQ10_TABLE), ["ta"], ["reco"])
    spec_c = ho.HybridSpec(spec.n_pred, list(spec.hidden), "rbq10_closure", dict(spec.parameters), list(spec.neural), list(spec.glob),
                           ["reco"], spec.activation, spec.scale_nn_outputs)
    l0, g0, _ = ho.loss_and_grad(spec, np.asarray(theta, np.float64), X, f, y)
    l1, g1, _ = ho.loss_and_grad(spec_c, np.asarray(theta, np.float64), X, f, y)
    assert abs(l0 - l1) <= 1e-7 * abs(l0)                 # 0.1f0 as an fp32 constant vs the fp64 literal
    assert util.relerr(g1, g0) <= 1e-6


def test_descriptor_carries_the_program():
    m = eh.constructHybridModel(["x0", "x1"], ["ta"], ["reco"], cl.rbq10_closure, dict(cl.RBQ10_TABLE), ["rb"], ["Q10"],
                                hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    d = m.to_desc()
    assert d.struct_size == ctypes.sizeof(L.ModelDesc)
    assert d.mech == L.EH_MECH_PROGRAM and d.n_params == 2 and d.prog_len == 4 and d.prog_n_forc == 1 and d.prog_n_out == 1
    assert d.prog_out[0] == 31 and list(d.prog_code[:4]) == m.mechanistic_model.program.words()
    assert m.mechanistic_model.name == "rbq10_closure"


def test_what_cannot_be_recorded_is_refused():
    def branches(*, ta, rb):
        return dict(y=rb if ta > 0 else -rb)
    def couples_samples(*, ta, rb):
        return dict(y=rb * ta / np.sum(ta))
    def too_long(*, ta, rb):
        y = rb
        for i in range(70):
            y = y * ta + float(i)
        return dict(y=y)
    def negative_base(*, ta, rb):
        return dict(y=(-2.0) ** ta * rb)
    def not_a_dict(*, ta, rb):
        return rb * ta
    for fn, exc in [(branches, NotImplementedError), (couples_samples, NotImplementedError), (too_long, NotImplementedError),
                    (negative_base, ValueError), (not_a_dict, TypeError)]:
        with pytest.raises(exc):
            P.trace(fn, ["rb"], ["ta"], ["y"])
    with pytest.raises(ValueError):
        P.trace(cl.rbq10_closure, ["rb", "Q10"], ["ta"], ["gpp"])           # not an output
    with pytest.raises(NotImplementedError):
        P.trace(cl.rbq10_closure, [f"p{i}" for i in range(9)], ["ta"], ["reco"])


def test_integer_powers_are_products_and_constants_fold():
    def f(*, x, a):
        return dict(y=a * x ** 3 + x ** -2 + (2.0 * 3.0 + np.exp(0.0)) * x ** 0.5 + x ** 0)
    pg = P.trace(f, ["a"], ["x"], ["y"])
    names = [P.OP_NAMES[c[0]] for c in pg.code]
    assert "pow" not in names and names.count("sqrt") == 1 and 7.0 in pg.consts and 1.0 in pg.consts


def test_bad_programs_are_refused_by_the_library_before_any_device_work():
    """eh_create checks every operand slot against the instruction's position: the kernel indexes a per-lane array with them."""
    m = eh.constructHybridModel(["x0", "x1"], ["ta"], ["reco"], cl.rbq10_closure, dict(cl.RBQ10_TABLE), ["rb"], ["Q10"], hidden_layers=[16])
    for edit, exc, msg in [
            (lambda d: d.prog_code.__setitem__(1, 2 | 12 << 8 | 40 << 16), ValueError, "undefined"),       # operand = a later instruction
            (lambda d: d.prog_code.__setitem__(0, 99), NotImplementedError, "opcode"),
            (lambda d: d.prog_out.__setitem__(0, 200), ValueError, "output"),
            (lambda d: setattr(d, "prog_len", 65), NotImplementedError, "instructions"),
            (lambda d: d.prog_code.__setitem__(0, 1 | 9 << 8 | 13 << 16), ValueError, "undefined"),          # a forcing the program does not declare
            (lambda d: d.prog_code.__setitem__(0, 1 | 2 << 8 | 13 << 16), ValueError, "undefined"),          # a parameter beyond n_params
            (lambda d: d.prog_code.__setitem__(0, 5 | 8 << 8 | 13 << 16), ValueError, "unused operand")]:
        d = m.to_desc()
        edit(d)
        with pytest.raises(exc, match=msg):
            eh.HybridEngine(d, 2, ["reco"], ["rb", "Q10"])


@pytest.mark.parametrize("name", sorted(cl.LOSSES))
def test_custom_loss_program_value_and_derivative(name):
    fn = cl.LOSSES[name]
    pg = P.trace_loss(fn)
    assert pg.params == ("yhat", "y") and len(pg.out) == 1 and len(pg.code) >= 2
    util.register_loss(name, fn)
    rng = np.random.default_rng(3)
    yh, y = rng.normal(size=400), rng.normal(size=400)
    dt = np.dtype(np.float64)
    _, fwd, vjp = ho.MECH["_loss_" + name]
    out, aux = fwd({"yhat": yh, "y": y}, {}, dt)
    direct = fn(yh, y)
    assert np.mean(out["loss"]) == pytest.approx(float(np.mean(direct)), rel=1e-6)            # (constants are held in fp32)
    d = vjp({"yhat": yh, "y": y}, {}, out, aux, {"loss": np.ones(400)}, dt)["yhat"]
    h = 1e-6
    up, _ = fwd({"yhat": yh + h, "y": y}, {}, dt); dn, _ = fwd({"yhat": yh - h, "y": y}, {}, dt)
    fd = (up["loss"] - dn["loss"]) / (2 * h)
    assert (np.abs(d - fd) <= 1e-5 * (1 + np.abs(fd))).mean() > 0.99                           # (kinks of |r| and the Huber switch)


def test_a_loss_that_is_not_a_mean_of_per_sample_terms_is_refused():
    with pytest.raises(NotImplementedError, match="nothing but scaling / shifting can be applied to the mean"):
        P.trace_loss(lambda yh, y: np.sqrt(np.mean((yh - y) ** 2)))
    with pytest.raises(NotImplementedError, match="nothing but scaling"):
        P.trace_loss(lambda yh, y: np.mean((yh - y) ** 2) / np.mean(y * y))
    # what is linear in the mean stays a mean (the reference's own test scales one: test/test_compute_loss.jl:36-47)
    a = P.trace_loss(lambda yh, y: 0.5 * np.mean((yh - y) ** 2))
    b = P.trace_loss(lambda yh, y: np.mean((yh - y) ** 2 * 0.5))
    assert a.code == b.code and a.consts == b.consts
    c = P.trace_loss(lambda yh, y: np.mean(np.abs(yh - y)) / 4 + np.mean((yh - y) ** 2) - 1.0)
    assert len(c.code) >= 5
    with pytest.raises(NotImplementedError, match="only elementwise"):
        P.trace_loss(lambda yh, y: np.sum((yh - y) ** 2))
    with pytest.raises(TypeError):
        P.trace_loss(lambda yh, y: 1.0)


def test_oracle_custom_loss_equals_builtin_mse_and_mae():
    spec, theta, X, f, y = util.rbq10_case(300, "tanh", True, 0.1)
    util.register_loss("my_mse", lambda yh, yy: np.mean((yh - yy) ** 2))
    util.register_loss("my_mae", lambda yh, yy: np.mean(np.abs(yh - yy)))
    for mine, builtin in (("my_mse", "mse"), ("my_mae", "mae")):
        l0, g0, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=builtin)
        l1, g1, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=mine)
        assert l1 == pytest.approx(l0, rel=1e-12) and util.relerr(g1, g0) <= 1e-12


# ---- entries of extra_loss(yhat, ps) that mix predictions / read global parameters: outputs of the model's program -------------------------
def test_mixed_extra_loss_entries_become_outputs_of_the_models_program():
    """program.trace_extra_loss_mixed (src/losses/compute_loss.jl:31-34): the per-sample expression of an entry over several outputs and the
    RAW global parameters, recorded behind the model's own outputs; its values equal the expression on NumPy arrays"""
    fn = lambda yhat, ps: {"balance": 0.25 * np.mean((yhat["gpp"] - yhat["reco"]) ** 2 + yhat["nee"] * yhat["gpp"]),
                           "scaled": np.sum(yhat["gpp"]) * ps.k[0] * 0.5}
    bounds = {g: (cl.FLUX_TABLE[g][1], cl.FLUX_TABLE[g][2]) for g in ("e0", "k")}
    prog, entries = P.trace_extra_loss_mixed(cl.flux_closure, fn, list(cl.FLUX_TABLE), ["sw", "ta", "vpd"], ["nee"], ["e0", "k"], bounds)
    assert prog.outputs == ("nee", "_xl1", "_xl2") and [(e[0], e[1], e[2]) for e in entries] == [("balance", "_xl1", "mean"), ("scaled", "_xl2", "sum")]
    B = 300
    rng = np.random.default_rng(2)
    par = {n: rng.uniform(lo + 0.1 * (hi - lo), hi - 0.1 * (hi - lo), B) for n, (_, lo, hi) in cl.FLUX_TABLE.items()}
    par["e0"] = np.full(B, 171.0); par["k"] = np.full(B, 0.11)          # global parameters: the same for every sample
    frc = _inputs(["sw", "ta", "vpd"], B, 6, 0.0, 30.0)
    val = ho.program_values(prog.as_dict(), par, frc, np.dtype(np.float64))
    out = cl.flux_closure(**frc, **par)
    raw_k = np.log(((0.11 - 0.0) / 0.5) / (1.0 - (0.11 - 0.0) / 0.5))
    assert util.relerr(val[prog.out[1]], 0.25 * ((out["gpp"] - out["reco"]) ** 2 + out["nee"] * out["gpp"])) <= 1e-6
    assert util.relerr(val[prog.out[2]], out["gpp"] * raw_k * 0.5) <= 1e-6
    # a function of the global parameters alone: one value for the batch, whatever reduction is written
    prog2, ent2 = P.trace_extra_loss_mixed(cl.flux_closure, lambda yhat, ps: [np.sum(ps.e0 ** 2) * 1e-4], list(cl.FLUX_TABLE), ["sw", "ta", "vpd"], ["nee", "gpp"], ["e0", "k"], bounds)
    assert ent2[0][2] == "mean" and prog2.outputs == ("nee", "gpp", "_xl1")
    # what is not linear in the reduction, the network's weights, too many outputs: refused
    with pytest.raises(NotImplementedError):
        P.trace_extra_loss_mixed(cl.flux_closure, lambda yhat, ps: [np.mean(yhat["gpp"]) * np.mean(yhat["nee"])], list(cl.FLUX_TABLE), ["sw", "ta", "vpd"], ["nee"], ["e0", "k"], bounds)
    with pytest.raises(NotImplementedError, match="global parameters"):
        P.trace_extra_loss_mixed(cl.flux_closure, lambda yhat, ps: [np.mean(yhat["gpp"]) * ps.alpha], list(cl.FLUX_TABLE), ["sw", "ta", "vpd"], ["nee"], ["e0", "k"], bounds)
    with pytest.raises(NotImplementedError, match="device limit"):
        P.trace_extra_loss_mixed(cl.flux_closure, lambda yhat, ps: [np.mean(yhat["gpp"] * yhat["nee"]), np.mean(yhat["reco"] * yhat["nee"])], list(cl.FLUX_TABLE), ["sw", "ta", "vpd"], ["nee", "gpp"], ["e0", "k"], bounds)
    # entries of one prediction still take the old route (a target of their own with their own per-sample program)
    assert len(P.trace_extra_loss(lambda yhat: [np.sum(np.abs(yhat["nee"]))], ["nee"])) == 1
