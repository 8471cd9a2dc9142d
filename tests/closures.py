"""(The literal constants of flux_closure / allops_closure are exactly representable in fp32, so the recorded program --
which holds constants in fp32 -- and the closure on fp64 arrays agree to rounding; rbq10_closure keeps the reference's 0.1.)
User-style mechanistic closures for the EH_MECH_PROGRAM tests: plain NumPy code, `f(**forcings, **params) -> dict`,
as a user of the reference writes `f(; forcing..., params...) -> NamedTuple` (src/models/GenericHybridModel.jl:420-425)."""
import numpy as np


def rbq10_closure(*, ta, rb, Q10):
    """test/test_split_data_train.jl:36-39, written by hand instead of taken from the registry"""
    reco = rb * Q10 ** (0.1 * (ta - 15.0))
    return dict(reco=reco, Q10=Q10, rb=rb)


RBQ10_TABLE = {"rb": (3.0, 0.0, 13.0), "Q10": (2.0, 1.0, 4.0)}


def flux_closure(*, sw, ta, vpd, alpha, gmax, rref, e0, k):
    """light-response GPP with a VPD limitation + Lloyd-Taylor respiration; three outputs"""
    lim = np.where(vpd > 10.0, np.exp(-k * (vpd - 10.0)), 1.0)
    gpp = lim * (alpha * sw * gmax) / (alpha * sw + gmax)
    reco = rref * np.exp(e0 * (1.0 / (10.0 + 46.0) - 1.0 / (np.maximum(ta, -40.0) + 46.0)))
    return dict(nee=reco - gpp, gpp=gpp, reco=reco)


FLUX_TABLE = {"alpha": (0.05, 0.001, 0.2), "gmax": (20.0, 1.0, 60.0), "rref": (3.0, 0.1, 10.0), "e0": (150.0, 50.0, 400.0), "k": (0.05, 0.0, 0.5)}


def allops_closure(*, u, v, a, b, c, d):
    """one of every operation the device program knows"""
    s = 1.0 / (1.0 + np.exp(-a * u))                    # neg exp add div
    t = np.tanh(b * v) + np.sqrt(c + u * u) - np.abs(v - d)
    w = np.sin(a * v) * np.cos(b * u) + np.minimum(c, np.maximum(d, u)) ** 3
    p = (c + 1.5) ** (0.25 * u) + np.log(d + 2.0 + v * v) + np.clip(u * a, -0.5, 0.5) + 2.0 ** (0.125 * v)
    q = np.where(u >= v, s * t, w / (1.0 + p * p)) + np.where(u < 0.25, a, b) + np.where(v <= 0.125, c, -d) * np.square(s)
    return dict(y=q + 0.25 * p, z=s - w)


ALLOPS_TABLE = {"a": (0.8, -2.0, 2.0), "b": (0.5, -2.0, 2.0), "c": (1.0, 0.1, 3.0), "d": (0.7, 0.0, 2.0)}


# name -> (closure, parameter table, forcings): what a fixture / spec that names a closure model needs to rebuild it
CLOSURES = {"rbq10_closure": (rbq10_closure, RBQ10_TABLE, ["ta"]), "flux_closure": (flux_closure, FLUX_TABLE, ["sw", "ta", "vpd"]),
            "allops_closure": (allops_closure, ALLOPS_TABLE, ["u", "v"])}


# custom training losses, as a user writes them for `training_loss = f` (src/losses/loss_fn.jl: f(yhat[mask], y[mask]))
def huber_loss(yhat, y, delta=0.5):
    r = np.abs(yhat - y)
    return np.mean(np.where(r <= delta, 0.5 * r * r, delta * (r - 0.5 * delta)))


def logcosh_loss(yhat, y):
    r = yhat - y
    return np.mean(np.abs(r) + np.log(1.0 + np.exp(-2.0 * np.abs(r))) - np.log(2.0))


def pinball_loss(yhat, y, q=0.75):
    r = y - yhat
    return np.mean(np.maximum(q * r, (q - 1.0) * r))


def relative_sq_loss(yhat, y):
    return ((yhat - y) / (np.abs(y) + 1.0)) ** 2          # per-sample terms: averaged by the engine


LOSSES = {"huber": huber_loss, "logcosh": logcosh_loss, "pinball": pinball_loss, "relative_sq": relative_sq_loss}
