"""Synthetic workloads of BASELINE.json (distributions of the reference's own fixtures; NumPy's
PCG64 stream -- Julia's MersenneTwister(42) stream of make_synth_df is not reproducible).

  RbQ10      test/test_split_data_train.jl:15-31 (make_synth_df), parameters :42-45
  Expo2Pool  inputs as projects/ExpoHybrid/ExpoHybridEstim.jl:39-46, 8 predictors, build-defined target
"""
from __future__ import annotations

import numpy as np

RBQ10_PARAMS = {"rb": (3.0, 0.0, 13.0), "Q10": (2.0, 1.0, 4.0)}
EXPO2POOL_PARAMS = {"R0a": (1.0, 0.0, 8.0), "ka": (0.05, 0.0, 0.2), "R0b": (0.5, 0.0, 8.0), "kb": (0.02, 0.0, 0.2)}


def make_synth_rbq10(n: int, seed: int = 42, nan_frac: float = 0.0):
    """-> dict of float32 columns ta, sw_pot, dsw_pot, reco."""
    rng = np.random.default_rng(seed)
    ta = 10 + 10 * rng.standard_normal(n)
    sw_pot = np.abs(50 + 20 * rng.standard_normal(n))
    dsw_pot = np.concatenate([[0.0], np.diff(sw_pot)])
    rb_true = 3.0 + 0.02 * (sw_pot - sw_pot.mean())
    reco = rb_true * 2.0 ** (0.1 * (ta - 15.0)) + 0.1 * rng.standard_normal(n)
    if nan_frac > 0:
        reco[rng.random(n) < nan_frac] = np.nan
    return {k: v.astype(np.float32) for k, v in dict(ta=ta, sw_pot=sw_pot, dsw_pot=dsw_pot, reco=reco).items()}


def make_synth_expo2pool(n: int, seed: int = 42, nan_frac: float = 0.0):
    """-> dict of float32 columns x0..x7, T, Resp_obs."""
    rng = np.random.default_rng(seed)
    X = rng.random((8, n))
    T = rng.random(n) * 40 - 10
    sm = X[0] * 0.8 + 0.1
    R0a = 1.1 * np.exp(-8.0 * (sm - 0.6) ** 2)
    R0b = 0.3 + 0.4 * X[1]
    resp = R0a * np.exp(0.07 * T) + R0b * np.exp(0.02 * T)
    resp = resp + 0.05 * resp.mean() * rng.standard_normal(n)
    if nan_frac > 0:
        resp[rng.random(n) < nan_frac] = np.nan
    cols = {f"x{i}": X[i].astype(np.float32) for i in range(8)}
    cols["T"] = T.astype(np.float32)
    cols["Resp_obs"] = resp.astype(np.float32)
    return cols


RS6_PARAMS = {**{f"Rb_{c}": (1.0, 0.0, 5.0) for c in ("het", "root", "myc")},
              **{f"Q10_{c}": (2.0 + 0.3 * i, 1.0, 4.0) for i, c in enumerate(("het", "root", "myc"))}}


def make_synth_fluxnet32(n: int, seed: int = 42, nan_frac: float = 0.0):
    """BASELINE.json configs[4] stand-in: 32 covariates x0..x31, air temperature ta (the forcing the
    Rs_components model reads), and a three-component soil-respiration target R_soil whose base rates
    depend on the covariates.  -> dict of float32 columns."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((32, n)).astype(np.float32) * 0.5
    ta = (10 + 10 * rng.standard_normal(n)).astype(np.float32)
    e = 0.1 * (ta - 15.0)
    rb = [1.0 + 0.8 * np.tanh(X[3 * c] + 0.5 * X[3 * c + 1]) for c in range(3)]
    y = sum(rb[c] * np.power(1.6 + 0.4 * c, e) for c in range(3)).astype(np.float32)
    y *= (1 + 0.05 * rng.standard_normal(n)).astype(np.float32)
    if nan_frac > 0:
        y[rng.random(n) < nan_frac] = np.nan
    cols = {f"x{i}": X[i] for i in range(32)}
    cols["ta"] = ta
    cols["R_soil"] = y
    return cols


def make_synth_fluxnet32_3f(n: int, seed: int = 42, nan_frac: float = 0.0):
    """BASELINE.json configs[4] as stated: 32 covariates x0..x31 ~ N(0, 0.5), three forcings -- air temperature ta, an
    irradiance-like sw_in and a vapour-pressure-deficit-like vpd, both U(0.2, 1.2) -- and the target of the build-defined
    three-forcing soil-respiration model Rs_components3F (R_het + sw_in R_root + vpd R_myc) whose base rates depend on the
    covariates, 5 % multiplicative noise.  -> dict of float32 columns."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((32, n)).astype(np.float32) * 0.5
    ta = (10 + 10 * rng.standard_normal(n)).astype(np.float32)
    sw = (0.2 + rng.random(n)).astype(np.float32)
    vpd = (0.2 + rng.random(n)).astype(np.float32)
    e = 0.1 * (ta - 15.0)
    rb = [1.0 + 0.8 * np.tanh(X[3 * c] + 0.5 * X[3 * c + 1]) for c in range(3)]
    w = [1.0, sw, vpd]
    y = sum(w[c] * rb[c] * np.power(1.6 + 0.4 * c, e) for c in range(3)).astype(np.float32)
    y *= (1 + 0.05 * rng.standard_normal(n)).astype(np.float32)
    if nan_frac > 0:
        y[rng.random(n) < nan_frac] = np.nan
    cols = {f"x{i}": X[i] for i in range(32)}
    cols.update(ta=ta, sw_in=sw, vpd=vpd, R_soil=y)
    return cols
