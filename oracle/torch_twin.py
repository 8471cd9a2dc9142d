"""PyTorch-CPU autograd twin of oracle/hybrid_oracle.py  --  TEST INFRASTRUCTURE ONLY.

Second, independent derivation of the gradients: the same forward
written op-by-op with torch tensors, differentiated by autograd's tape.  It is the closest
structural analogue of the reference's Lux + Zygote path that can run in the build container
(BLAS GEMMs, un-fused broadcasts, boolean-mask gather exactly as loss_fn.jl:61-63 does
`mean(abs2, yhat[mask] .- y[mask])`).  Used (a) to validate the hand VJP in tests and (b) as the
"eager autograd" CPU baseline figure quoted in DESIGN.md.  Never imported by the product path.
"""
from __future__ import annotations

import numpy as np
import torch

from . import hybrid_oracle as ho


def _act(name, z):
    if name == "tanh":
        return torch.tanh(z)
    if name == "sigmoid":
        return torch.sigmoid(z)
    if name == "relu":
        return torch.relu(z)
    if name == "swish":
        return z * torch.sigmoid(z)
    if name == "identity":
        return z
    raise ValueError(name)


def _round_bf16(x):
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


class _RoundedAct(torch.autograd.Function):
    """spec.precision == "bf16_fwd": a hidden layer hands on bf16(act(z)) and its derivative is taken from that stored value
    (tanh: 1 - h^2, sigmoid: h (1 - h), relu: h > 0), the rounding itself treated as the identity."""

    @staticmethod
    def forward(ctx, z, name):
        h = _round_bf16(_act(name, z))
        ctx.save_for_backward(h)
        ctx.name = name
        return h

    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        n = ctx.name
        d = 1 - h * h if n == "tanh" else h * (1 - h) if n == "sigmoid" else (h > 0).to(h.dtype) if n == "relu" else torch.ones_like(h)
        return g * d, None


class _RoundedGrad(torch.autograd.Function):
    """spec.precision == "bf16": the identity forward; the gradient passing back through it is rounded to bfloat16.  Placed on the
    pre-activation W h + b of a Dense layer, it makes everything the layer's delta feeds -- dW = d h^T, dh = W^T d and the bias gradient
    d 1 -- take the rounded delta."""

    @staticmethod
    def forward(ctx, x):
        return x

    @staticmethod
    def backward(ctx, g):
        return _round_bf16(g)


def _mech(spec, par, frc):
    m = spec.mech
    if m == "rbq10":
        return {"reco": par["rb"] * par["Q10"] ** (0.1 * (frc["ta"] - 15.0))}
    if m == "expo":
        return {"Resp_obs": par["Resp0"] * torch.exp(par["k"] * frc["T"])}
    if m == "linear":
        return {"obs": par["alpha"] * frc["x"] + par["beta"]}
    if m == "expo2pool":
        return {"Resp_obs": par["R0a"] * torch.exp(par["ka"] * frc["T"]) + par["R0b"] * torch.exp(par["kb"] * frc["T"])}
    if m == "rs_components":
        e = 0.1 * (frc["ta"] - 15.0)
        return {"R_soil": sum(par[f"Rb_{c}"] * par[f"Q10_{c}"] ** e for c in ("het", "root", "myc"))}
    if m == "rs_components3f":
        e = 0.1 * (frc["ta"] - 15.0)
        return {"R_soil": par["Rb_het"] * par["Q10_het"] ** e + frc["sw_in"] * par["Rb_root"] * par["Q10_root"] ** e + frc["vpd"] * par["Rb_myc"] * par["Q10_myc"] ** e}
    if m == "fluxpart":
        gpp = frc["SW_IN"] * par["RUE"] / 12.011
        reco = par["Rb"] * par["Q10"] ** (0.1 * (frc["TA"] - 15.0))
        return {"NEE": reco - gpp, "GPP": gpp, "RECO": reco}
    raise ValueError(m)


def forward(spec: ho.HybridSpec, theta: torch.Tensor, X, forcings):
    dt = theta.dtype
    off = 0
    h = torch.as_tensor(X, dtype=dt)
    if getattr(spec, "input_batchnorm", False):            # train-mode batch statistics, biased variance, eps = 1e-5
        h = (h - h.mean(dim=1, keepdim=True)) / torch.sqrt(h.var(dim=1, unbiased=False, keepdim=True) + 1e-5)
    X0, outs = h, []
    for k_net, (rows, dims) in enumerate(spec.net_list):
        bf = getattr(spec, "precision", "f32") in ("bf16_fwd", "bf16")
        bfb = getattr(spec, "precision", "f32") == "bf16"
        h = _round_bf16(X0[rows]) if bf else X0[rows]
        for li, (o, i) in enumerate(dims):
            W = theta[off:off + o * i].reshape(i, o).T          # column-major (out,in)
            if bf:
                W = W + (_round_bf16(W) - W).detach()           # value bf16(W), gradient passed straight through to W
            off += o * i
            b = theta[off:off + o]
            off += o
            z = _RoundedGrad.apply(W @ h + b[:, None]) if bfb else W @ h + b[:, None]
            if li == len(dims) - 1: h = z
            elif bf: h = _RoundedAct.apply(z, spec.act_of(k_net, li))
            else: h = _act(spec.act_of(k_net, li), z)
        outs.append(h)
    h = torch.cat(outs, dim=0)
    par = {}
    for k, n in enumerate(spec.neural):
        par[n] = (spec.lo(n) + (spec.hi(n) - spec.lo(n)) * torch.sigmoid(h[k])) if spec.scale_nn_outputs else h[k]
    for j, g in enumerate(spec.glob):
        par[g] = spec.lo(g) + (spec.hi(g) - spec.lo(g)) * torch.sigmoid(theta[off + j:off + j + 1])
    for f in spec.fixed:
        par[f] = torch.full((1,), spec.default(f), dtype=dt)
    frc = {k: torch.as_tensor(v, dtype=dt) for k, v in forcings.items()}
    return _mech(spec, par, frc)


def loss(spec, theta, X, forcings, targets, kind="mse", unnormalised=False):
    """unnormalised (mse on one target only): the SUM of squared residuals -- what the engine back-propagates before it divides the
    finished gradient by n; matters where the backward pass rounds (precision = "bf16")"""
    out = forward(spec, theta, X, forcings)
    tot = 0
    for t in spec.targets:
        y = torch.as_tensor(targets[t], dtype=theta.dtype)
        m = ~torch.isnan(y)
        if int(m.sum()) == 0:
            continue
        a, b = out[t][m], y[m]
        if kind == "mse" and unnormalised:
            tot = tot + torch.sum((a - b) ** 2)
        elif kind == "mse":
            tot = tot + torch.mean((a - b) ** 2)                       # loss_fn.jl:61-63
        elif kind == "rmse":
            tot = tot + torch.sqrt(torch.mean((a - b) ** 2))           # :58-60
        elif kind == "mae":
            tot = tot + torch.mean(torch.abs(a - b))                   # :64-66
        elif kind == "nseLoss":
            tot = tot + torch.sum((a - b) ** 2) / torch.sum((b - torch.mean(b)) ** 2)   # :79-81
        else:
            raise ValueError(kind)
    return tot


def loss_and_grad(spec, theta_np, X, forcings, targets, dtype=torch.float64, kind="mse"):
    theta = torch.tensor(np.asarray(theta_np), dtype=dtype, requires_grad=True)
    # precision = "bf16" on a one-target mse model: the deltas are rounded in the un-normalised scale (oracle/hybrid_oracle.py, HybridSpec.precision)
    un = getattr(spec, "precision", "f32") == "bf16" and len(spec.targets) == 1 and kind == "mse"
    l = loss(spec, theta, X, forcings, targets, kind, unnormalised=un)
    if not torch.is_tensor(l):
        return 0.0, np.zeros(theta.numel())
    l.backward()
    n = float(np.sum(~np.isnan(np.asarray(targets[spec.targets[0]])))) if un else 1.0
    return float(l.detach()) / n, theta.grad.numpy().copy() / n


def train_step_timed(spec, theta_np, X, forcings, targets, n_steps, lr=0.01, threads=None):
    """Eager autograd + torch.optim.Adam steps on one batch; returns seconds per step."""
    import time
    if threads:
        torch.set_num_threads(threads)
    theta = torch.tensor(np.asarray(theta_np, np.float32), requires_grad=True)
    opt = torch.optim.Adam([theta], lr=lr, betas=(0.9, 0.999), eps=1e-8)
    Xt = torch.as_tensor(X)
    ft = {k: torch.as_tensor(v) for k, v in forcings.items()}
    tt = {k: torch.as_tensor(v) for k, v in targets.items()}
    for _ in range(3):
        opt.zero_grad(); loss(spec, theta, Xt, ft, tt).backward(); opt.step()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        opt.zero_grad(); loss(spec, theta, Xt, ft, tt).backward(); opt.step()
    return (time.perf_counter() - t0) / n_steps
