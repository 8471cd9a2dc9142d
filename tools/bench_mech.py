"""Bandwidth of the stand-alone mechanistic + loss + VJP kernel (eh_mech_loss_vjp) against the HBM roof.

    python tools/bench_mech.py [--batch 16777216] [--steps 50] [--mech rbq10|expo2pool|fluxpart]

Algorithmic bytes per sample: 4 (K + F + T) read + 4 K written (inputs once, d loss / d o once).
Timing: torch events on the stream the engine launches on (eh_set_stream), around `steps` back-to-back calls with the
counts of valid targets handed in (one streaming kernel + a one-block finish kernel per call) and, separately, with the
counting pass (a second read of the targets)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


CONFIGS = {
    "rbq10": ("RbQ10", "RBQ10", ["rb"], ["Q10"], ["ta"], ["reco"]),
    "expo2pool": ("Expo2Pool", {"R0a": (1.0, 0.0, 8.0), "ka": (0.05, 0.0, 0.2), "R0b": (0.5, 0.0, 8.0), "kb": (0.02, 0.0, 0.2)},
                  ["R0a", "ka", "R0b", "kb"], [], ["T"], ["Resp_obs"]),
    "fluxpart": ("FluxPartModelQ10", {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}, ["RUE", "Rb"], ["Q10"],
                 ["SW_IN", "TA"], ["NEE", "RECO"]),
}


def measure(mech="rbq10", batch=1 << 24, steps=50, stagger=0, blocks=0, tiles=0):
    """-> dict (one JSON line of this tool).  Needs a GPU; creates its own engine and a named torch stream."""
    import torch
    import easyhybrid_jl_amd as eh
    from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS
    fn, tab, neural, glob, forc, targ = CONFIGS[mech]
    fn = getattr(eh, fn)
    tab = dict(RBQ10_PARAMS) if tab == "RBQ10" else tab
    model = eh.constructHybridModel(["x0", "x1"], forc, targ, fn, tab, neural, glob, hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    eng = model.engine(0)
    prev = torch.cuda.current_stream()
    stream = torch.cuda.Stream()             # (a named stream: eh_set_stream(NULL) would mean the engine's own)
    torch.cuda.set_stream(stream)
    try:
        eng.set_stream(stream.cuda_stream)
        eng.set_params(model.initialparameters(1))
        if blocks:
            eng.set_option("mech_blocks", blocks)      # cap on the streaming kernel's workgroups (then a grid-striding launch)
        if tiles:
            eng.set_option("mech_tiles", tiles)        # consecutive 1 024-sample tiles per workgroup (default 1)
        B, K, F, T = batch, len(neural), len(forc), len(targ)
        g = torch.Generator(device="cuda").manual_seed(0)
        nplane = [0]

        def plane(rows, fill):
            """a [rows][B] array whose start is shifted by (its ordinal x stagger) bytes: experiment on how the planes' relative
            placement meets the HBM channel interleave (stagger = 0: wherever the allocator puts them)"""
            pad = (nplane[0] * stagger) // 4
            nplane[0] += 1
            buf = torch.empty(rows * B + pad, device="cuda")
            v = buf[pad:].view(rows, B)
            fill(v)
            return v

        o = plane(K, lambda v: v.normal_(generator=g))
        fr = [plane(1, lambda v: v.uniform_(1, 26, generator=g))[0] for _ in forc]
        ys = []
        for _ in targ:
            y = plane(1, lambda v: v.uniform_(0.5, 6.5, generator=g))[0]
            y[torch.rand(B, device="cuda", generator=g) < 0.05] = float("nan")
            ys.append(y)
        d_o = plane(K, lambda v: v.zero_())
        nvalid = [int((~torch.isnan(y)).sum()) for y in ys]
        fp, tp = [t.data_ptr() for t in fr], [t.data_ptr() for t in ys]
        bytes_alg = 4 * (K + F + T) * B + 4 * K * B

        def timed(counts):
            for _ in range(5):
                eng.mech_loss_vjp(B, o.data_ptr(), fp, tp, d_o.data_ptr(), n_valid_in=counts, results=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                eng.mech_loss_vjp(B, o.data_ptr(), fp, tp, d_o.data_ptr(), n_valid_in=counts, results=False)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / steps

        ms_known = timed(nvalid)
        ms_count = timed(None)
        loss, gg, nv = eng.mech_loss_vjp(B, o.data_ptr(), fp, tp, d_o.data_ptr())
    finally:
        torch.cuda.set_stream(prev)
        eng.close()
    return {"kernel": "eh_mech_vjp_kernel<4, %s> + eh_mech_finish_kernel (one eh_mech_loss_vjp call, counts of valid targets handed in)" % mech,
            "mech": mech, "batch": B, "stagger_bytes": stagger, "mech_blocks": blocks or "default", "mech_tiles": tiles or "default", "K": K, "F": F, "T": T, "bound": "hbm",
            "algorithmic_bytes_per_sample": bytes_alg // B, "ms_per_call": ms_known, "achieved": bytes_alg / ms_known / 1e6, "peak": 8000.0,
            "unit": "GB/s", "frac": bytes_alg / ms_known / 1e6 / 8000, "ms_per_call_with_counting_pass": ms_count,
            "GBps_with_counting_pass": (bytes_alg + 4 * T * B) / ms_count / 1e6, "samples_per_s": B / ms_known * 1e3,
            "timing": "torch events on the stream the engine launches on, %d back-to-back calls" % steps, "loss": loss, "n_valid": nv}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1 << 24)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--mech", default="rbq10", choices=sorted(CONFIGS))
    ap.add_argument("--stagger", type=int, default=0, help="shift the start of plane k by k x this many bytes (multiple of 16)")
    ap.add_argument("--blocks", type=int, default=0, help="workgroups of the streaming kernel (the library's default when 0)")
    ap.add_argument("--tiles", type=int, default=0, help="consecutive 1 024-sample tiles per workgroup (the library's default when 0)")
    args = ap.parse_args()
    print(json.dumps(measure(args.mech, args.batch, args.steps, args.stagger, args.blocks, args.tiles)))


if __name__ == "__main__":
    main()
