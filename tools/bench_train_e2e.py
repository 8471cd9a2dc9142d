"""What a USER runs: wall-clock of `eh.train(model, data, ...)` end to end -- data preparation and upload, every
training step, the two evaluation passes and the host bookkeeping of every epoch, the final predictions -- not `eh_train_step` in
isolation.  The reference's only benchmark is exactly such a call (docs/literate/tutorials/synthetic_respiration_gpu.jl:96-146:
`tune(model, df, cfg)` timed with BenchmarkTools after a warm-up call).

Workloads
  tutorial_small / tutorial_large: the tutorial's two models -- RbQ10, predictors (sw_pot, dsw_pot), hidden [16, 16] /
      [1024, 512, 256, 128, 64], sigmoid, scale_nn_outputs, input_batchnorm -- on 5 000 synthetic rows, batchsize 64, 20 epochs,
      RMSProp(0.01), loss_types [mse, nse], keep_history = false (:79-104)
  headline: the bench's own data set (64 x 65 536 rows, MLP [2,16,16,1] tanh), batchsize 65 536, 10 epochs, Adam(0.01)
For each: the second of two identical calls (the first pays the one-time costs the tutorial's warm-up pays: run-time compilation,
disk cache), a third one with TrainConfig.timing for the split of the epoch loop into steps / evaluation / host, and the same
optimiser steps + per-epoch evaluation forwards in PyTorch-CPU eager autograd (oracle/torch_twin.py: the structurally closest
stand-in for Lux + Zygote this box can run) on the host cores.

  python tools/bench_train_e2e.py            -> one JSON object
Also the evaluation kernel's own roofline entry: `eh_eval` over the headline training split, 16 B and 608 flop per sample."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def _tutorial_model(eh, hidden):
    from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS
    return eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                   hidden_layers=list(hidden), activation="sigmoid", scale_nn_outputs=True, input_batchnorm=True)


def _eager_cpu(spec, cols, batchsize, nepochs, opt_name, lr, threads, max_seconds=20.0):
    """the same number of optimiser steps (shuffled minibatches of `batchsize` over the 80 % training split) and one forward over
    train + validation split per epoch, PyTorch-CPU eager; bounded: extrapolated from the steps that fit into max_seconds"""
    import torch
    from oracle import torch_twin as tt
    torch.set_num_threads(threads)
    n = len(cols["ta"])
    ntr = int(0.8 * n)
    X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
    f, y = {"ta": cols["ta"].astype(np.float32)}, {"reco": cols["reco"].astype(np.float32)}
    from oracle import hybrid_oracle as ho
    theta = torch.tensor(ho.init_theta(spec, 1, np.float32), requires_grad=True)
    opt = (torch.optim.RMSprop([theta], lr=lr, alpha=0.9) if opt_name == "RMSProp" else torch.optim.Adam([theta], lr=lr))
    Xt = torch.as_tensor(X); ft = {k: torch.as_tensor(v) for k, v in f.items()}; yt = {k: torch.as_tensor(v) for k, v in y.items()}
    steps_per_epoch = -(-ntr // batchsize)
    rng = np.random.default_rng(0)
    done_steps, done_evals, t0 = 0, 0, time.perf_counter()
    t_steps = t_eval = 0.0
    for ep in range(nepochs):
        perm = torch.as_tensor(rng.permutation(ntr))
        ts = time.perf_counter()
        for s in range(steps_per_epoch):
            ix = perm[s * batchsize:(s + 1) * batchsize]
            opt.zero_grad()
            tt.loss(spec, theta, Xt[:, ix], {k: v[ix] for k, v in ft.items()}, {k: v[ix] for k, v in yt.items()}).backward()
            opt.step()
            done_steps += 1
            if time.perf_counter() - t0 > max_seconds:
                break
        t_steps += time.perf_counter() - ts
        if time.perf_counter() - t0 > max_seconds:
            break
        te = time.perf_counter()
        with torch.no_grad():
            tt.forward(spec, theta, Xt, ft)
        t_eval += time.perf_counter() - te
        done_evals += 1
    total_steps = nepochs * steps_per_epoch
    per_step = t_steps / max(1, done_steps)
    per_eval = t_eval / max(1, done_evals) if done_evals else float("nan")
    est = total_steps * per_step + (nepochs * per_eval if done_evals else 0.0)
    return {"seconds": est, "measured_steps": done_steps, "of_steps": total_steps, "ms_per_step": 1e3 * per_step,
            "ms_per_epoch_evaluation": 1e3 * per_eval, "threads": threads,
            "what": "PyTorch-CPU eager autograd (oracle/torch_twin.py): the same optimiser steps + one forward over both splits per epoch; "
                    + ("all steps run" if done_steps == total_steps else f"extrapolated from the first {done_steps} steps (bounded to {max_seconds:.0f} s)")}


def _one(eh, name, model, cols, kw, spec, eager_args, threads):
    out = {"workload": name}
    t0 = time.perf_counter(); eh.train(model, cols, **kw); out["first_call_s"] = time.perf_counter() - t0      # pays compilation / caches once, like the tutorial's warm-up
    reps = []
    for _ in range(5):
        t0 = time.perf_counter(); res = eh.train(model, cols, **kw); reps.append(time.perf_counter() - t0)
    out["train_call_s"] = float(np.median(reps)); out["train_call_s_runs"] = reps
    out["train_call_max_over_min"] = max(reps) / min(reps)
    t0 = time.perf_counter(); n_pred = sum(len(v) for v in res.val_obs_pred.values()) + sum(len(v) for v in res.train_obs_pred.values())
    out["predictions_first_read_s"] = time.perf_counter() - t0      # TrainConfig.predictions = "lazy" (default): the tables are made when they are read
    eager = []
    for _ in range(3):
        t0 = time.perf_counter(); eh.train(model, cols, predictions="eager", **kw); eager.append(time.perf_counter() - t0)
    out["train_call_s_eager_predictions"] = float(np.median(eager))
    tm = eh.train(model, cols, timing=True, **kw).timing
    out["epoch_loop_split_s"] = {k: tm[k] for k in ("steps_s", "eval_s", "host_s", "loop_s")}
    out["eval_plus_host_share_of_loop"] = (tm["eval_s"] + tm["host_s"]) / tm["loop_s"]
    out["outside_the_loop_s"] = out["train_call_s"] - tm["loop_s"]          # (timing run: one extra synchronisation per epoch -- the loop itself is a little slower than in the plain call)
    # ... and what is outside, part by part (TrainConfig.timing): split of the caller's columns | engine | interleave + upload | parameters,
    # optimiser, options | evaluation of epoch 0 | final predictions (lazy: none) ; the rest of the call is closing the engine
    out["outside_the_loop_split_s"] = {k: tm[k] for k in ("prepare_s", "engine_s", "upload_s", "setup_s", "initial_eval_s", "final_predictions_s")}
    # the step mode a seeded run takes by default (fused_update = "auto" with random_seed set: bitwise reproducible) against one kernel per
    # step everywhere (float-atomic sums; what bench.py's headline line measures at the engine)
    try:
        t0 = time.perf_counter(); eh.train(model, cols, fused_update=True, **kw); out["train_call_s_fused_update_true"] = time.perf_counter() - t0
        tmf = eh.train(model, cols, fused_update=True, timing=True, **kw).timing
        out["steps_s_fused_update_true"] = tmf["steps_s"]
    except NotImplementedError:          # (no one-kernel step for this model: the layer-wise form)
        out["train_call_s_fused_update_true"] = out["steps_s_fused_update_true"] = None
    out["epochs"], out["best_loss"] = tm["epochs"], float(res.best_loss)
    try:
        out["eager_cpu"] = _eager_cpu(spec, cols, *eager_args, threads)
        out["speedup_vs_eager_cpu"] = out["eager_cpu"]["seconds"] / out["train_call_s"]
    except Exception as e:
        out["eager_cpu"] = {"error": repr(e)}
    return out


def measure(device=0, with_large=True, with_headline=True):
    import easyhybrid_jl_amd as eh
    from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
    from oracle import hybrid_oracle as ho
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(avail, 32)
    runs = []
    cols = make_synth_rbq10(5000, seed=42)
    kw = dict(nepochs=20, batchsize=64, opt=eh.RMSProp(0.01), loss_types=["mse", "nse"], keep_history=False, device=device)
    for name, hidden in (("tutorial_small [16,16]", (16, 16)),) + ((("tutorial_large [1024,512,256,128,64]", (1024, 512, 256, 128, 64)),) if with_large else ()):
        spec = ho.rbq10_spec(hidden, "sigmoid", True); spec.input_batchnorm = True
        runs.append(_one(eh, name, _tutorial_model(eh, hidden), cols, kw, spec, (64, 20, "RMSProp", 0.01), threads))
    eval_roof = None
    if with_headline:
        B, NB = 65536, 64
        colsh = make_synth_rbq10(NB * B, seed=42)
        model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                        hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
        kwh = dict(nepochs=10, batchsize=B, opt=eh.Adam(0.01), loss_types=["mse", "r2"], keep_history=False, device=device)
        runs.append(_one(eh, "headline data set (64 x 65 536 rows), batch 65 536, 10 epochs", model, colsh, kwh, ho.rbq10_spec((16, 16), "tanh", True),
                         (B, 10, "Adam", 0.01), threads))
        # the evaluation kernel on its own: eh_eval over the 80 % training split, metrics only (what evaluate_epoch costs per epoch and split)
        X = np.stack([colsh["sw_pot"], colsh["dsw_pot"]]).astype(np.float32)
        n = int(0.8 * NB * B)
        eng = model.engine(device)
        eng.set_data(eh.EH_SPLIT_TRAIN, X[:, :n], [colsh["ta"][:n]], [colsh["reco"][:n]])
        eng.set_params(model.initialparameters(161803))
        eng.set_option("specialize", 1)
        for _ in range(3):
            eng.eval(eh.EH_SPLIT_TRAIN)
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            eng.eval(eh.EH_SPLIT_TRAIN)
        per = (time.perf_counter() - t0) / reps
        # the kernel alone, by HIP events on the engine's stream around each launch (94 us per call at the driver against a 78 us kernel)
        eng.profile_enable(1)
        for _ in range(reps):
            eng.eval(eh.EH_SPLIT_TRAIN)
        kern = eng.profile_samples()
        eng.profile_enable(False)
        eng.close()
        eval_roof = {"kernel": "eh_step_kernel<..., EVAL> over the headline training split (metrics only: no write-back), one eh_eval call = kernel + 8 x T sums to the host",
                     "samples": n, "ms_per_call": 1e3 * per, "bound": "hbm", "algorithmic_bytes_per_sample": 16, "algorithmic_flop_per_sample": 608,
                     "achieved": 16 * n / per / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": 16 * n / per / 1e9 / 8000.0,
                     "algorithmic_TFLOPs": 608 * n / per / 1e12, "frac_f32_peak": 608 * n / per / 1e12 / 157.3,
                     "kernel_ms_by_events": float(np.median(kern)) if len(kern) else None, "kernel_ms_by_events_min_max": [float(np.min(kern)), float(np.max(kern))] if len(kern) else None,
                     "call_minus_kernel_us": 1e3 * (1e3 * per - float(np.median(kern))) if len(kern) else None,
                     "timing": "host clock around 20 synchronous eh_eval calls (each ends with the copy of the sums to the host); kernel_ms_by_events: HIP events on the "
                               "engine's stream around each of 20 more launches -- the difference is the launch, the synchronisation and the host's reading of the sums"}
    return {"what": "wall-clock of eh.train(...) end to end (median of 3 calls after one warm-up call) and the split of its epoch loop; the same work in PyTorch-CPU eager beside it",
            "host_threads_for_eager": threads, "runs": runs, "eh_eval_roofline": eval_roof}


if __name__ == "__main__":
    print(json.dumps(measure(with_large="--no-large" not in sys.argv, with_headline="--no-headline" not in sys.argv)))
