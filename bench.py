#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: training samples/s + ms/step, RbQ10 hybrid
([2,16,16,1] MLP -> rb, Q10 global), batch 65 536 per GPU, fp32, synthetic data resident in HBM.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: this script spawns its own N ranks,
                                                           one process per GPU, BEFORE anything touches a GPU -- spawn_ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: forward + mechanistic model + masked MSE + VJP
+ Adam update.  On one GPU that is ONE kernel per step (fused-update mode: the update of step s is
applied in the prologue of step s+1's kernel and flushed at the end of the timed region); for N > 1
it is the step kernel, the partial reduction, one RCCL all-reduce of n_theta+2 raw sums and the
Adam kernel.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

BATCH = 65536
NBATCHES = 64                 # distinct resident batches per GPU (67 MB of 16-byte records)
FLOP_PER_SAMPLE = 1824        # 3 x 2 x (2*16 + 16*16 + 16*1)    SURVEY.md section 8(d)
BYTES_PER_SAMPLE = 16         # 4 x (P + F + T) floats; the mask is the NaN in the target, so no +T byte
PEAK_F32_TFLOPS = 157.3       # MI355X_MICROARCH.md: f32 MFMA == f32 vector peak
PEAK_HBM_GBPS = 8000.0


EX_RENDEZVOUS = 75            # a rank could not join the rendezvous (the port was taken between the probe and rank 0's bind): retried once


def spawn_ranks(nprocs, cmd, env=None, capture=False, timeout=None):
    """spawn_ranks_once, and once more on a fresh port when a rank reports that the rendezvous itself failed (exit code
    EX_RENDEZVOUS): the free port is found by bind(0) / close, so another job on the box can take it before rank 0 binds."""
    out = spawn_ranks_once(nprocs, cmd, env=env, capture=capture, timeout=timeout)
    if (out[0] if capture else out) == EX_RENDEZVOUS:
        print("bench.py: the rendezvous failed (port taken?); one more try on a fresh port", file=sys.stderr, flush=True)
        out = spawn_ranks_once(nprocs, cmd, env=env, capture=capture, timeout=timeout)
    return out


def spawn_ranks_once(nprocs, cmd, env=None, capture=False, timeout=None):
    """The launcher `python bench.py --gpus N` is its own: start `cmd` N times, one process per rank (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment, rendezvous on 127.0.0.1 at a free port), wait for all of them,
    and stop the others (by their exact PIDs) as soon as one fails -- a rank that died would leave its peers in a barrier.
    The calling process never initialises a GPU (a process that has must not start programs on the GPU boxes).
    Returns the largest exit code; capture=True: (code, rank 0's stdout)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(nprocs), LOCAL_WORLD_SIZE=str(nprocs), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                EH_BENCH_LAUNCHER="bench.py:spawn_ranks")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL and the peer-to-peer exchange need it on this driver
    procs = []
    for r in range(nprocs):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if (capture and r == 0) else None))
    t0, rc, chunks, reader = time.monotonic(), 0, [], None
    try:
        if capture:                                          # a thread drains rank 0's pipe: the wait loop below must keep running
            import threading
            reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
            reader.start()
        live, stopped = list(procs), set()
        while live:
            for p in list(live):
                c = p.poll()
                if c is None:
                    continue
                live.remove(p)
                if c != 0 and p.pid not in stopped:          # (a rank this loop stopped reports the signal, not a failure of its own)
                    rc = max(rc, c if c > 0 else 128 - c)
                    for q in live:                           # the exact processes started above, nothing by pattern
                        stopped.add(q.pid)
                        q.terminate()
            if timeout is not None and time.monotonic() - t0 > timeout and live:
                rc = max(rc, 124)
                for q in live:
                    stopped.add(q.pid)
                    q.terminate()
                timeout = None
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if reader is not None:
        reader.join(10)
    return (rc, b"".join(chunks).decode(errors="replace")) if capture else rc


def _cpu_model():
    try:
        models = {}
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                m = line.split(":", 1)[1].strip()
                models[m] = models.get(m, 0) + 1
        return "; ".join(f"{v} x {k}" for k, v in models.items()) or "unknown"
    except Exception:
        return "unknown"


def cpu_baseline(batch, seconds=12.0):
    """The CPU oracle (C port in its blocked, vectorised form -- oracle/eh_oracle_fast.c -- OpenMP over sample blocks) timed on this box's host cores, on a
    bounded sample of the same workload: steps of `batch` samples for ~`seconds` of CPU work.  The
    thread count is the best of a short sweep (all cores is not the fastest on a 256-thread host)."""
    from oracle import c_oracle as co
    from oracle import hybrid_oracle as ho
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    spec = ho.rbq10_spec((16, 16), "tanh", True)
    X, f, y = ho.make_synth_rbq10(4 * batch, 42)
    theta = ho.init_theta(spec, 1, np.float32)
    best, per, sweep = None, None, {}
    for nt in sorted({min(avail, k) for k in (8, 16, 32, 64, 128, 256, avail)}):
        co.train_steps(spec, theta, X, f, y, batch, 1, nthreads=nt, fast=True)          # warm-up (page-in, thread pool)
        t0 = time.perf_counter()
        co.train_steps(spec, theta, X, f, y, batch, 2, nthreads=nt, fast=True)
        t = (time.perf_counter() - t0) / 2
        sweep[str(nt)] = batch / t                                                       # samples/s at this thread count (two steps: a sweep, not a measurement)
        if per is None or t < per:
            best, per = nt, t
    n, chunk, t0 = 0, max(2, min(64, int(1.0 / max(per, 1e-4)))), time.perf_counter()
    while time.perf_counter() - t0 < seconds:              # bounded by wall time, whatever the sweep estimated
        co.train_steps(spec, theta, X, f, y, batch, chunk, nthreads=best, fast=True)
        n += chunk
    dt = time.perf_counter() - t0
    out = {"value": batch * n / dt, "unit": "samples/s", "cores": best, "kind": "port",
           "sample": f"{n} Adam steps of batch {batch} (RbQ10 [2,16,16,1], fp32) = {dt:.1f} s of the C port in its blocked form "
                     f"(oracle/eh_oracle_fast.c: 16 samples per SIMD block, AVX2, rational tanh, vector exp / log; the scalar checker "
                     f"oracle/eh_oracle.c is not what is timed), OpenMP over blocks on {best} of {avail} host threads (fastest of a short sweep)",
           "ms_per_step": 1e3 * dt / n,
           # "core count and CPU model stated"; and the sweep the thread count came from -- the last entry is ALL host threads
           "cpu_model": _cpu_model(), "host_threads": avail, "thread_sweep_samples_per_s": sweep, "all_host_threads_samples_per_s": sweep.get(str(avail))}
    # PyTorch-CPU eager autograd + Adam on the same batch -- the structurally closest stand-in for the
    # reference's Lux + Zygote step this box can run (BLAS GEMMs, un-fused broadcasts, tape, boolean-mask gather); ~3 s
    try:
        from oracle import torch_twin as tt
        import torch
        eager = None
        for nt in sorted({min(avail, k) for k in (8, 32, 64)}):
            sec = tt.train_step_timed(spec, theta, X[:, :batch], {k: v[:batch] for k, v in f.items()}, {k: v[:batch] for k, v in y.items()}, 8, threads=nt)
            if eager is None or sec < eager[1]:
                eager = (nt, sec)
        out["eager"] = {"value": batch / eager[1], "unit": "samples/s", "cores": eager[0], "kind": "port", "ms_per_step": 1e3 * eager[1],
                        "sample": f"8 steps of batch {batch}, torch {torch.__version__} CPU eager autograd + torch.optim.Adam (oracle/torch_twin.py), best of 8 / 32 / 64 threads"}
    except Exception as e:
        out["eager"] = {"error": repr(e)}
    # the primary figure is the FASTER of the two ports; the other one stays beside it
    if "value" in out.get("eager", {}) and out["eager"]["value"] > out["value"]:
        c_port = {k: out[k] for k in ("value", "unit", "cores", "kind", "sample", "ms_per_step")}
        eager = out.pop("eager")
        out.update(eager)
        out["c_port"] = c_port
        out["which"] = "PyTorch-CPU eager (the faster of the two CPU ports on this host); the blocked C / OpenMP port under c_port"
    else:
        out["which"] = "blocked C / OpenMP port, oracle/eh_oracle_fast.c (the faster of the two CPU ports on this host); PyTorch-CPU eager under eager"
    # BASELINE.json configs[0]: "batch = 1024, CPU reference path" (B = 1 024 and 65 536, >= 50 steps, median / p10 / p90)
    try:
        out["c1_batch_1024"] = cpu_baseline_c1(spec, theta, X, f, y, avail)
    except Exception as e:
        out["c1_batch_1024"] = {"error": repr(e)}
    return out


def cpu_baseline_c1(spec, theta, X, f, y, avail, batch=1024, nsteps=60):
    """configs[0] on the host cores: `nsteps` single Adam steps of batch 1 024 per port, per-step wall time -> median / p10 / p90"""
    from oracle import c_oracle as co
    res = {"batch": batch, "steps": nsteps}

    def stats(ts):
        ts = np.sort(np.asarray(ts))
        return {"median_ms": 1e3 * float(np.median(ts)), "p10_ms": 1e3 * float(ts[int(0.1 * (len(ts) - 1))]), "p90_ms": 1e3 * float(ts[int(0.9 * (len(ts) - 1))]),
                "samples_per_s_at_median": batch / float(np.median(ts))}
    best = None
    for nt in sorted({min(avail, k) for k in (1, 4, 8, 16, 32)}):            # (1 024 samples: few threads win)
        for _ in range(5):
            co.train_steps(spec, theta, X, f, y, batch, 1, nthreads=nt, fast=True)
        ts = []
        for _ in range(nsteps):
            t0 = time.perf_counter(); co.train_steps(spec, theta, X, f, y, batch, 1, nthreads=nt, fast=True); ts.append(time.perf_counter() - t0)
        st = stats(ts)
        if best is None or st["median_ms"] < best[1]["median_ms"]:
            best = (nt, st)
    res["c_port"] = {**best[1], "cores": best[0], "kind": "port", "what": "C port, blocked form (oracle/eh_oracle_fast.c), one Adam step per call (fastest thread count of 1 / 4 / 8 / 16 / 32)"}
    try:
        import torch
        from oracle import torch_twin as tt
        Xb = torch.as_tensor(X[:, :batch]); fb = {k: torch.as_tensor(v[:batch]) for k, v in f.items()}; yb = {k: torch.as_tensor(v[:batch]) for k, v in y.items()}
        beste = None
        for nt in sorted({min(avail, k) for k in (1, 4, 8)}):
            torch.set_num_threads(nt)
            th = torch.tensor(np.asarray(theta, np.float32), requires_grad=True)
            opt = torch.optim.Adam([th], lr=0.01)
            ts = []
            for i in range(nsteps + 5):
                t0 = time.perf_counter()
                opt.zero_grad(); tt.loss(spec, th, Xb, fb, yb).backward(); opt.step()
                if i >= 5:
                    ts.append(time.perf_counter() - t0)
            st = stats(ts)
            if beste is None or st["median_ms"] < beste[1]["median_ms"]:
                beste = (nt, st)
        res["eager"] = {**beste[1], "cores": beste[0], "kind": "port", "what": f"torch {torch.__version__} CPU eager autograd + Adam (fastest of 1 / 4 / 8 threads)"}
    except Exception as e:
        res["eager"] = {"error": repr(e)}
    return res


def parity_replay(model, eng_factory, cols, X, B, nsteps=20):
    """Checker leg (after the timed region, never measured): the same `nsteps` Adam steps on the bench's own un-scaled inputs and
    initial parameters, on the GPU engine and on the plain-C oracle port, then the loss of the next batch on both.  Ties the
    headline workload itself -- raw sw_pot ~ 50, i.e. the first layer's tanh saturated -- to the oracle."""
    from oracle import c_oracle as co
    from oracle import hybrid_oracle as ho
    import easyhybrid_jl_amd as eh
    n = (nsteps + 1) * B
    spec = ho.rbq10_spec((16, 16), "tanh", True)
    theta0 = np.asarray(model.initialparameters(161803), np.float32)
    f, y = {"ta": cols["ta"][:n]}, {"reco": cols["reco"][:n]}
    th_ref, _ = co.train_steps(spec, theta0, X[:, :nsteps * B], {"ta": f["ta"][:nsteps * B]}, {"reco": y["reco"][:nsteps * B]}, B, nsteps, nthreads=16)
    l_ref, _, _ = co.loss_and_grad(spec, th_ref, X[:, nsteps * B:n], {"ta": f["ta"][nsteps * B:]}, {"reco": y["reco"][nsteps * B:]}, nthreads=16)
    eng = eng_factory()
    try:
        eng.set_params(theta0)
        # where the 1e-5 bar of the north_star applies: ONE loss + gradient on the headline batch itself, against the fp64 oracle
        l1, g1, _ = eng.loss_and_grad(eh.EH_SPLIT_TRAIN, 0, B)
        l64, g64, _ = ho.loss_and_grad(spec, theta0.astype(np.float64), X[:, :B], {"ta": f["ta"][:B]}, {"reco": y["reco"][:B]})
        step0 = {"loss_rel_diff": abs(l1 - l64) / abs(l64), "grad_relerr": float(np.max(np.abs(g1 - g64)) / np.max(np.abs(g64))),
                 "what": "loss and gradient of batch 0 at the initial parameters, GPU engine vs oracle/hybrid_oracle.py in fp64 (bar: 1e-5)"}
        eng.opt_init("Adam", 0.01, 0.9, 0.999, 1e-8)
        for s in range(nsteps):
            eng.train_step(s * B, B, want_loss=False)
        th = eng.get_params()
        l_gpu, _, _ = eng.loss_and_grad(eh.EH_SPLIT_TRAIN, nsteps * B, B)
    finally:
        eng.close()
    # anchored in fp64: the same steps in the fp64 oracle are what both fp32 trajectories drift from
    th64, _ = ho.train_steps(spec, theta0.astype(np.float64), X[:, :nsteps * B], {"ta": f["ta"][:nsteps * B]}, {"reco": y["reco"][:nsteps * B]},
                             [(s * B, B) for s in range(nsteps)], dtype=np.float64)
    l64, _, _ = ho.loss_and_grad(spec, th64, X[:, nsteps * B:n], {"ta": f["ta"][nsteps * B:]}, {"reco": y["reco"][nsteps * B:]})
    return {"steps": nsteps, "final_loss": l_gpu, "oracle_final_loss": l_ref, "fp64_final_loss": float(l64),
            "loss_distance_from_fp64": {"engine": abs(l_gpu - l64) / abs(l64), "c_fp32_port": abs(l_ref - l64) / abs(l64)},
            "theta_max_distance_from_fp64": {"engine": float(np.max(np.abs(th - th64))), "c_fp32_port": float(np.max(np.abs(th_ref - th64)))},
            "theta_mean_distance_from_fp64": {"engine": float(np.mean(np.abs(th - th64))), "c_fp32_port": float(np.mean(np.abs(th_ref - th64)))},
            "rel_diff": abs(l_gpu - l_ref) / abs(l_ref),
            "step0": step0,
            "what": f"{nsteps} Adam steps from initialparameters(161803) on batches 0..{nsteps - 1} of the bench's own (un-scaled) dataset, then the "
                    f"loss of batch {nsteps}: GPU engine (same kernel and mode as the timed run), oracle/eh_oracle.c (fp32 port, checker only) and the fp64 "
                    "oracle (oracle/hybrid_oracle.py), whose trajectory is the truth the two fp32 ones drift from -- the `*_distance_from_fp64` pairs say "
                    "how far each got (Adam's first steps are lr * sign-like, so a gradient entry near zero -- saturated first-layer units at this input "
                    "scale -- turns rounding into a full step for that parameter: single entries of theta part by up to nsteps * lr on EITHER fp32 side); "
                    "the one-step comparison `step0` is the 1e-5 parity statement"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mech-stage", action="store_true", help="skip the secondary measurement of the stand-alone mechanistic + VJP kernel")
    ap.add_argument("--no-epoch", action="store_true", help="skip the secondary measurement of the shuffled epoch (eh_train_epoch)")
    ap.add_argument("--no-layerwise", action="store_true", help="skip the secondary measurement of the reference's GPU tutorial network (layer-wise form)")
    ap.add_argument("--no-train-e2e", action="store_true", help="skip the secondary end-to-end measurement of eh.train(...) (tools/bench_train_e2e.py)")
    ap.add_argument("--no-specialize", action="store_true",
                    help="never compile a step kernel at run time (the headline descriptor has a kernel specialised ahead of time, csrc/eh_spec.hip, and runs it "
                         "either way; EH_NO_AOT_SPEC=1 takes that one away: then this flag picks the generic kernels over the hiprtc-built ones)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  Nothing in this process has touched a GPU yet (no torch import, no HIP call).
        sys.exit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1 and not args.no_cpu_baseline:
        # the CPU checker is compiled (gcc: a fork + exec) BEFORE this process touches the GPU -- a process that has initialised
        # HIP must not start other programs on the GPU boxes; once the GPU is up, c_oracle.build() refuses instead of compiling
        from oracle import c_oracle
        c_oracle.build()
    import torch
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # EH_BENCH_SHARE_GPU=1 (testing only): all ranks on GPU 0 with gloo carrying the collectives -- the only way to walk
    # the multi-rank flow of this script on a one-GPU box (RCCL refuses two ranks on one device); numbers are meaningless
    share = os.environ.get("EH_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but this node shows {torch.cuda.device_count()} GPU(s) (EH_BENCH_SHARE_GPU=1 walks the multi-rank flow on one GPU: testing only)")
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if share:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        except Exception as e:                              # nothing has run yet: the launcher may try again on another port
            print(f"bench.py rank {rank}: rendezvous failed: {e!r}", file=sys.stderr, flush=True)
            sys.exit(EX_RENDEZVOUS if os.environ.get("EH_BENCH_LAUNCHER") else 1)

    import easyhybrid_jl_amd as eh
    from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10

    B = args.batch
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
    cols = make_synth_rbq10(NBATCHES * B, seed=42 + rank)           # each rank: its own shard (weak scaling)
    X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
    t_up = [0.0]

    def new_engine(single):
        """a fresh engine holding this rank's shard; single: the one-GPU step mode (one kernel per step on the specialised kernel)"""
        e = model.engine(local)
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        t = time.perf_counter()
        e.set_data(eh.EH_SPLIT_TRAIN, X, [cols["ta"]], [cols["reco"]])
        e.synchronize()
        t_up[0] = time.perf_counter() - t   # one-time host -> HBM upload (interleave on the host + PCIe); never part of `value`
        e.set_params(model.initialparameters(161803))                # same seed on every rank: replicas start equal
        e.opt_init("Adam", 0.01, 0.9, 0.999, 1e-8)
        if share:
            e.set_option("max_blocks", max(1, 128 // world))          # every rank's kernel has to fit on the shared GPU at once
        if single:
            # one kernel per step: the optimiser update of step s runs in the prologue of step s+1 and the
            # partial sums are accumulated with float atomics (opt-in mode)
            e.set_option("fused_update", int(os.environ.get("EH_FUSED", "1")))
            if not args.no_specialize:
                # the "specialize" option: the step kernel is compiled at run time (hiprtc, ~1 s) with this model's descriptor as a
                # compile-time constant; one forward over one sample builds it here, ahead of the warm-up
                e.set_option("specialize", 1)
            e.forward(eh.EH_SPLIT_TRAIN, 0, 1, params=False)          # (builds / picks the step kernels here, ahead of the warm-up; eh_jit_status then says which)
        return e

    def fence(e):
        e.synchronize()                # also applies the last step's pending update (fused-update mode)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(stepfn, e):
        for s in range(args.warmup):
            stepfn((s % NBATCHES) * B)
        fence(e)
        t0 = time.perf_counter()
        for s in range(args.steps):
            stepfn(((args.warmup + s) % NBATCHES) * B)
        fence(e)
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device="cpu" if share else "cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    force_dp = os.environ.get("EH_FORCE_DP", "0") == "1"          # exercise the data-parallel seam on one GPU
    n1_ref = None
    if world > 1:
        # the N = 1 figure measured in THIS run, for comparison: every rank runs the one-GPU step mode on its own GPU and shard,
        # same warm-up and step count, no exchange; the slowest rank's time counts (what weak scaling is measured against)
        e1 = new_engine(True)
        dt1 = timed(lambda first: e1.train_step(first, B, want_loss=False), e1)
        e1.close()
        n1_ref = {"value": B * args.steps / dt1, "unit": "samples/s", "ms_per_step": 1e3 * dt1 / args.steps,
                  "what": "one-GPU step mode (no exchange), all ranks at once on their own GPUs, max over ranks; same steps / warm-up"}
    eng = new_engine(world == 1 and not force_dp)
    if world > 1 or force_dp:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        dp = eh.dp.DataParallel(eng, specialize=not args.no_specialize)
        step = lambda first: dp.step(first, B)
    else:
        dp = None
        step = lambda first: eng.train_step(first, B, want_loss=False)
    jit_kernels, jit_log = eng.jit_status()
    built = ("specialised for this model descriptor AHEAD of time (csrc/eh_spec.hip: no run-time compiler involved)" if jit_log.startswith("ahead-of-time")
             else "compiled at run time around the model descriptor (hiprtc)" if jit_kernels else "generic kernel built ahead of time")

    exchange_cal = None
    if dp is not None and dp.p2p:
        # burn-in of the peer-to-peer exchange (a deadline hit on any rank drops every rank back to the RCCL
        # all-reduce), then time both exchanges for a few hundred steps and keep the faster one
        for s in range(64):
            step((s % NBATCHES) * B)
        if dp.check():
            exchange_cal = dp.calibrate(0, B, 300)
    dt = timed(step, eng)

    # live kernel timing: more steps with HIP events bracketing the fused step kernel
    roof = None
    if rank == 0 or dp is not None:
        # one HIP event pair (on the engine's stream) around every burst of BURST consecutive launches: an event
        # between two back-to-back 13 us kernels would add ~2 us to each, a burst measures the steady-state rate.
        # At least five bursts whatever --steps says, so that the percentiles below are over several samples.
        # Under data parallelism every rank runs the loop (the step kernels of the peer-to-peer mode wait for each
        # other), with the exchange inside the bracket when it is part of the kernel.
        BURST = 50
        nprof = max(5 * BURST, min(args.steps, 4000) // BURST * BURST)
        eng.profile_enable(BURST)
        if dp is None or dp.p2p:
            for s in range(nprof):
                step((s % NBATCHES) * B)
        else:
            for s in range(nprof):                                 # local part only: no collective inside the bracket
                if dp.fused:
                    eng.dp_fused_step((s % NBATCHES) * B, B)
                else:
                    eng.dp_grad((s % NBATCHES) * B, B)
        per_launch = eng.profile_samples() / BURST                 # ms per launch, one value per burst
        n, _, _ = eng.profile_read()
        eng.profile_enable(False)
        n *= BURST
        ms_step = float(per_launch.mean()) if per_launch.size else 0.0
        if dp is not None and dp.p2p:
            fence(eng)
    if rank == 0:
        if n and ms_step > 0:
            tf = FLOP_PER_SAMPLE * B / (ms_step * 1e-3) / 1e12
            gbs = BYTES_PER_SAMPLE * B / (ms_step * 1e-3) / 1e9
            traffic, traffic_src = None, None
            try:      # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KiB)
                tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
                if tj.get("batch") == B and tj.get("fused") == (dp is None or dp.fused):
                    traffic, traffic_src = (2.0 * tj["FETCH_SIZE_KiB"] + tj["WRITE_SIZE_KiB"]) * 1024.0, tj["source"]
            except Exception:
                pass
            roof = {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F32_TFLOPS,
                    "traffic": traffic, "traffic_kind": "static: read from the committed PMC passes, NOT measured in this run (hardware counters need rocprofv3 around the process)" if traffic is not None else None,
                    "traffic_source": traffic_src, "kernel": "eh_step_kernel<NBI=1,NBH=1,NL=2,NT=2,NW=8,tanh,train+p2p,K1|PS> (fused update, peer-to-peer exchange)" if (dp is not None and dp.p2p) else
                    "eh_step_kernel<NBI=1,NBH=1,NL=2,NT=2,NW=8,tanh,train,K1|PS> (fused update)" if (dp is None or dp.fused) else
                    "eh_step_kernel<NBI=1,NBH=1,NL=2,NT=2,NW=8,tanh,train,K1|PS> + eh_reduce_kernel", "kernel_build": built, "kernel_ms": ms_step, "launches_timed": n,
                    "bursts_timed": int(per_launch.size),
                    "kernel_ms_p10_p50_p90": [float(np.percentile(per_launch, q)) for q in (10, 50, 90)],
                    "timing": f"HIP events on the engine stream around bursts of {BURST} launches" + (" (step kernel + reduce kernel per launch)" if (dp is not None and not dp.fused) else "")
                              + (" (includes the wait for the peer GPUs' sums: the exchange is part of the kernel)" if (dp is not None and dp.p2p) else ""),
                    "algorithmic": {"flop_per_launch": FLOP_PER_SAMPLE * B, "bytes_per_launch": BYTES_PER_SAMPLE * B},
                    "hbm_achieved_GBps": gbs, "hbm_frac": gbs / PEAK_HBM_GBPS}
    devices = None
    if world > 1:
        props = torch.cuda.get_device_properties(local)
        mine = {"rank": rank, "device": local, "name": props.name, "pid": os.getpid()}
        devices = [None] * world
        dist.all_gather_object(devices, mine)
    fence(eng)

    # the path train() really runs: a shuffled epoch gathers 16-byte records through a device-side permutation
    # (eh_train_epoch, reference src/data/loaders.jl:1-12 + src/training/epoch.jl:13-33); contiguous epoch beside it
    epoch = None
    if rank == 0 and world == 1 and dp is None and not args.no_epoch:
        try:
            epoch = {}
            for name, shuffle in (("contiguous", False), ("shuffled", True)):
                for k in range(2):
                    eng.train_epoch(B, seed=11 + k, shuffle=shuffle, want_loss=False)
                eng.synchronize()
                reps, t0 = 8, time.perf_counter()
                for k in range(reps):
                    eng.train_epoch(B, seed=100 + k, shuffle=shuffle, want_loss=False)
                eng.synchronize()
                per = (time.perf_counter() - t0) / (reps * NBATCHES)
                epoch[name] = {"us_per_step": 1e6 * per, "samples_per_s": B / per, "algorithmic_GBps": BYTES_PER_SAMPLE * B / per / 1e9}
            epoch["shuffled_over_contiguous"] = epoch["shuffled"]["us_per_step"] / epoch["contiguous"]["us_per_step"]
            epoch["what"] = (f"eh_train_epoch(batchsize={B}) over the {NBATCHES} resident batches, {reps} epochs, host clock around the calls incl. the "
                             "permutation kernel; shuffled = every step gathers its 16-byte records through the epoch's device-side permutation")
        except Exception as e:
            epoch = {"error": repr(e)}

    if rank == 0:
        out = {
            "metric": "training samples/sec", "value": world * B * args.steps / dt, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RbQ10 hybrid, MLP [2,16,16,1] tanh -> rb (sigmoid-scaled), Q10 global, MSE + Adam(0.01), "
                                   f"batch={B} per GPU, fp32 (BASELINE.json configs[1])",
                       "global_batch": world * B, "resident_batches_per_gpu": NBATCHES, "parallelism": f"dp{world}",
                       "ranks_seen": (dist.get_world_size() if dist.is_initialized() else 1),
                       "collective_backend": (dist.get_backend() if dist.is_initialized() else None),
                       "launcher": os.environ.get("EH_BENCH_LAUNCHER", "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else ("none" if world == 1 else "external")),
                       "rank_devices": devices,
                       "gradient_exchange": ("none (one GPU)" if dp is None else "peer-to-peer stores from the step kernel (eh_p2p_*), no collective call per step" if dp.p2p
                                             else "one RCCL all-reduce per step" if not share else "one gloo all-reduce per step (EH_BENCH_SHARE_GPU=1: all ranks on one GPU, testing only)"),
                       "gradient_exchange_calibration_us_per_step": exchange_cal,
                       "gradient_exchange_negotiation": getattr(dp, "p2p_report", None) if dp is not None else None, "step_kernel": built,
                       "step_mode": "one kernel per step (fused_update = 1: float-atomic sums) on the kernel specialised for this descriptor (see step_kernel): what "
                                    "train(fused_update = True) or an unseeded train() runs; a seeded train() -- the default, bitwise reproducible -- takes the deterministic "
                                    "step + reduce pair at this batch size (train_e2e: steps_s against steps_s_fused_update_true)" if dp is None else "see step_kernel / gradient_exchange"},
            "roofline": roof,
        }
        if n1_ref is not None:
            out["n1_reference"] = n1_ref
            out["weak_scaling_vs_n1_in_this_run"] = out["value"] / (world * n1_ref["value"])
        out["dataset_upload_ms_once"] = 1e3 * t_up[0]
        if epoch is not None:
            out["epoch"] = epoch
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B)
            try:
                out["parity"] = parity_replay(model, lambda: new_engine(dp is None), cols, X, B)
                out["final_loss"], out["oracle_final_loss"] = out["parity"]["final_loss"], out["parity"]["oracle_final_loss"]
            except Exception as e:
                out["parity"] = {"error": repr(e)}
        if world == 1 and dp is None and not args.no_mech_stage:
            # the HBM-bound stage of the path measured as its own kernel (the fused step is compute / latency
            # bound, so the north_star's HBM yardstick applies to the mechanistic + loss + VJP stage alone): eh_mech_loss_vjp on
            # 1 024 resident batches of the headline workload (a working set the Infinity Cache cannot hold).  A secondary figure; `roofline` above stays the step kernel's.
            try:
                import importlib.util
                spec_ = importlib.util.spec_from_file_location("eh_bench_mech", os.path.join(ROOT, "tools", "bench_mech.py"))
                bm = importlib.util.module_from_spec(spec_); spec_.loader.exec_module(bm)
                out["hbm_stage"] = bm.measure("rbq10", 1024 * B, 50)      # 1 GiB of planes: four times the 256 MB Infinity Cache
            except Exception as e:      # never lose the headline line over the secondary measurement
                out["hbm_stage"] = {"error": repr(e)}
        if world == 1 and dp is None and not args.no_layerwise:
            # the other kernel family of the path: networks no fused kernel holds run layer by layer -- the
            # reference's own GPU tutorial network (docs/literate/tutorials/synthetic_respiration_gpu.jl:79-103: [1024, 512, 256, 128, 64],
            # sigmoid, input BatchNorm, RMSProp) at the tutorial's batch of 64 and at the headline batch.  A secondary figure.
            try:
                import importlib.util
                spec_ = importlib.util.spec_from_file_location("eh_bench_lform", os.path.join(ROOT, "tools", "bench_lform.py"))
                bl = importlib.util.module_from_spec(spec_); spec_.loader.exec_module(bl)
                out["layerwise"] = {"what": "eh_train_step on the reference's GPU tutorial network, fp32, steady state", "bound": "mfma", "peak_TFLOPs": 157.3,
                                    "runs": [bl.measure(64, local), bl.measure(B, local)]}
            except Exception as e:
                out["layerwise"] = {"error": repr(e)}
        if world == 1 and dp is None and not args.no_train_e2e:
            # what a user runs: eh.train(...) end to end on the reference tutorial's two models and on the headline data set, split into
            # steps / evaluation / host, next to the same work in PyTorch-CPU eager; and the evaluation kernel's own roofline entry
            try:
                import importlib.util
                spec_ = importlib.util.spec_from_file_location("eh_bench_e2e", os.path.join(ROOT, "tools", "bench_train_e2e.py"))
                be = importlib.util.module_from_spec(spec_); spec_.loader.exec_module(be)
                out["train_e2e"] = be.measure(local)
            except Exception as e:
                out["train_e2e"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    eng.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
